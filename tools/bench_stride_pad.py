#!/usr/bin/env python3
"""Element stride of the unit-fastest operands vs time: the ANYmal node-Jacobian kernel and the Gauss-Newton tiles kernel with the
element stride = count + pad nodes (count = 81 920 = 5 * 2^14: consecutive elements 5 * 2^17 bytes apart)."""
import json
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402

rows, cols, N, batch = 37, 49, 20, 4096
count = N * batch
m = ungar_amd.NodeModel("anymal")
x0, u0, _, p = W.synth_device_inputs("anymal", count, 0, torch)
Op = ungar_amd.Operand


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


out = {}
for pad in [int(a) for a in sys.argv[1:]] or (0, 16, 48, 80, 144, 1040):
    st = count + pad
    def padded(r):
        return torch.zeros((r, st), dtype=torch.float64, device="cuda")
    x, u = padded(m.nx), padded(m.nu)
    x[:, :count], u[:, :count] = x0, u0
    f, J, d, G = padded(rows), padded(rows * cols), padded(rows), padded(cols * cols)
    d[:, :count] = torch.rand((rows, count), device="cuda", dtype=torch.float64)
    ops = (count, Op.soa(x, st, N), Op.soa(u, st, N), None, Op.per_instance(p, 1, shared=True), Op.soa(f, st, N), Op.soa(J, st, N))
    jac = lambda: m.dense_jacobian(*ops, knots=N)
    gn = lambda: ungar_amd.gn_hessian_tiles(J, d, G, rows, cols, count, True)
    jac()
    out[str(pad)] = {"node_jacobian_ms": timeit(jac), "gn_tiles_ms": timeit(gn), "chain_ms": timeit(lambda: (jac(), gn()))}
    if pad == 0:
        ref_sum = float(J[:, :count].sum())
    out[str(pad)]["jacobian_sum_matches"] = abs(float(J[:, :count].sum()) - ref_sum) <= 1e-9 * abs(ref_sum)
print(json.dumps(out))
