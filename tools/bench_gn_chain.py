#!/usr/bin/env python3
"""BASELINE config 4 chain on one MI355X: ANYmal node Jacobians (lane-per-leg kernel, unit-fastest J with the padded element stride)
-> Gauss-Newton term G = upper(J^T diag(d) J) by the lane-per-(node, block) kernel (gn_hessian_tiles.hip), 4096 instances x 20 knots.
Prints one JSON line: per-kernel and chain times after a pre-warm, algorithmic bytes, fraction of 8 TB/s.
usage: bench_gn_chain.py [--natural-stride] [--reps R]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402
from ungar_amd.sharding import padded_stride, unit_fastest  # noqa: E402

natural = "--natural-stride" in sys.argv
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
rows, cols, N, batch = 37, 49, 20, 4096
count = N * batch
m = ungar_amd.NodeModel("anymal")
x0, u0, _, p = W.synth_device_inputs("anymal", count, 0, torch)
Op = ungar_amd.Operand


def operand(elements, src=None):
    t = torch.empty((elements, count), dtype=torch.float64, device="cuda") if natural else unit_fastest(elements, count, torch)
    if src is not None:
        t.copy_(src)
    return t


x, u, f, J = operand(m.nx, x0), operand(m.nu, u0), operand(rows), operand(rows * cols)
d, G = operand(rows, torch.rand((rows, count), device="cuda", dtype=torch.float64)), operand(cols * cols)
es = J.stride(0)
ops = (count, Op.soa(x, es, N), Op.soa(u, es, N), None, Op.per_instance(p, 1, shared=True), Op.soa(f, es, N), Op.soa(J, es, N))
jac = lambda: m.dense_jacobian(*ops, knots=N)  # noqa: E731
gn = lambda: ungar_amd.gn_hessian_tiles(J, d, G, rows, cols, count, True)  # noqa: E731


def timeit(fn):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


t_end = time.perf_counter() + 0.5  # steady-state clocks (bench.py does the same)
while time.perf_counter() < t_end:
    for _ in range(20):
        jac()
        gn()
    torch.cuda.synchronize()
upper = cols * (cols + 1) // 2
jac_bytes = count * W.algorithmic_bytes(m.nx, m.nu, None)
gn_bytes = count * 8 * (rows * cols + rows + upper)
t_jac, t_gn, t_chain = timeit(jac), timeit(gn), timeit(lambda: (jac(), gn()))
sl = slice(count - 1024, count)
Jv = J.view(rows, cols, count)[:, :, sl] if natural else J.reshape(rows, cols, count)[:, :, sl]
ref = torch.einsum("ran,rn,rbn->abn", Jv, d[:, sl], Jv)
iu = torch.triu_indices(cols, cols, device="cuda")
err = float((G.reshape(cols, cols, count)[iu[0], iu[1]][:, sl] - ref[iu[0], iu[1]]).abs().max() / ref.abs().max())
print(json.dumps({"workload": "anymal node Jacobians -> upper(J^T diag(d) J), N=20 x 4096 instances = 81920 nodes", "element_stride": es, "reps": reps,
                  "node_jacobian": {"ms": t_jac, "algorithmic_GB": jac_bytes / 1e9, "frac_of_8TBs": jac_bytes / t_jac / 1e6 / 8000},
                  "gn_tiles": {"ms": t_gn, "algorithmic_GB": gn_bytes / 1e9, "frac_of_8TBs": gn_bytes / t_gn / 1e6 / 8000,
                               "useful_TFLOPs": count * 2 * rows * upper / t_gn / 1e9},
                  "chain": {"ms": t_chain, "algorithmic_GB": (jac_bytes + gn_bytes) / 1e9, "frac_of_8TBs": (jac_bytes + gn_bytes) / t_chain / 1e6 / 8000,
                            "nodes_per_s": count / t_chain * 1e3},
                  "max_rel_err_vs_torch_on_1024_nodes": err}))
