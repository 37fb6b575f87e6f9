#!/usr/bin/env python3
"""Rates of the rigid-body quantity node models (correctness-first lane-per-node kernels, not tuned):
value and, where implemented, value + sparse Jacobian, 32 768 nodes, unit-fastest operands."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd.sharding import unit_fastest  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


count = int(sys.argv[1]) if len(sys.argv) > 1 else 32768  # (32 768 nodes are 512 wavefronts of a lane-per-node kernel: half of the 1024 SIMDs idle; pass 81920, 262144, ... for the sweep)
gen = torch.Generator(device="cuda")
gen.manual_seed(0)
out = {}
Op = ungar_amd.Operand
for name in ungar_amd.RBD_MODELS:
    m = ungar_amd.NodeModel(name)
    x, u = unit_fastest(m.nx, count, torch), unit_fastest(max(m.nu, 1), count, torch)
    x.copy_(torch.rand((m.nx, count), generator=gen, device="cuda", dtype=torch.float64) * 0.6 - 0.3)
    u.copy_(torch.rand((max(m.nu, 1), count), generator=gen, device="cuda", dtype=torch.float64) - 0.5)
    if m.nx >= 7:  # unit quaternion of the floating base
        q = x[3:7] + torch.tensor([0.0, 0.0, 0.0, 1.0], device="cuda", dtype=torch.float64)[:, None]
        x[3:7] = q / q.norm(dim=0, keepdim=True)
    p = torch.zeros((max(m.np, 1),), dtype=torch.float64, device="cuda")
    f = unit_fastest(m.ny, count, torch)
    st = x.stride(0)
    uo = Op.soa(u, st) if m.nu else None
    r = {"nx": m.nx, "nu": m.nu, "ny": m.ny}
    t = timeit(lambda: m.forward_zero(count, Op.soa(x, st), uo, None, Op.per_instance(p, max(m.np, 1), shared=True), Op.soa(f, st)))
    r["value_ms"], r["value_nodes_per_s"] = t, count / t * 1e3
    if m.implements_jacobian():
        J = unit_fastest(m.jac_nnz, count, torch)
        t = timeit(lambda: m.sparse_jacobian(count, Op.soa(x, st), uo, None, Op.per_instance(p, max(m.np, 1), shared=True), Op.soa(f, st), Op.soa(J, st)))
        r["jac_nnz"], r["jacobian_ms"], r["jacobian_nodes_per_s"] = m.jac_nnz, t, count / t * 1e3
        r["jacobian_GBs_written"] = count * 8 * (m.jac_nnz + m.ny) / t / 1e6
        if name == "anymal_rnea":  # dense block as well (structural zeros written): the other mode of the lane-per-leg program
            Jd = unit_fastest(m.ny * (m.nx + m.nu), count, torch)
            t = timeit(lambda: m.dense_jacobian(count, Op.soa(x, st), uo, None, Op.per_instance(p, max(m.np, 1), shared=True), Op.soa(f, st), Op.soa(Jd, st)))
            r["dense_jacobian_ms"], r["dense_jacobian_GBs_written"] = t, count * 8 * (m.ny * (m.nx + m.nu) + m.ny) / t / 1e6
    out[name] = r
out["nodes"] = count
print(json.dumps(out))
