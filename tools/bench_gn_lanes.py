#!/usr/bin/env python3
"""Gauss-Newton term for unit-fastest Jacobians: the lane-per-node FP64-vector kernel (gn_hessian_lanes.hip) against the LDS-staged
MFMA kernel (gn_hessian_soa.hip), ANYmal block 37 x 49, 81 920 nodes, and the config-4 chain node Jacobians -> GN term.
UNGAR_GN_LANES_UNROLL selects the unroll variant (1, 2, 4) of the lanes kernel; run once per value."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402

if os.environ.get("UNGAR_GN_LANES_UNROLL") or os.environ.get("UNGAR_GN_TILES_DMA"):
    ungar_amd.use_library(ungar_amd.measurement_library_path())  # the knobs exist only in the measurement build
from ungar_amd import workloads as W  # noqa: E402

rows, cols, N, batch = 37, 49, 20, 4096
count = N * batch
m = ungar_amd.NodeModel("anymal")
x, u, _, p = W.synth_device_inputs("anymal", count, 0, torch)
f = torch.empty((rows, count), dtype=torch.float64, device="cuda")
J = torch.empty((rows * cols, count), dtype=torch.float64, device="cuda")
d = torch.rand((rows, count), device="cuda", dtype=torch.float64)
Op = ungar_amd.Operand
ops = (count, Op.soa(x, count, N), Op.soa(u, count, N), None, Op.per_instance(p, 1, shared=True), Op.soa(f, count, N), Op.soa(J, count, N))
m.dense_jacobian(*ops, knots=N)
G_uf = torch.zeros((cols * cols, count), dtype=torch.float64, device="cuda")
G_nm = torch.zeros((count, cols, cols), dtype=torch.float64, device="cuda")


def timeit(fn, reps=20):
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


upper = cols * (cols + 1) // 2
alg_bytes = count * 8 * (rows * cols + rows + upper)
flops = count * 2 * rows * upper  # useful multiply-adds of the upper triangle
out = {"unroll": os.environ.get("UNGAR_GN_LANES_UNROLL", "default"), "stage": os.environ.get("UNGAR_GN_TILES_STAGE", "default"), "nodes": count}
for name, fn in (("tiles_unit_fastest_out", lambda: ungar_amd.gn_hessian_tiles(J, d, G_uf, rows, cols, count, True)),
                 ("tiles_node_major_out", lambda: ungar_amd.gn_hessian_tiles(J, d, G_nm, rows, cols, count, False)),
                 ("lanes_unit_fastest_out", lambda: ungar_amd.gn_hessian_lanes(J, d, G_uf, rows, cols, count, True)),
                 ("lanes_node_major_out", lambda: ungar_amd.gn_hessian_lanes(J, d, G_nm, rows, cols, count, False)),
                 ("mfma_lds_staged_node_major_out", lambda: ungar_amd.gn_hessian_unit_fastest(J, d, G_nm, rows, cols, count))):
    ms = timeit(fn)
    out[name] = {"ms": ms, "nodes_per_s": count / ms * 1e3, "hbm_frac_of_8TBs_algorithmic": alg_bytes / ms / 1e6 / 8000, "useful_TFLOPs": flops / ms / 1e9}
ref = torch.einsum("ran,rn,rbn->abn", J.view(rows, cols, count)[:, :, :4096], d[:, :4096], J.view(rows, cols, count)[:, :, :4096])
iu = torch.triu_indices(cols, cols, device="cuda")
out["max_err_vs_torch"] = float((G_uf.view(cols, cols, count)[iu[0], iu[1], :4096] - ref[iu[0], iu[1]]).abs().max() / ref.abs().max())
ungar_amd.gn_hessian_tiles(J, d, G_uf, rows, cols, count, True)
out["max_err_tiles_vs_torch"] = float((G_uf.view(cols, cols, count)[iu[0], iu[1], :4096] - ref[iu[0], iu[1]]).abs().max() / ref.abs().max())
chain = timeit(lambda: (m.dense_jacobian(*ops, knots=N), ungar_amd.gn_hessian_tiles(J, d, G_uf, rows, cols, count, True)))
out["chain_node_jacobian_plus_gn_ms"] = chain
out["node_jacobian_ms"] = timeit(lambda: m.dense_jacobian(*ops, knots=N))
print(json.dumps(out))
