"""BASELINE config 3 end to end on one MI355X: ANYmal shooting-node Jacobians (lane-per-leg kernel) followed
by the Gauss-Newton term G = J^T diag(d) J on the FP64 matrix cores, J handed over in the unit-fastest
layout with no transpose in between.  The batch is processed in slices; a slice whose Jacobian fits the
256 MB last-level cache is written with write-back stores (UseStreamingStores) and read back by the
contraction from cache instead of HBM."""
import json
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd.workloads import synth_device_inputs  # noqa: E402

N, batch = 20, 4096
count = N * batch
m = ungar_amd.NodeModel("anymal")
nx, ncols = m.nx, m.nx + m.nu
x, u, _, p = synth_device_inputs("anymal", count, 0, torch)
d = torch.rand((nx, count), device="cuda", dtype=torch.float64)
Op = ungar_amd.Operand
results = []
for nodes_per_slice in (count, 32768, 16384, 8192):
    f = torch.empty((nx, nodes_per_slice), dtype=torch.float64, device="cuda")
    J = torch.empty((nx * ncols, nodes_per_slice), dtype=torch.float64, device="cuda")
    G = torch.empty((count, ncols, ncols), dtype=torch.float64, device="cuda")
    slices = [(s, min(nodes_per_slice, count - s)) for s in range(0, count, nodes_per_slice)]

    def step():
        for s, n in slices:
            xs, us, ds = x[:, s:s + n], u[:, s:s + n], d[:, s:s + n]
            ops = (n, Op(xs, 1, 1, xs.stride(0)), Op(us, 1, 1, us.stride(0)), None, Op.per_instance(p, m.np, shared=True),
                   Op(f, 1, 1, f.stride(0)), Op(J, 1, 1, J.stride(0)))
            m.dense_jacobian(*ops)
            ungar_amd.gn_hessian_unit_fastest(J[:, :n], ds, G[s:s + n], nx, ncols, n)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(10):
        step()
    e0.record()
    torch.cuda.synchronize()
    ms = s0.elapsed_time(e0) / 10
    results.append({"nodes_per_slice": nodes_per_slice, "ms_per_batch": ms, "nodes_per_s": count / ms * 1e3})
print(json.dumps({"pipeline": "anymal dense [A|B] -> upper(J^T diag(d) J), N=20 x batch 4096", "results": results}))
