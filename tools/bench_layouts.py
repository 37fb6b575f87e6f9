"""Kernel time of a node model's dense-Jacobian evaluation in the two operand layouts of the C ABI:
unit-fastest ("soa": element stride = count) and instance-major ("aos": element stride 1, what a
VariableMap-style [x | u | ...] buffer per instance gives)."""
import json
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4096 * 32
m = ungar_amd.NodeModel(name)
nx, nu, npar = m.nx, m.nu, m.np
ncols = nx + nu
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
Op = ungar_amd.Operand
out = {}
for layout in ("soa", "aos"):
    shape = (lambda n: (n, count)) if layout == "soa" else (lambda n: (count, n))
    x = torch.rand(shape(nx), generator=gen, device="cuda", dtype=torch.float64)
    u = torch.rand(shape(nu), generator=gen, device="cuda", dtype=torch.float64) + 1.0
    p = torch.rand(npar, generator=gen, device="cuda", dtype=torch.float64) + 0.5
    f = torch.empty(shape(nx), dtype=torch.float64, device="cuda")
    J = torch.empty(shape(nx * ncols), dtype=torch.float64, device="cuda")
    mk = (lambda t, n: Op.soa(t, count, 1)) if layout == "soa" else (lambda t, n: Op.aos(t, n, 1))
    ops = (count, mk(x, nx), mk(u, nu), None, Op.per_instance(p, npar, shared=True), mk(f, nx), mk(J, nx * ncols))
    for _ in range(3):
        m.dense_jacobian(*ops)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        m.dense_jacobian(*ops)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    out[layout] = {"kernel_ms": ms, "GBs": count * 8 * (ncols + nx + nx * ncols) / ms / 1e6}
print(json.dumps({"model": name, "count": count, **out}))
