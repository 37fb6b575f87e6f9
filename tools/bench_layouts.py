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
# instance-major caller through the unit-fastest kernel: transpose (x, u) in, launch, transpose (f, J) out
from ungar_amd.sharding import unit_fastest  # noqa: E402
xa = torch.rand((count, nx), generator=gen, device="cuda", dtype=torch.float64)
ua_ = torch.rand((count, nu), generator=gen, device="cuda", dtype=torch.float64) + 1.0
fa = torch.empty((count, nx), dtype=torch.float64, device="cuda")
Ja = torch.empty((count, nx * ncols), dtype=torch.float64, device="cuda")
xs, us, fs, Js = (unit_fastest(r, count, torch) for r in (nx, nu, nx, nx * ncols))
st = xs.stride(0)
ops = (count, Op.soa(xs, st, 1), Op.soa(us, st, 1), None, Op.per_instance(p, npar, shared=True), Op.soa(fs, st, 1), Op.soa(Js, st, 1))


def via():
    ungar_amd.transpose_nodes(xa, xs, count, nx, (nx, 1), (1, st))
    ungar_amd.transpose_nodes(ua_, us, count, nu, (nu, 1), (1, st))
    m.dense_jacobian(*ops)
    ungar_amd.transpose_nodes(fs, fa, count, nx, (1, st), (nx, 1))
    ungar_amd.transpose_nodes(Js, Ja, count, nx * ncols, (1, st), (nx * ncols, 1))


for _ in range(3):
    via()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    via()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
s.record()
for _ in range(20):
    ungar_amd.transpose_nodes(Js, Ja, count, nx * ncols, (1, st), (nx * ncols, 1))
e.record()
torch.cuda.synchronize()
tms = s.elapsed_time(e) / 20
out["aos_via_transposes"] = {"ms": ms, "jacobian_transpose_ms": tms, "jacobian_transpose_GBs": 2 * count * 8 * nx * ncols / tms / 1e6}
print(json.dumps({"model": name, "count": count, **out}))
