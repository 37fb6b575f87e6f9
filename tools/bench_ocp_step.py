"""Everything one soft-SQP iteration of the single-rigid-body quadruped OCP needs from the derivative engine
(quadruped.example.cpp, N = 30), for a batch of independent MPC instances, as device kernels only:
  dynamics value + dense [A | B] per knot      ungar_model_dense_jacobian("srbd")
  whole-horizon equality Jacobian assembly     ungar_ocp_assemble_equality
  stage cost value + gradient + upper Hessian  ungar_model_sparse_hessian("srbd_cost")
  inequality rows + Jacobian                   ungar_model_dense_jacobian("srbd_ineq")
  barrier Gauss-Newton term J^T b''(-h) J      torch elementwise + ungar_gn_hessian_upper_unit_fastest
Reports the time of one such sweep (what the reference does with seven generated-C calls per instance)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from oracle import ungar_oracle as O  # noqa: E402  (parameter values and input ranges only)

N, batch = 30, 4096
count = N * batch
dev = "cuda"
Op = ungar_amd.Operand
dyn, cost, ineq = (ungar_amd.NodeModel(n) for n in ("srbd", "srbd_cost", "srbd_ineq"))
xs, us, ws, _ = O.synthetic_inputs("srbd", 4096, seed=5)
rep = (count + 4095) // 4096
t = lambda a: torch.as_tensor(np.tile(a, (rep, 1))[:count].T.copy(), device=dev)  # noqa: E731  (elements, count)
x, u, w = t(xs), t(us), t(ws)
p_dyn = torch.as_tensor(O.default_params("srbd"), device=dev)
p_ineq = torch.as_tensor(O.default_params("srbd_ineq"), device=dev)
_, _, refs = O.synthetic_cost_inputs(4096, seed=5, name="srbd_cost")
p_cost = t(refs)
nx, nu = dyn.nx, dyn.nu
ncols = nx + nu
f = torch.empty((nx, count), dtype=torch.float64, device=dev)
J = torch.empty((nx * ncols, count), dtype=torch.float64, device=dev)
y = torch.empty((1, count), dtype=torch.float64, device=dev)
grad = torch.empty((ncols, count), dtype=torch.float64, device=dev)
hrows, _ = cost.hessian_sparsity()
hes = torch.empty((len(hrows), count), dtype=torch.float64, device=dev)
h = torch.empty((ineq.ny, count), dtype=torch.float64, device=dev)
Jh = torch.empty((ineq.ny * ncols, count), dtype=torch.float64, device=dev)
G = torch.empty((count, ncols, ncols), dtype=torch.float64, device=dev)
k, e = 100.0, 2e-5
a1, a2 = k, (0.5 * k * e - k * e) / e ** 2


def sweep():
    dyn.dense_jacobian(count, Op.soa(x, count, N), Op.soa(u, count, N), Op.soa(w, count, N), Op.per_instance(p_dyn, dyn.np, shared=True), Op.soa(f, count, N),
                       Op.soa(J, count, N), knots=N)
    cost.sparse_hessian(count, Op.soa(x, count, N), Op.soa(u, count, N), None, Op.soa(p_cost, count, N), Op.soa(y, count, N), Op.soa(grad, count, N),
                        Op.soa(hes, count, N), knots=N)
    ineq.dense_jacobian(count, Op.soa(x, count, N), Op.soa(u, count, N), Op.soa(w, count, N), Op.per_instance(p_ineq, ineq.np, shared=True), Op.soa(h, count, N),
                        Op.soa(Jh, count, N), knots=N)
    z = -h
    d2 = torch.where(z < 0.0, a1, torch.where(z < e, 2.0 * a2 * z + a1, 0.0))
    ungar_amd.gn_hessian_unit_fastest(Jh, d2, G, ineq.ny, ncols, count)


for _ in range(3):
    sweep()
torch.cuda.synchronize()
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
for _ in range(20):
    sweep()
e0.record()
torch.cuda.synchronize()
ms = s0.elapsed_time(e0) / 20
assert torch.isfinite(J).all() and torch.isfinite(G[:, 0, 0]).all() and torch.isfinite(hes).all()
print(json.dumps({"workload": f"SRBD quadruped OCP, N={N}, batch={batch}: dynamics [A|B] + stage cost (value, gradient, Hessian) + inequality rows and Jacobian "
                              "+ barrier Gauss-Newton term, per knot", "ms_per_sweep": ms, "knots_per_s": count / ms * 1e3, "instances_per_s": batch / ms * 1e3}))
