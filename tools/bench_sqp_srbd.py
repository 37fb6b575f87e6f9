#!/usr/bin/env python3
"""One batched SQP iteration of the reference's quadruped OCP (example/mpc/quadruped.example.cpp: single-rigid-body dynamics with
contact flags, nx = 13, nu = 24, N = 30, friction-cone rows behind the POLY barrier), 4096 instances."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ungar_amd import sqp  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402

batch, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 30
x, u, w, p = W.synth_device_inputs("srbd", batch * (N + 1), 11, torch)
X = x.t().reshape(batch, N + 1, 13).contiguous()
U = u.t().reshape(batch, N + 1, 24)[:, :N].contiguous()
Wn = w.t().reshape(batch, N + 1, 4)[:, :N].contiguous()
xm = X[:, 0] + 0.01 * torch.randn((batch, 13), device="cuda", dtype=torch.float64)
ref = X[:, 0] + 0.3 * torch.randn((batch, 13), device="cuda", dtype=torch.float64)
ref[:, 3:7] /= ref[:, 3:7].norm(dim=1, keepdim=True)
feet = torch.cat([U[:, 0, 6 * leg + 3:6 * leg + 6] for leg in range(4)], dim=1)
pc = torch.cat((ref, feet), dim=1).contiguous()
pd = (p if p.dim() == 1 else p[0]).contiguous()
pi = torch.as_tensor(W.default_params("srbd_ineq"), device="cuda")
solver = sqp.BatchedSoftSqp("srbd", "srbd_cost", N, batch, inequality="srbd_ineq", constraint_violation_multiplier=1.0 / 30.0, stiffness=1.0, epsilon=1.0)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


qp_ms = timeit(lambda: solver.qp_step(X, U, xm, pd, pc, pi, w=Wn), 10)
it_ms = timeit(lambda: solver.iterate(X, U, xm, pd, pc, pi, w=Wn), 5)
print(json.dumps({"workload": f"SRBD quadruped OCP (srbd + srbd_cost + srbd_ineq), nx=13 nu=24 N={N}, {batch} instances", "ms_per_qp_step": qp_ms, "ms_per_sqp_iteration": it_ms,
                  "instances_per_s": batch / it_ms * 1e3, "knots_per_s": batch * N / it_ms * 1e3, "riccati_status_nonzero": int((solver.status != 0).sum())}))
