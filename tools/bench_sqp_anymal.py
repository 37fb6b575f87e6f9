#!/usr/bin/env python3
"""One batched SQP iteration of the full-body quadruped OCP (BASELINE config 4's model and sizes: nx = 37, nu = 12, N = 20, 4096
instances): ANYmal node Jacobians, `anymal_cost`, stage QP data, Riccati solve, stacked line search -- all on the device."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ungar_amd import sqp  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402

batch, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 20
x, u, _, p = W.synth_device_inputs("anymal", batch * (N + 1), 3, torch)  # unit-fastest (elements, nodes)
X = x.t().reshape(batch, N + 1, 37).contiguous()
X[:, 1:] = X[:, :1] + 0.05 * (X[:, 1:] - X[:, :1])
X[:, :, 3:7] /= X[:, :, 3:7].norm(dim=2, keepdim=True)
U = u.t().reshape(batch, N + 1, 12)[:, :N].contiguous()
xm = X[:, 0] + 0.01 * torch.randn((batch, 37), device="cuda", dtype=torch.float64)
ref = X[:, 0].clone()
ref[:, 19:] = 0.0
pc = torch.cat((ref, torch.tensor([10.0, 10.0, 1.0, 0.1, 1e-3], device="cuda", dtype=torch.float64).expand(batch, 5)), dim=1).contiguous()
pd = p if p.dim() == 1 else p[0]
solver = sqp.BatchedSoftSqp("anymal", "anymal_cost", N, batch)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


X0, U0 = X.clone(), U.clone()
qp_ms = timeit(lambda: solver.qp_step(X, U, xm, pd, pc), 10)
it_ms = timeit(lambda: solver.iterate(X, U, xm, pd, pc), 5)
theta_after = solver.theta0.clone()
X.copy_(X0)
U.copy_(U0)
solver.iterate(X, U, xm, pd, pc)
theta_first = solver.theta0.clone()
print(json.dumps({"workload": f"full-body quadruped OCP (anymal + anymal_cost), nx=37 nu=12 N={N}, {batch} instances", "ms_per_qp_step": qp_ms, "ms_per_sqp_iteration": it_ms,
                  "instances_per_s": batch / it_ms * 1e3, "knots_per_s": batch * N / it_ms * 1e3, "riccati_status_nonzero": int((solver.status != 0).sum()),
                  "median_theta_first": float(theta_first.median()), "median_theta_after_6_iterations": float(theta_after.median())}))
