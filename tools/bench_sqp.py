#!/usr/bin/env python3
"""Batched soft SQP on the device (ungar_amd.sqp.BatchedSoftSqp): time per iteration and per QP step for the quadrotor OCP
(N = 30, rotor bounds) over a batch of instances; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
usage: bench_sqp.py [batch]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ungar_amd import sqp  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402

batch, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 30
rng = np.random.default_rng(3)
hover = np.sqrt(1.5 * 9.80665 / (4 * 0.015))
quat = rng.normal(size=(batch, N + 1, 4)) * 0.1 + np.array([0, 0, 0, 1.0])
quat /= np.linalg.norm(quat, axis=2, keepdims=True)
X = np.concatenate((rng.uniform(-0.5, 0.5, (batch, N + 1, 3)), quat, rng.uniform(-0.3, 0.3, (batch, N + 1, 6))), axis=2)
U = hover * rng.uniform(0.7, 1.3, (batch, N, 4))
xm = X[:, 0] + rng.normal(size=(batch, 13)) * 0.02
qref = rng.normal(size=(batch, 4)) * 0.1 + np.array([0, 0, 0, 1.0])
p_cost = np.concatenate((rng.uniform(-1, 1, (batch, 3)), qref / np.linalg.norm(qref, axis=1, keepdims=True), np.zeros((batch, 6))), axis=1)
p_dyn = np.tile(W.default_params("quadrotor"), (batch, 1))
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
Xd, Ud, xmd, pd, pc, pi = dev(X), dev(U), dev(xm), dev(p_dyn), dev(p_cost), dev(np.full((batch, 1), 2.0 * hover))
solver = sqp.BatchedSoftSqp("quadrotor", "quadrotor_cost", N, batch, inequality="quadrotor_ineq")


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


qp_ms = timeit(lambda: solver.qp_step(Xd, Ud, xmd, pd, pc, pi), 20)
X0, U0 = Xd.clone(), Ud.clone()
it_seq_ms = timeit(lambda: solver.iterate(Xd, Ud, xmd, pd, pc, pi, stacked=False), 10)  # six small launches per candidate
Xd.copy_(X0)
Ud.copy_(U0)
it_ms = timeit(lambda: solver.iterate(Xd, Ud, xmd, pd, pc, pi), 10)  # stacked line search (default)
print(json.dumps({"workload": f"quadrotor OCP N={N}, {batch} instances, rotor bounds behind the POLY barrier", "ms_per_qp_step": qp_ms, "ms_per_sqp_iteration": it_ms, "ms_per_sqp_iteration_candidate_by_candidate": it_seq_ms, "line_search_candidates": len(solver.candidate_steps()),
                  "instances_per_s": batch / it_ms * 1e3, "knots_per_s": batch * N / it_ms * 1e3, "riccati_status_nonzero": int((solver.status != 0).sum())}))
