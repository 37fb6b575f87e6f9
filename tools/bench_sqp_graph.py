#!/usr/bin/env python3
"""One batched SQP iteration of the quadrotor OCP replayed from a HIP graph (torch.cuda.CUDAGraph captures the stream-ordered C-ABI
launches: nothing in BatchedSoftSqp.iterate synchronises or reads back) against the eager launch sequence, over batch sizes: the
iteration is ~25 launches, so small batches are launch-bound.   usage: bench_sqp_graph.py [batch ...]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ungar_amd import sqp  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402

N = 30
out = {}
for batch in ([int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]):
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
    X0, U0 = Xd.clone(), Ud.clone()

    def run(fn, reps=50):
        Xd.copy_(X0)
        Ud.copy_(U0)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    eager = run(lambda: solver.iterate(Xd, Ud, xmd, pd, pc, pi))
    Xe, Ue = Xd.clone(), Ud.clone()  # state after 51 eager iterations
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on the capture stream (allocations of the stacked search happen here, not under capture)
        solver.iterate(Xd, Ud, xmd, pd, pc, pi)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        solver.iterate(Xd, Ud, xmd, pd, pc, pi)
    replay = run(graph.replay)
    out[batch] = {"eager_ms": eager, "graph_ms": replay, "same_iterates": bool(torch.equal(Xd, Xe) and torch.equal(Ud, Ue))}
print(json.dumps(out))
