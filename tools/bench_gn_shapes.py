#!/usr/bin/env python3
"""Gauss-Newton contraction for the block shapes of the built-in models: lane-per-node kernel vs lane-per-(node, block) kernel."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd.sharding import unit_fastest  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


out = {}
t_end = time.perf_counter() + 0.5
warm = torch.rand((1 << 24,), device="cuda", dtype=torch.float64)
while time.perf_counter() < t_end:
    warm.mul_(1.0000001)
    torch.cuda.synchronize()
for name, rows, cols, count in (("anymal 37x49", 37, 49, 81920), ("srbd 13x37", 13, 37, 30 * 4096), ("quadrotor 13x17", 13, 17, 128 * 4096), ("rc_car 6x8", 6, 8, 200 * 16384),
                                ("srbd_ineq 24x37", 24, 37, 30 * 4096)):
    J, d, G = unit_fastest(rows * cols, count, torch), unit_fastest(rows, count, torch), unit_fastest(cols * cols, count, torch)
    J.copy_(torch.rand((rows * cols, count), device="cuda", dtype=torch.float64))
    d.copy_(torch.rand((rows, count), device="cuda", dtype=torch.float64))
    alg = count * 8 * (rows * cols + rows + cols * (cols + 1) // 2)
    r = {}
    for k in ("gn_hessian_lanes", "gn_hessian_tiles"):
        ms = timeit(lambda: getattr(ungar_amd, k)(J, d, G, rows, cols, count, True))
        r[k] = {"ms": ms, "frac_of_8TBs": alg / ms / 1e6 / 8000}
    out[name] = r
    del J, d, G
print(json.dumps(out))
