#!/usr/bin/env python3
"""Batched Riccati solve alone, for the block sizes of the reference's OCPs: time per launch at 4096 instances (HIP events) and, against a
diagnostic library (tools/make_riccati_clocks.sh, UNGAR_AMD_LIBRARY=...), the share of every phase of a knot.
  sizes: 37+12 N=20 (full-body quadruped), 25+24 N=30 (reference quadruped, feet carried, rows eliminated), 13+24 N=30, 17+4 N=30
  (reference quadrotor, inputs carried), 13+4 N=30, 8+2 N=30 (reference RC car, inputs carried), 6+2 N=30."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd.sqp import riccati_solve  # noqa: E402

PHASES = ("-", "operands", "P[A|B]", "H", "factor+gains", "cost-to-go", "-", "forward pass")
SIZES = [(37, 12, 20), (25, 24, 30), (13, 24, 30), (17, 4, 30), (13, 4, 30), (8, 2, 30), (6, 2, 30),
         (10, 3, 30), (20, 9, 30), (31, 30, 20)]  # the last three: sizes the library is not compiled for -- register-resident kernels from the kernel factory
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
only = sys.argv[2:]  # e.g. 37x12; a size that is not in the list (e.g. 12x5) is added with N = 30
for item in only:
    nx_, nu_ = (int(v) for v in item.split("x"))
    if not any((nx_, nu_) == (a, b_) for a, b_, _ in SIZES):
        SIZES.append((nx_, nu_, 30))
lib = ungar_amd.load_library()
clocks = hasattr(lib, "ungar_amd_debug_riccati_clocks")
buf = (ctypes.c_ulonglong * 8)()
g = torch.Generator(device="cuda").manual_seed(1)
for nx, nu, N in SIZES:
    if only and f"{nx}x{nu}" not in only:
        continue
    n = nx + nu
    r = lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=torch.float64)  # noqa: E731
    AB = 0.3 * r(batch, N, nx, n)
    AB[:, :, :, :nx] += torch.eye(nx, device="cuda", dtype=torch.float64)
    L = r(batch, N, n, n)
    W = torch.triu(0.1 * L @ L.transpose(-1, -2)).contiguous()
    LN = r(batch, nx, nx)
    WN = torch.triu(LN @ LN.transpose(-1, -2)).contiguous()
    b, w, wN, dx0 = 0.1 * r(batch, N, nx), r(batch, N, n), r(batch, nx), r(batch, nx)
    for _ in range(3):
        out = riccati_solve(nx, nu, N, batch, AB, b, W, w, dx0, WN, wN)
    torch.cuda.synchronize()
    if clocks:
        lib.ungar_amd_debug_riccati_clocks(buf)  # clear
    times = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = riccati_solve(nx, nu, N, batch, AB, b, W, w, dx0, WN, wN)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    assert int(out[2].abs().max()) == 0 or os.environ.get("UNGAR_BENCH_NO_STATUS_CHECK")  # (timing-only library variants compute garbage)
    lib.ungar_ocp_riccati_route.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int32]
    route = {0: "LDS-resident", 1: "register-resident (compiled in)", 2: "register-resident (kernel factory)"}[lib.ungar_ocp_riccati_route(nx, nu, 0, 0)]
    line = {"nx": nx, "nu": nu, "N": N, "batch": batch, "route": route, "ms_median": sorted(times)[len(times) // 2], "ms_min": min(times)}
    flops = N * (2 * nx * nx * n + 2 * nx * n * (n + 1) / 2 + nu ** 3 / 3 + 2 * nu * nu * (nx + 1) + 2 * nx * nx * nu)  # per instance: P[A|B], H (upper), Cholesky, solves, cost-to-go
    byts = 8 * N * (nx * n + n * (n + 1) / 2 + n + nx + 2 * nu * (nx + 1) + 2 * n)  # operands once, gains out and back, steps out
    line["GFLOP_per_s"] = batch * flops / (line["ms_min"] * 1e-3) / 1e9
    line["HBM_GB_per_s"] = batch * byts / (line["ms_min"] * 1e-3) / 1e9
    line["frac_of_roofline"] = max(line["GFLOP_per_s"] / 78600.0, line["HBM_GB_per_s"] / 8000.0)  # FP64 vector peak 78.6 TFLOP/s (spec), HBM 8 TB/s
    if clocks:
        lib.ungar_amd_debug_riccati_clocks(buf)
        total = sum(buf) or 1
        line["phase_share"] = {PHASES[i]: round(buf[i] / total, 3) for i in range(8) if buf[i]}
        line["ticks_per_knot"] = {PHASES[i]: int(buf[i] / 10 / N) for i in range(8) if buf[i]}
    print(json.dumps(line), flush=True)
