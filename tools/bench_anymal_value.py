#!/usr/bin/env python3
"""Value-only evaluation of the ANYmal node (forward_zero; what the SQP's stacked line search launches 14 x batch x N times per iteration):
lane-per-leg value program vs the lane-per-node body (UNGAR_AMD_ANYMAL_VALUE_LANE_PER_NODE=1), unit-fastest and node-major operands."""
import json
import os
import subprocess

MEASUREMENT_LIBRARY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ungar_amd", "lib", "measurement", "libungar_amd.so")  # the A/B switch exists only there
import sys

import torch

sys.path.insert(0, ".")

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import ungar_amd
    from ungar_amd import workloads as W
    from ungar_amd.sharding import unit_fastest
    count = 14 * 4096 * 20
    m = ungar_amd.NodeModel("anymal")
    x, u, _, p = W.synth_device_inputs("anymal", count, 5, torch)
    f = unit_fastest(m.nx, count, torch)
    Op = ungar_amd.Operand
    P = Op.per_instance(p if p.dim() == 1 else p[0], m.np, shared=True)
    xa, ua_, fa = x.t().contiguous(), u.t().contiguous(), torch.empty((count, m.nx), dtype=torch.float64, device="cuda")

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    out = {"nodes": count,
           "unit_fastest_ms": timeit(lambda: m.forward_zero(count, Op.soa(x, x.stride(0)), Op.soa(u, u.stride(0)), None, P, Op.soa(f, f.stride(0)))),
           "node_major_ms": timeit(lambda: m.forward_zero(count, Op.aos(xa, m.nx), Op.aos(ua_, m.nu), None, P, Op.aos(fa, m.nx)))}
    out["layouts_max_abs_difference"] = float((fa.t() - f[:, :count]).abs().max())
    out["finite"] = bool(torch.isfinite(f[:, :count]).all())
    print(json.dumps(out))
else:
    res = {}
    for tag, env in (("lane_per_leg", {}), ("lane_per_node", {"UNGAR_AMD_ANYMAL_VALUE_LANE_PER_NODE": "1"})):
        r = subprocess.run([sys.executable, __file__, "--child"], env={**os.environ, "UNGAR_AMD_LIBRARY": MEASUREMENT_LIBRARY, **env}, capture_output=True, text=True)
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
    print(json.dumps(res))
