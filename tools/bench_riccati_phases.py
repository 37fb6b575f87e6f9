#!/usr/bin/env python3
"""Where a knot of the batched Riccati recursion spends its cycles: runs tools/bench_sqp{,_srbd,_anymal}.py against a
diagnostic library built with -DUNGAR_RICCATI_CLOCKS (tools/make_riccati_clocks.sh) and prints the cycles the first lane of
the first workgroup accumulated per phase (s_memtime ticks at 100 MHz), averaged per kernel launch."""
import ctypes
import json
import os
import runpy
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402

torch.zeros(1, device="cuda")  # torch's HIP runtime first

PHASES = ("-", "operands", "P[A|B]", "H", "factor+gains", "cost-to-go", "-", "forward pass")
lib = ungar_amd.load_library()
if not hasattr(lib, "ungar_amd_debug_riccati_clocks"):
    sys.exit("not a diagnostic library: build one with tools/make_riccati_clocks.sh and set UNGAR_AMD_LIBRARY")
buf = (ctypes.c_ulonglong * 8)()
for script in ("tools/bench_sqp.py", "tools/bench_sqp_srbd.py", "tools/bench_sqp_anymal.py"):
    lib.ungar_amd_debug_riccati_clocks(buf)  # clear
    argv, sys.argv = sys.argv, [script, "4096"]
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = argv
    lib.ungar_amd_debug_riccati_clocks(buf)
    total = sum(buf) or 1
    print(json.dumps({"script": os.path.basename(script), "share": {PHASES[i]: round(buf[i] / total, 3) for i in range(8) if buf[i]},
                      "ticks_100MHz": {PHASES[i]: int(buf[i]) for i in range(8) if buf[i]}}), flush=True)
