#!/usr/bin/env bash
# The (node, block)-per-lane Gauss-Newton kernel: parity tests, timing for the staging depths, kernel stats.
source "$(dirname "$0")/use_measurement_build.sh"  # the A/B switches below exist only in the measurement build of the library
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gn_hessian or chain" 2>&1 | tail -5
for st in 2 4 8; do UNGAR_GN_TILES_STAGE=$st timeout 300 python tools/bench_gn_lanes.py 2>&1 | tail -1 | tee gpurun_out/gn_tiles_stage$st.json; done
