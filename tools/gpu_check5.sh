#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== whole horizon"; timeout 1700 python -m pytest tests/test_whole_horizon.py tests/test_cpp_facade.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu5.log
echo "== dual stream"; timeout 600 python tools/exp_dual_stream.py 2>&1 | tail -9 | tee gpurun_out/dual.log
