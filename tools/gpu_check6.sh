#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity anymal"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anymal or structured" 2>&1 | tail -6 | tee gpurun_out/pytest_gpu6.log
bash tools/gpu_bench_variants.sh anymal anymal_reg
