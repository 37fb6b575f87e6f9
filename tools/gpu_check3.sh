#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu (full)"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu3.log
