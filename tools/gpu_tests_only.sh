#!/usr/bin/env bash
# The GPU parity suite alone (what the driver runs at round end).
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
