#!/usr/bin/env bash
# GPU box: parity of the sparse lane-per-leg kernel + timings of dense / sparse ANYmal output.
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_sparse_quad.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_dense.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --jacobian sparse 2>&1 | tail -1 | tee gpurun_out/bench_sparse.log
