#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gn_hessian"; timeout 600 python -m pytest tests -m gpu -x -q -k "gn_hessian or errors or sparsity" 2>&1 | tail -15 | tee gpurun_out/pytest_gpu2.log
echo "== rocprofv3 kernel trace"
rm -rf gpurun_out/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o anymal -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof -type f | head; tail -1 gpurun_out/prof_bench.log
echo "== pmc"
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o anymal -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc2 -o anymal -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1
find gpurun_out/pmc1 gpurun_out/pmc2 -type f | head
