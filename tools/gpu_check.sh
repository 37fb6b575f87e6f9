#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): parity tests, smoke, bench, rocprofv3 kernel trace.
# Everything lands under gpurun_out/ (merged back); summaries worth keeping are copied to profiles/.
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench.log
for w in quadrotor rc_car srbd; do timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --cpu-seconds 3 2>&1 | tail -1 | tee gpurun_out/bench_$w.log; done
echo "== rocprofv3 kernel trace"
rm -rf gpurun_out/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o anymal -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof -name "*stats*" | head; tail -2 gpurun_out/prof_bench.log
