#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for mdl in "$@"; do
  timeout 300 python bench.py --model $mdl --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], 'evals/s=%.3e'%d['value'], 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'frac=%.4f'%d['roofline']['frac'])" | tee gpurun_out/benchv_$mdl.log
done
