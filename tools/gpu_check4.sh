#!/usr/bin/env bash
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity of anymal variants"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anymal or structured" 2>&1 | tail -6 | tee gpurun_out/pytest_gpu4.log
for mdl in anymal anymal_lds anymal_ad; do
  echo "== bench $mdl"; timeout 300 python bench.py --model $mdl --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], 'evals/s=%.3e'%d['value'], 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'frac=%.4f'%d['roofline']['frac'])" | tee gpurun_out/bench4_$mdl.log
done
