#!/usr/bin/env bash
# GPU box: parity suite of the node kernels + one bench line per workload.
set -uo pipefail
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for w in anymal quadrotor rc_car srbd; do timeout 300 python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_all_$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], 'value=%.3e frac=%.3f kernel_ms=%.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; done
timeout 300 python bench.py --jacobian sparse --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], 'value=%.3e frac=%.3f kernel_ms=%.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))"
