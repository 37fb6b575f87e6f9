#!/usr/bin/env bash
# Per-kernel split of one batched SQP iteration (quadrotor OCP, 4096 instances) with rocprofv3.
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/sqp_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sqp_prof -o sqp -- python tools/bench_sqp.py 4096 2>/dev/null | grep "^{" | tail -1 | tee gpurun_out/sqp_bench.log
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/sqp_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{float(r["Percentage"]):6.2f}%  calls {r["Calls"]:>6}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
