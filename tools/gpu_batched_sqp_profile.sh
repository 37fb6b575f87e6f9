#!/usr/bin/env bash
# Wall clock and rocprofv3 kernel split of the batched SQP iteration on the reference's OCPs as written (C++ driver, 4096 instances).
# Outputs under gpurun_out/; copy the summaries into profiles/<tag>_batched_{quadrotor,quadruped}_*.
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
B=${1:-4096}
for p in quadrotor rc_car quadruped; do
  # first run fills the code-object cache (compiles are not part of the iteration)
  timeout 900 build/batched_${p}_test /tmp/cg_$p $B 0 > gpurun_out/batched_${p}_timing.log 2>&1
  grep -E "timing|PASS|FAIL" gpurun_out/batched_${p}_timing.log
  rm -rf gpurun_out/bprof_$p
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/bprof_$p -o b -- $OLDPWD/build/batched_${p}_test /tmp/cg_$p $B 0 > /dev/null 2>&1)
  f=$(find gpurun_out/bprof_$p -name "b_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/batched_${p}_kernel_stats.csv && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{float(r['Percentage']):6.2f}%  calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
done
