#!/usr/bin/env bash
# Quick loop: facade comparison of the three batched OCPs (1024 instances, 4 compared), timing at 4096 instances, assembly section clocks.
source "$(dirname "$0")/use_measurement_build.sh"  # the A/B switches below exist only in the measurement build of the library
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in quadruped quadrotor rc_car; do
  timeout 900 build/batched_${t}_test /tmp/cg_$t 1024 4 2>&1 | grep -E "^iteration [12]:|PASS|FAIL|EXCEPTION"
  for rep in 1 2; do timeout 300 build/batched_${t}_test /tmp/cg_$t 4096 0 2>&1 | grep -E "timing" | sed "s/^/[$t] /"; done
done
UNGAR_AMD_ASSEMBLE_CLOCKS=1 timeout 900 build/batched_quadruped_test /tmp/cg_quadruped 4096 0 2>&1 | grep -E "assemble wave clocks" | head -6
rm -rf gpurun_out/bprof_asm
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/bprof_asm -o b -- $OLDPWD/build/batched_quadruped_test /tmp/cg_quadruped 4096 0 > /dev/null 2>&1)
f=$(find gpurun_out/bprof_asm -name "b_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/quick_quadruped_kernel_stats.csv && head -10 "$f" | cut -c1-200
rm -rf gpurun_out/bprof_asm
