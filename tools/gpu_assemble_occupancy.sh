#!/usr/bin/env bash
# The one-wavefront assembly kernel at three nodes per SIMD (one LDS region, tiles of W_e fetched one at a time) against the previous tree's kernel
# (build/variants/shooting_old/libungar_amd.so, if present: two per SIMD): facade comparison, wall clock per SQP iteration of 4096 quadruped instances,
# section clocks, kernel split.  Outputs under gpurun_out/.
source "$(dirname "$0")/use_measurement_build.sh"
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=build/batched_quadruped_test
timeout 900 $T /tmp/cg_q 1024 8 2>&1 | grep -E "^iteration [12]:|PASS|FAIL|EXCEPTION"
for rep in 1 2 3; do timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "timing" | sed "s/^/[three per SIMD] /"; done
if [ -f build/variants/shooting_old/libungar_amd.so ]; then
  for rep in 1 2 3; do LD_LIBRARY_PATH=build/variants/shooting_old:${LD_LIBRARY_PATH:-} UNGAR_AMD_LIBRARY=build/variants/shooting_old/libungar_amd.so timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "timing" | sed "s/^/[previous kernel] /"; done
fi
UNGAR_AMD_ASSEMBLE_CLOCKS=1 timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "assemble wave clocks" | head -4
# kernel durations: three profiled runs of each library (a run's average over its 7 launches moves by 5 % from run to run)
profile() {  # label, library directory ("" = the tree's)
  for rep in 1 2 3; do
    rm -rf gpurun_out/bprof_asm
    (cd /tmp && if [ -n "$2" ]; then export LD_LIBRARY_PATH=$OLDPWD/$2:${LD_LIBRARY_PATH:-} UNGAR_AMD_LIBRARY=$OLDPWD/$2/libungar_amd.so; fi; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/bprof_asm -o b -- $OLDPWD/$T /tmp/cg_q 4096 0 > /dev/null 2>&1)
    f=$(find gpurun_out/bprof_asm -name "b_kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep ShootingAssembleWaveKernel "$f" | awk -F, -v l="$1" -v r=$rep '{gsub(/"/, "", $0); printf "[%s] profiled run %d: ShootingAssembleWaveKernel %d launches, mean %.1f us\n", l, r, $(NF-6), $(NF-4) / 1000}'
    [ -n "$f" ] && [ -z "$2" ] && [ $rep = 3 ] && cp "$f" gpurun_out/assemble_occupancy_kernel_stats.csv && head -6 "$f" | cut -c1-200
  done
  rm -rf gpurun_out/bprof_asm
}
profile "three per SIMD" ""
[ -f build/variants/shooting_old/libungar_amd.so ] && profile "previous kernel" build/variants/shooting_old
for v in wave_eu2; do  # tools/make_shooting_variants.sh: this code held at two nodes per SIMD
  [ -f build/variants/shooting_$v/libungar_amd.so ] && profile "variant $v" build/variants/shooting_$v
done
