#!/usr/bin/env bash
# One-wavefront shooting assembly kernel (default for quadruped-shaped nodes) against the workgroup kernel (UNGAR_AMD_ASSEMBLE_VARIANT=workgroup):
# facade comparison, dumped QP data (tolerance: tests/test_batched_sqp.py::test_assembly_kernels_agree), wall clock at 4096 instances, kernel split.
source "$(dirname "$0")/use_measurement_build.sh"  # the A/B switches below exist only in the measurement build of the library
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=build/batched_quadruped_test
timeout 900 $T /tmp/cg_q 1024 8 > gpurun_out/assemble_wave_facade.log 2>&1; echo "wavefront rc $?"
grep -E "^iteration [12]:|PASS|FAIL|EXCEPTION" gpurun_out/assemble_wave_facade.log
for mode in wavefront workgroup; do
  for rep in 1 2; do
    if [ $mode = workgroup ]; then export UNGAR_AMD_ASSEMBLE_VARIANT=workgroup; else unset UNGAR_AMD_ASSEMBLE_VARIANT; fi
    timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "timing" | sed "s/^/[$mode] /"
  done
done
unset UNGAR_AMD_ASSEMBLE_VARIANT
rm -rf gpurun_out/bprof_asm
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/bprof_asm -o b -- $OLDPWD/$T /tmp/cg_q 4096 0 > /dev/null 2>&1)
f=$(find gpurun_out/bprof_asm -name "b_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/assemble_wave_kernel_stats.csv && head -8 "$f" | cut -c1-220
rm -rf gpurun_out/bprof_asm
timeout 1500 python -m pytest tests/test_batched_sqp.py -m gpu -x -q 2>&1 | tail -5
