#!/usr/bin/env bash
# A/B of the shooting assembly kernels on the GPU box: the WORKGROUP kernel's wavefront-specialised sections (UNGAR_AMD_ASSEMBLE_VARIANT=workgroup) against
# its generic sections (UNGAR_AMD_ASSEMBLE_GENERIC=1) -- same bits expected: the facade comparison lines and the dumped QP data of both runs are compared --
# then wall clock at 4096 instances of these two and of the default route (the one-wavefront kernel of DESIGN 4.12; its agreement with the others within
# rounding is tests/test_batched_sqp.py::test_assembly_kernels_agree), the kernel split (rocprofv3) and the section clocks.  Outputs under gpurun_out/.
source "$(dirname "$0")/use_measurement_build.sh"  # the A/B switches below exist only in the measurement build of the library
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=build/batched_quadruped_test
rm -rf /tmp/dumpA /tmp/dumpB; mkdir -p /tmp/dumpA /tmp/dumpB
UNGAR_AMD_ASSEMBLE_VARIANT=workgroup timeout 900 $T /tmp/cg_q 1024 8 /tmp/dumpA > gpurun_out/assemble_ab_specialised.log 2>&1; echo "specialised rc $?"
UNGAR_AMD_ASSEMBLE_GENERIC=1 timeout 900 $T /tmp/cg_q 1024 8 /tmp/dumpB > gpurun_out/assemble_ab_generic.log 2>&1; echo "generic rc $?"
grep -E "^iteration [12]:|PASS|FAIL" gpurun_out/assemble_ab_specialised.log
if diff <(grep -E "^iteration" gpurun_out/assemble_ab_specialised.log) <(grep -E "^iteration" gpurun_out/assemble_ab_generic.log) > gpurun_out/assemble_ab_diff.log; then echo "facade comparison lines: identical"; else echo "facade comparison lines DIFFER"; head -5 gpurun_out/assemble_ab_diff.log; fi
same=0; differ=0
for f in /tmp/dumpA/*; do if cmp -s "$f" "/tmp/dumpB/$(basename $f)"; then same=$((same+1)); else differ=$((differ+1)); echo "differs: $(basename $f)"; fi; done 2>/dev/null | head -5
echo "dumped files compared: $(ls /tmp/dumpA | wc -l)"
# the fallback of the barrier terms (one lane per entry: patterns whose pairs do not fit a wavefront) is a measurement variant for this problem: checked against the same dumps
if [ -f build/variants/shooting_entry_lanes/libungar_amd.so ]; then
  rm -rf /tmp/dumpC; mkdir -p /tmp/dumpC
  LD_LIBRARY_PATH=build/variants/shooting_entry_lanes:${LD_LIBRARY_PATH:-} timeout 900 $T /tmp/cg_q 1024 8 /tmp/dumpC > gpurun_out/assemble_ab_entry_lanes.log 2>&1; echo "entry_lanes variant rc $?"
  bad=0; for f in /tmp/dumpB/*; do cmp -s "$f" "/tmp/dumpC/$(basename $f)" || bad=$((bad+1)); done; echo "entry_lanes variant: $bad of $(ls /tmp/dumpB | wc -l) dumped files differ from the generic sections"
fi
for mode in wavefront specialised generic; do
  for rep in 1 2; do
    unset UNGAR_AMD_ASSEMBLE_GENERIC UNGAR_AMD_ASSEMBLE_VARIANT
    if [ $mode = generic ]; then export UNGAR_AMD_ASSEMBLE_GENERIC=1; fi
    if [ $mode = specialised ]; then export UNGAR_AMD_ASSEMBLE_VARIANT=workgroup; fi
    timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "timing" | sed "s/^/[$mode] /"
  done
done
unset UNGAR_AMD_ASSEMBLE_GENERIC UNGAR_AMD_ASSEMBLE_VARIANT
UNGAR_AMD_ASSEMBLE_CLOCKS=1 timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "assemble wave clocks" | head -4
for v in eu3_k4 eu4_k4 eu3_k2 entry_lanes; do  # measurement variants (tools/make_shooting_variants.sh)
  if [ -f build/variants/shooting_$v/libungar_amd.so ]; then
    for rep in 1 2; do LD_LIBRARY_PATH=build/variants/shooting_$v:${LD_LIBRARY_PATH:-} timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "timing" | sed "s/^/[variant $v] /"; done
  fi
done
rm -rf gpurun_out/bprof_asm
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/bprof_asm -o b -- $OLDPWD/$T /tmp/cg_q 4096 0 > /dev/null 2>&1)
f=$(find gpurun_out/bprof_asm -name "b_kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/assemble_ab_kernel_stats.csv && head -8 "$f" | cut -c1-200
rm -rf gpurun_out/bprof_asm
if [ -f build/variants/shooting_clocks/libungar_amd.so ]; then
  LD_LIBRARY_PATH=build/variants/shooting_clocks:${LD_LIBRARY_PATH:-} UNGAR_AMD_LIBRARY=build/variants/shooting_clocks/libungar_amd.so timeout 600 $T /tmp/cg_q 1024 0 2>&1 | grep -E "assemble clocks" | head -12 > gpurun_out/assemble_ab_clocks.log
  cat gpurun_out/assemble_ab_clocks.log
fi
