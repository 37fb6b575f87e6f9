#!/usr/bin/env bash
# SQ counters of the shooting assembly kernel (the reference's quadruped OCP, 4096 instances, C++ driver): wavefront-specialised sections and, with
# UNGAR_AMD_ASSEMBLE_GENERIC=1, the generic ones.  Separate --pmc passes with --kernel-trace only.  Output: gpurun_out/assemble_sq_counters.log
source "$(dirname "$0")/use_measurement_build.sh"  # the A/B switches below exist only in the measurement build of the library
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=$PWD/build/batched_quadruped_test
timeout 600 $T /tmp/cg_q 4096 0 > /dev/null 2>&1  # fills the code-object cache
: > gpurun_out/assemble_sq_counters.log
for mode in specialised generic; do
  if [ $mode = generic ]; then export UNGAR_AMD_ASSEMBLE_GENERIC=1; else unset UNGAR_AMD_ASSEMBLE_GENERIC; fi
  rm -rf gpurun_out/asq1 gpurun_out/asq2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OLDPWD/gpurun_out/asq1 -o a -- $T /tmp/cg_q 4096 0 > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OLDPWD/gpurun_out/asq2 -o a -- $T /tmp/cg_q 4096 0 > /dev/null 2>&1)
  python - $mode <<'PY' | tee -a gpurun_out/assemble_sq_counters.log
import csv, glob, collections, sys
print(f"== ShootingAssembleKernel, {sys.argv[1]} sections (per launch of 4096 x 31 nodes, mean over the launches of the run)")
for d in ("gpurun_out/asq1", "gpurun_out/asq2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ShootingAssembleKernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:24s} {sum(v) / len(v):.6g}")
PY
done
unset UNGAR_AMD_ASSEMBLE_GENERIC
rm -rf gpurun_out/asq1 gpurun_out/asq2
