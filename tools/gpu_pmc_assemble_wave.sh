#!/usr/bin/env bash
# SQ counters of the one-wavefront assembly kernel (the reference's quadruped OCP, 4096 instances, C++ driver), the tree's kernel and -- if
# build/variants/shooting_old/libungar_amd.so exists -- the previous one.  Separate --pmc passes with --kernel-trace only.  Output: gpurun_out/assemble_wave_sq_counters.log
source "$(dirname "$0")/use_measurement_build.sh"
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=$PWD/build/batched_quadruped_test
timeout 600 $T /tmp/cg_q 4096 0 > /dev/null 2>&1  # fills the code-object cache
: > gpurun_out/assemble_wave_sq_counters.log
for lib in tree previous; do
  if [ $lib = previous ]; then
    [ -f build/variants/shooting_old/libungar_amd.so ] || continue
    export LD_LIBRARY_PATH=$PWD/build/variants/shooting_old:${LD_LIBRARY_PATH:-} UNGAR_AMD_LIBRARY=$PWD/build/variants/shooting_old/libungar_amd.so
  fi
  rm -rf gpurun_out/asq1 gpurun_out/asq2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OLDPWD/gpurun_out/asq1 -o a -- $T /tmp/cg_q 4096 0 > /dev/null 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $OLDPWD/gpurun_out/asq2 -o a -- $T /tmp/cg_q 4096 0 > /dev/null 2>&1)
  python - $lib <<'PY' | tee -a gpurun_out/assemble_wave_sq_counters.log
import csv, glob, collections, sys
print(f"== ShootingAssembleWaveKernel<25, 24, 16>, {sys.argv[1]} (per launch of 4096 x 31 nodes, mean over the launches of the run)")
for d in ("gpurun_out/asq1", "gpurun_out/asq2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ShootingAssembleWaveKernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:24s} {sum(v) / len(v):.6g}")
PY
done
rm -rf gpurun_out/asq1 gpurun_out/asq2
