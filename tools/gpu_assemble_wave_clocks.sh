#!/usr/bin/env bash
# Section clocks of the one-wavefront assembly kernel (two nodes per launch print their cycles) + SQ counters of the kernel.
source "$(dirname "$0")/use_measurement_build.sh"  # the A/B switches below exist only in the measurement build of the library
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
T=build/batched_quadruped_test
UNGAR_AMD_ASSEMBLE_CLOCKS=1 timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "assemble wave clocks|timing" | head -12
timeout 900 $T /tmp/cg_q 4096 0 2>&1 | grep -E "timing"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pmc_asm
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_asm -o p -- $OLDPWD/$T /tmp/cg_q 4096 0 > /dev/null 2>&1)
  f=$(find /tmp/pmc_asm -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "ShootingAssembleWaveKernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v) / len(v), "(per launch, %d launches)" % len(v))
PY
done
