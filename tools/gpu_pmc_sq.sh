#!/usr/bin/env bash
# SQ counters for one kernel variant: where do the wave cycles go?
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
mdl=${1:-anymal}
rm -rf gpurun_out/sq1 gpurun_out/sq2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/sq1 -o a -- python bench.py --model $mdl --steps 3 --warmup 1 --no-cpu-baseline --no-sub-results > gpurun_out/sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/sq2 -o a -- python bench.py --model $mdl --steps 3 --warmup 1 --no-cpu-baseline --no-sub-results > gpurun_out/sq2.log 2>&1
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/sq1","gpurun_out/sq2"):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ungar_amd" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, sum(v)/len(v))
PY
