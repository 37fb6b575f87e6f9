#!/usr/bin/env bash
# Instruction-fetch / stall counters for the ANYmal lane-per-leg kernel: is the 100+ KiB straight-line
# body fetch-bound?
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]*IFETCH[A-Z_0-9]*|SQC_[A-Z_0-9]*|SQ_WAIT_[A-Z_0-9]*|SQ_INST_LEVEL[A-Z_0-9]*|SQ_IFETCH[A-Z_0-9]*|SQ_INSTS_[A-Z_0-9]*|SQ_ACTIVE_INST_[A-Z_0-9]*)" | sort -u > gpurun_out/avail_sq.txt
wc -l gpurun_out/avail_sq.txt
run() { # tag counters...
  tag=$1; shift
  rm -rf gpurun_out/$tag
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$tag -o a -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.log 2>&1 || tail -3 gpurun_out/$tag.log
}
run if1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run if2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run if3 SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/if1","gpurun_out/if2","gpurun_out/if3"):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "QuadNodeKernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k, sum(v)/len(v))
PY
