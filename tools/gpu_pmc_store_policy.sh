#!/usr/bin/env bash
# Counters of the fused lane-per-leg kernel with write-back vs non-temporal result stores (tools/quad_split_bench.hip variants):
# why do write-back stores, faster on their own (tools/store_ceiling.hip), slow the real kernel down?
set -uo pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { tag=$1; shift
  rm -rf $R/gpurun_out/$tag
  QUAD_BENCH_QUICK=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$tag -o a -- $R/build/variants/split2 pmc 81920 "fused" > $R/gpurun_out/$tag.log 2>&1 || tail -3 $R/gpurun_out/$tag.log
}
run sp1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sp2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VALU
run sp3 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_WRITEBACK_sum
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/sp1","gpurun_out/sp2","gpurun_out/sp3"):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "QuadNodeKernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(k[0], k[1], sum(v)/len(v), len(v))
PY
