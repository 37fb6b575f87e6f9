#!/usr/bin/env bash
# Round 6, item 1(a): why do the result stores of the headline kernel stop at ~5 TB/s?  TCC / TCP / TA counters (separate --pmc passes, 4 TCC counters each)
# on the store-only kernels of tools/store_ceiling_tiles.hip -- the product's present pattern (quad8_nt), the wave-native tiles (wave8 / wave16) and the
# column tiles (col16) -- and on the product kernel itself (bench.py, shipped library).  Output: gpurun_out/store_tcc.log (one table).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PASSES=(
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum"
  "TCC_REQ_sum TCC_STREAMING_REQ_sum TCC_WRITE_SECTORS_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"
  "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum"
  "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_CYCLE_sum"
)
VARIANTS=${STORE_TCC_VARIANTS:-"quad8_nt wave8_nt wave16_nt wave16_wb wave16swap_nt col16_nt"}
rm -rf $R/gpurun_out/store_tcc; mkdir -p $R/gpurun_out/store_tcc
p=0
for pass in "${PASSES[@]}"; do
  p=$((p + 1))
  for v in $VARIANTS; do
    timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/store_tcc/p${p}_$v -o a -- $R/tools/_bin/store_ceiling_tiles $v 5 > $R/gpurun_out/store_tcc/p${p}_$v.log 2>&1 || echo "pass $p variant $v failed: $(tail -1 $R/gpurun_out/store_tcc/p${p}_$v.log)"
  done
  if [ "${STORE_TCC_PRODUCT:-1}" = 1 ]; then
    (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/store_tcc/p${p}_product -o a -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub-results --prewarm-seconds 0 > $R/gpurun_out/store_tcc/p${p}_product.log 2>&1) || echo "pass $p product failed"
  fi
done
cd $R
python3 - <<'PY' | tee gpurun_out/store_tcc.log
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/store_tcc/**/*counter_collection.csv", recursive=True):
    variant = f.split("/")[2].split("_", 1)[1]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if variant == "product" and "QuadNodeKernel" not in k: continue
        if variant != "product" and not any(s in k for s in ("Wave8", "Wave16", "Col16", "Quad8")): continue
        acc[variant][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
variants = sorted(acc)
print("per launch (mean over the profiled launches); 81 920 nodes x 1813 entries x 8 B = 1.188 GB written per launch")
print("%-44s" % "counter" + "".join("%16s" % v for v in variants))
for c in names:
    print("%-44s" % c + "".join("%16.4g" % (sum(acc[v][c]) / len(acc[v][c])) if acc[v].get(c) else "%16s" % "-" for v in variants))
PY
