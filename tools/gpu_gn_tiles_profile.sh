#!/usr/bin/env bash
# GPU box: the config-4 chain (node Jacobians -> Gauss-Newton term, lane-per-(node, block) kernel): timing, rocprofv3 kernel stats,
# HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the contraction kernel.
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/bench_gn_chain.py | tail -1 | tee gpurun_out/gn_chain.json
python tools/bench_gn_chain.py --natural-stride | tail -1 | tee gpurun_out/gn_chain_natural_stride.json
rm -rf gpurun_out/gnt_prof gpurun_out/gnt_pmc1 gpurun_out/gnt_pmc2 gpurun_out/gnt_pmc3
B="python tools/bench_gn_chain.py --reps 10"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gnt_prof -o gnt -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/gnt_pmc1 -o gnt -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/gnt_pmc2 -o gnt -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/gnt_pmc3 -o gnt -- $B > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("gpurun_out/gnt_pmc1", "gpurun_out/gnt_pmc2", "gpurun_out/gnt_pmc3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "GnHessianTiles" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
for f in glob.glob("gpurun_out/gnt_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "GnHessianTiles" in r["Name"]:
            out["rocprof_avg_ns"], out["rocprof_calls"] = float(r["AverageNs"]), int(r["Calls"])
        if "QuadNodeKernel" in r["Name"]:
            out["node_jacobian_rocprof_avg_ns"] = float(r["AverageNs"])
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_bytes_fetch_x2_plus_write"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024  # gfx950 correction of MI355X_MICROARCH.md
    out["algorithmic_bytes"] = 81920 * 8 * (37 * 49 + 37 + 49 * 50 // 2)
json.dump(out, open("gpurun_out/gn_tiles_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
