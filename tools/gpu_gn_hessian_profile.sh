#!/usr/bin/env bash
# GPU box: timing + rocprofv3 kernel stats + MFMA / HBM counters for the Gauss-Newton contraction
# (BASELINE config 3: "Gauss-Newton J^T J on MFMA").  Counters in separate passes.
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_gn_hessian.py | tee gpurun_out/gn_hessian_bench.json
rm -rf gpurun_out/gn_prof gpurun_out/gn_pmc1 gpurun_out/gn_pmc2 gpurun_out/gn_pmc3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gn_prof -o gn -- python tools/bench_gn_hessian.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/gn_pmc1 -o gn -- python tools/bench_gn_hessian.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/gn_pmc2 -o gn -- python tools/bench_gn_hessian.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/gn_pmc3 -o gn -- python tools/bench_gn_hessian.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("gpurun_out/gn_pmc1", "gpurun_out/gn_pmc2", "gpurun_out/gn_pmc3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "GnHessian" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
for f in glob.glob("gpurun_out/gn_prof/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "GnHessian" in r["Name"]:
            out["rocprof_avg_ns"] = float(r["AverageNs"])
json.dump(out, open("gpurun_out/gn_hessian_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
