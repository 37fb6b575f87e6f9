#!/usr/bin/env bash
# SQ counters of the batched Riccati kernel (quadrotor OCP, 4096 instances): where do its cycles go?
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/ric_pmc1 gpurun_out/ric_pmc2
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d gpurun_out/ric_pmc1 -o ric -- python ${RICCATI_BENCH:-tools/bench_sqp.py} 4096 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/ric_pmc2 -o ric -- python ${RICCATI_BENCH:-tools/bench_sqp.py} 4096 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("gpurun_out/ric_pmc1", "gpurun_out/ric_pmc2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "RiccatiKernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
json.dump(out, open("gpurun_out/riccati_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
