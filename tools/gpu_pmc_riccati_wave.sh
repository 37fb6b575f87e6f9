#!/usr/bin/env bash
# Counters of the register-resident Riccati kernels (ocp_riccati_wave.hip): wave cycles, issue, waits, matrix-core busy, HBM traffic.
#   usage: tools/gpu_pmc_riccati_wave.sh [sizes, e.g. 37x12]      (outputs under gpurun_out/)
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
sizes=${*:-37x12}
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
             "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"; do
  i=$((i + 1))
  rm -rf gpurun_out/rw_pmc$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OLDPWD/gpurun_out/rw_pmc$i -o rw -- python $OLDPWD/tools/check_riccati_wave.py 4096 $sizes > /dev/null 2>&1) || echo "pass $i failed: $group"
done
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/rw_pmc*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "Riccati" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/riccati_wave_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
