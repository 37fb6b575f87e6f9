#!/usr/bin/env bash
# Matrix-core counters (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES next to the busy cycles of the launch) of the kernels that use v_mfma_f64_16x16x4_f64:
# the Gauss-Newton contraction of BASELINE config 4 and the register-resident Riccati kernels.  Writes gpurun_out/mfma_counters.json (copy to profiles/).
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
here=$(pwd)
i=0
for group in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i + 1))
  rm -rf gpurun_out/mfma_pmc_gn$i gpurun_out/mfma_pmc_ric$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $here/gpurun_out/mfma_pmc_gn$i -o m -- python $here/tools/run_gn_mfma.py > /dev/null 2>&1) || echo "gn pass $i failed"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $here/gpurun_out/mfma_pmc_ric$i -o m -- python $here/tools/check_riccati_wave.py 4096 37x12 25x24 13x24 > /dev/null 2>&1) || echo "riccati pass $i failed"
done
python - <<'PY'
import csv, glob, collections, json
out = {}
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/mfma_pmc_*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob("gpurun_out/mfma_pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        key = "gn_hessian_unit_fastest" if "GnHessianSoa" in k or "gn_hessian" in k.lower() or "GnHessian" in k else ("riccati_wave_" + k.split("RiccatiWaveKernel<")[1].split(",")[0] + "+" + k.split("RiccatiWaveKernel<")[1].split(",")[1].strip() if "RiccatiWaveKernel<" in k else None)
        if key is None:
            continue
        out.setdefault(key, {"kernel": k[:140]})[c] = sum(v) / len(v)
        if dur[k]:
            out[key]["duration_us"] = sorted(dur[k])[len(dur[k]) // 2] / 1e3
for key, e in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "duration_us" in e:
        # busy cycles are summed over the 1024 SIMDs of the device; a v_mfma_f64_16x16x4_f64 keeps its SIMD's matrix pipe busy for 64 cycles (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA)
        e["cycles_per_mfma"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, e.get("SQ_INSTS_MFMA", 1.0))
        if "GRBM_GUI_ACTIVE" in e:
            e["mfma_busy_fraction_of_all_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e["useful_TFLOPs_from_counter"] = e.get("SQ_INSTS_MFMA", 0.0) * 2048.0 / e["duration_us"] / 1e6
json.dump(out, open("gpurun_out/mfma_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
