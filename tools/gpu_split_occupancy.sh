#!/usr/bin/env bash
# The fused lane-per-leg kernel and the split (producer / consumer) kernel side by side: launch resources from the kernel trace (registers, scratch, LDS,
# workgroup) and the SQ wave counters -- resident wavefronts per SIMD = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES summed over the device.   build/variants/quad_split_bench
# comes from tools/make_split_bench.sh (built in the container).   Output: gpurun_out/split_occupancy.log
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
exe=$PWD/build/variants/quad_split_bench
out=$PWD/gpurun_out
[ -x "$exe" ] || { echo "missing $exe: run tools/make_split_bench.sh"; exit 1; }
rm -rf $out/split_trace $out/split_sq
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/split_trace -o s -- $exe occ 81920 > $out/split_trace.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/split_sq -o s -- $exe occ 81920 > $out/split_sq.log 2>&1)
python - <<'PY' | tee gpurun_out/split_occupancy.log
import csv, glob, collections
trace = glob.glob("gpurun_out/split_trace/**/s_kernel_trace.csv", recursive=True)
seen = {}
dur = collections.defaultdict(list)
for f in trace:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        seen[n] = r
print("# launch resources (rocprofv3 --kernel-trace) and median duration per 81 920 nodes")
for n, r in seen.items():
    d = sorted(dur[n])
    print(f"{n[:120]}\n    workgroup {r['Workgroup_Size_X']}  arch VGPRs {r['VGPR_Count']}  accum VGPRs {r['Accum_VGPR_Count']}  SGPRs {r['SGPR_Count']}  scratch {r['Scratch_Size']} B/lane  LDS {r['LDS_Block_Size']} B/workgroup  launches {len(d)}  median {d[len(d)//2]:.4f} ms")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/split_sq/**/s_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# SQ counters per launch (mean); resident wavefronts per SIMD = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES (both summed over the device's SIMDs x 4 cycles granularity cancels)")
for n, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    ratio = m.get("SQ_WAVE_CYCLES", 0) / m["SQ_BUSY_CYCLES"] if m.get("SQ_BUSY_CYCLES") else float("nan")
    print(f"{n[:120]}\n    " + "  ".join(f"{k} {v:.4g}" for k, v in sorted(m.items())) + f"\n    SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = {ratio:.3f}")
PY
