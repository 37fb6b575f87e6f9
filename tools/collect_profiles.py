#!/usr/bin/env python3
"""Copies the judged summaries out of gpurun_out/ (scratch) into profiles/ (tracked):
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (all kernels)
  profiles/<tag>_kernel_stats_warm.json  the same trace's durations of the most-launched ungar_amd kernel WITHOUT its first launches (cold clocks,
                                    first-touch page faults): the figure bench.py's roofline line is to be compared with
  profiles/<tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per launch of the ungar_amd kernels
  profiles/traffic.json             {"<workload>:<batch>": HBM bytes per launch}  read by bench.py
FETCH_SIZE / WRITE_SIZE are reported in KiB (hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024,
cdna_hip_programming.md section 7); per MI355X_MICROARCH.md §HBM the gfx950 FETCH_SIZE under-counts wide
coalesced streaming reads by exactly 2x, so the read side is doubled ("corrected"); both raw and
corrected figures are stored.
usage: collect_profiles.py <tag> <workload:batch> [stats_dir] [pmc_fetch_dir] [pmc_write_dir]
"""
import csv
import json
import os
import shutil
import sys

tag, key = sys.argv[1], sys.argv[2]
stats_dir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/prof"
fetch_dir = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/pmc1"
write_dir = sys.argv[5] if len(sys.argv) > 5 else "gpurun_out/pmc2"
os.makedirs("profiles", exist_ok=True)
for f in os.listdir(stats_dir):
    if f.endswith("kernel_stats.csv"):
        shutil.copy(os.path.join(stats_dir, f), f"profiles/{tag}_kernel_stats.csv")


def warm_stats(d, skip_seconds=0.5, skip_launches=20):
    """Durations from the kernel trace, most-launched ungar_amd kernel, after dropping what bench.py itself does not time: the launches of its pre-warm
    (0.5 s) and warm-up.  (`--stats` averages over ALL launches: r03p's 260.6 us with a 1077 us maximum mixed cold launches into the figure.)"""
    import statistics
    rows = {}
    for f in os.listdir(d):
        if not f.endswith("kernel_trace.csv"):
            continue
        for r in csv.DictReader(open(os.path.join(d, f))):
            if "ungar_amd" in r["Kernel_Name"]:
                rows.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not rows:
        return None
    name, launches = max(rows.items(), key=lambda kv: len(kv[1]))
    launches.sort()
    t0 = launches[0][0]
    warm = [(e - s) / 1e3 for i, (s, e) in enumerate(launches) if i >= skip_launches and s - t0 >= skip_seconds * 1e9]
    if len(warm) < 10:
        warm = [(e - s) / 1e3 for s, e in launches[len(launches) // 2:]]
    allus = [(e - s) / 1e3 for s, e in launches]
    return {"kernel": name, "launches": len(launches), "warm_launches": len(warm), "dropped": f"first {skip_launches} launches and everything within {skip_seconds} s of the first launch",
            "warm_us": {"mean": statistics.fmean(warm), "median": statistics.median(warm), "stdev": statistics.pstdev(warm), "min": min(warm), "max": max(warm)},
            "all_us": {"mean": statistics.fmean(allus), "max": max(allus)}}


w = warm_stats(stats_dir)
if w:
    json.dump(w, open(f"profiles/{tag}_kernel_stats_warm.json", "w"), indent=1)
    print("warm launches:", json.dumps(w["warm_us"]))


def mean_counter(d, counter):
    vals = {}
    for f in os.listdir(d):
        if not f.endswith("counter_collection.csv"):
            continue
        for r in csv.DictReader(open(os.path.join(d, f))):
            if r["Counter_Name"] == counter and "ungar_amd" in r["Kernel_Name"]:
                vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


fetch, write = mean_counter(fetch_dir, "FETCH_SIZE"), mean_counter(write_dir, "WRITE_SIZE")
out = {}
for k in fetch:
    raw = (fetch[k] + write.get(k, 0.0)) * 1024
    corrected = (2 * fetch[k] + write.get(k, 0.0)) * 1024
    out[k] = {"FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0), "hbm_bytes_raw": raw, "hbm_bytes_fetch_x2": corrected}
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
tpath = "profiles/traffic.json"
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
dense = [v for k, v in out.items() if ", 2, " in k or "Jacobian" in k or True]
if dense:
    traffic[key] = max(v["hbm_bytes_fetch_x2"] for v in dense)
    traffic["_source"] = (f"profiles/{tag}_pmc.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --no-cpu-baseline "
                          f"--no-sub-results`, run {tag}; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (gfx950 correction of MI355X_MICROARCH.md); "
                          "NOT measured in the run that printed this line")
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps(out, indent=1)[:600])
