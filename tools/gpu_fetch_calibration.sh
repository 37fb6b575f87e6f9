#!/usr/bin/env bash
# FETCH_SIZE / TCC request counters of tools/fetch_calibration.hip (known byte counts) -> gpurun_out/fetch_calibration.log
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
run() { tag=$1; shift; rm -rf gpurun_out/$tag; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$tag -o c -- $R/tools/_bin/fetch_calibration > $R/gpurun_out/$tag.log 2>&1) || tail -3 gpurun_out/$tag.log; }
run fc1 FETCH_SIZE
run fc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run fc3 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run fc4 WRITE_SIZE
run fc5 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python3 - <<'PY' | tee gpurun_out/fetch_calibration.log
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/fc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "calib_" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
known = 2 << 30
print("kernel                 counter                     mean value      x known bytes (2 GiB; FETCH_SIZE is in KiB)")
for (k, c), v in sorted(acc.items()):
    m = sum(v) / len(v)
    byts = m * 1024 if c in ("FETCH_SIZE", "WRITE_SIZE") else m * 64 if "WRREQ" in c else m * 64 if "RDREQ_sum" in c and "32B" not in c else m * 32 if "32B" in c else m * 128
    print(f"{k:22s} {c:26s} {m:16.1f}   -> {byts / known:6.3f}")
PY
