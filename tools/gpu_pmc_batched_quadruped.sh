#!/usr/bin/env bash
# HBM traffic and issue counters of the kernels of one batched SQP iteration of the reference's quadruped OCP (4096 instances): FETCH_SIZE / WRITE_SIZE in
# their own passes (MI355X_MICROARCH.md: KiB, gfx950 FETCH x 2 correction applied in the summary), SQ counters in a third.  Output:
# gpurun_out/batched_quadruped_counters.json  (per kernel, mean per launch).
set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY"; do
  i=$((i + 1))
  rm -rf /tmp/bq_pmc$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $group --output-format csv -d /tmp/bq_pmc$i -o bq -- $OLDPWD/build/batched_quadruped_test /tmp/cg_quadruped 4096 0 > /dev/null 2>&1) || echo "pass $i failed: $group"
done
python3 - <<'PY'
import csv, glob, collections, json, re
out = collections.defaultdict(dict)
for f in glob.glob("/tmp/bq_pmc*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"(\w+Kernel(?:<[^>]*>)?)\(", name)
        short = m.group(1) if m else name[:40]
        acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
        out[k]["launches"] = len(v)
nodes = 4096 * 31
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch_fetch_x2"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        v["hbm_bytes_per_node_fetch_x2"] = v["hbm_bytes_per_launch_fetch_x2"] / nodes
json.dump(out, open("gpurun_out/batched_quadruped_counters.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("hbm_bytes_per_launch_fetch_x2", 0))[:8]:
    v = out[k]
    print(f"{k:50s} launches {v.get('launches')}: fetch {v.get('FETCH_SIZE', 0) / 1024:.1f} MiB (x2: {2 * v.get('FETCH_SIZE', 0) / 1024:.1f}), write {v.get('WRITE_SIZE', 0) / 1024:.1f} MiB, "
          f"{v.get('hbm_bytes_per_node_fetch_x2', 0) / 1024:.1f} KiB per node; VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:.0f} M, SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:.0f} M, MFMA {v.get('SQ_INSTS_MFMA', 0) / 1e6:.1f} M")
PY
