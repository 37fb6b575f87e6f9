#!/usr/bin/env bash
# GPU box: rates (tools/bench_rbd_nodes.py) and parity (tests/test_rbd_nodes.py) of every build/variants/lib_*.so
# made by tools/make_rbd_variants.sh.  Output: gpurun_out/rbdv/<name>.json + one summary line per library.
mkdir -p gpurun_out/rbdv
for lib in build/variants/lib_*.so; do
  n=$(basename $lib .so)
  UNGAR_AMD_LIBRARY=$PWD/$lib timeout 300 python tools/bench_rbd_nodes.py ${RBD_VARIANT_NODES:-32768} > gpurun_out/rbdv/$n.json 2> gpurun_out/rbdv/$n.err
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/rbdv/{n}.json"))
    print(n, {k: round(v.get("jacobian_ms", 0), 4) for k, v in d.items() if isinstance(v, dict)})
except Exception as e:  # noqa: BLE001
    print(n, "ERR", e)
PY
  UNGAR_AMD_LIBRARY=$PWD/$lib timeout 600 python -m pytest tests/test_rbd_nodes.py -m gpu -x -q 2>&1 | tail -1
done
