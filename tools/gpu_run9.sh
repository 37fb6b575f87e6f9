set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in fixed 64; do echo "variant $v"; UNGAR_AMD_RICCATI_VARIANT=$v python tools/bench_sqp.py 4096 2>&1 | tail -1 | cut -c90-200; UNGAR_AMD_RICCATI_VARIANT=$v timeout 900 python -m pytest tests/test_ocp_sqp.py -m gpu -q -x 2>&1 | tail -1; done
