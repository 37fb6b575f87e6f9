set -uo pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/bench_gn_lanes.py 2>&1 | tail -1 | tee gpurun_out/gn_lanes_lds.json
timeout 1200 python -m pytest tests/test_ocp_sqp.py tests/test_gpu_parity.py -m gpu -q -x -k "sqp or gn_hessian or chain" 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_partial.log
cat gpurun_out/sqp_quadrotor_timing.json
