#!/usr/bin/env bash
# Full round evidence on the GPU box: parity tests, smoke, bench (default line + 2-rank dry run of the config-5 partition), rocprofv3
# kernel trace + HBM counters for the headline kernel, Gauss-Newton and SQP benches.  Outputs under gpurun_out/
# (tools/collect_profiles.py and tools/collect_round.sh copy the judged summaries into profiles/).
# Tests, smoke and every bench line run on the SHIPPED library; only the A/B lines (switches that exist in the measurement build alone) load the other one.
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 3000 python -m pytest tests -m gpu -x -q --durations=30 2>&1 | tail -50 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.log; cut -c1-400 gpurun_out/bench.log
echo "== bench, the driver's command (20 steps, 5 warm-up) with and without the pre-warm"
for p in 0 0.5 0 0.5; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results --prewarm-seconds $p 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prewarm $p: kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']/1e8,3), 'e8 evals/s')"; done | tee gpurun_out/prewarm.log
echo "== bench, 2 ranks sharing the device over gloo: control flow of the config-5 partition (65 536 instances)"
UNGAR_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 2>gpurun_out/bench2.err | tail -1 > gpurun_out/bench_2rank_gloo.log
timeout 600 python bench.py --total-batch 65536 --steps 20 --warmup 3 --no-cpu-baseline --no-sub-results 2>&1 | tail -1 > gpurun_out/bench_config5_1gpu.log
timeout 600 python bench.py --total-batch 8192 --steps 50 --warmup 3 --no-cpu-baseline --no-sub-results 2>&1 | tail -1 > gpurun_out/bench_config5_shard_of_8.log
timeout 300 python bench.py --layout soa --steps 50 --warmup 3 --no-cpu-baseline --no-sub-results 2>&1 | tail -1 > gpurun_out/bench_anymal_unit_fastest.log
timeout 300 python tools/bench_tiles_gather.py 2>/dev/null | tail -1 | tee gpurun_out/tiles_gather.json | cut -c1-500
for w in srbd; do timeout 300 python bench.py --workload $w --steps 50 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$w.log; done
timeout 300 python bench.py --jacobian sparse --steps 50 --warmup 3 --no-cpu-baseline --no-sub-results 2>&1 | tail -1 > gpurun_out/bench_anymal_sparse.log
for m in anymal_reg anymal_ad; do timeout 300 python bench.py --model $m --steps 20 --warmup 3 --no-cpu-baseline --no-sub-results 2>&1 | tail -1 > gpurun_out/bench_$m.log; done
for f in bench_anymal_unit_fastest bench_2rank_gloo bench_config5_1gpu bench_config5_shard_of_8 bench_srbd bench_anymal_sparse bench_anymal_reg bench_anymal_ad; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/{sys.argv[1]}.log").read())
    print(f"{sys.argv[1]:28s} value {d['value']:.4g} evals/s  frac {d['roofline']['frac']:.3f}  kernel {d['roofline']['kernel_ms']:.4f} ms x {d['roofline']['launches_per_step']}  checksum {d['checksum']:.12g}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
echo "== Gauss-Newton term (MFMA kernels, lane-per-node kernel, chain) and layouts"
timeout 300 python tools/bench_gn_hessian.py 2>&1 | tail -1 > gpurun_out/gn_full.json
timeout 300 python tools/bench_gn_hessian.py --upper 2>&1 | tail -1 > gpurun_out/gn_upper.json
timeout 300 python tools/bench_gn_lanes.py 2>&1 | tail -1 > gpurun_out/gn_lanes.json
echo "== config-4 chain: node Jacobians -> Gauss-Newton term (lane-per-(node, block) kernel), kernel stats + HBM / SQ counters"
bash tools/gpu_gn_tiles_profile.sh 2>&1 | tail -24
tools/_bin/gn_tiles_bench 2>&1 | tail -4 | tee gpurun_out/gn_tiles_ablation.log
timeout 300 python tools/bench_gn_shapes.py 2>/dev/null | tail -1 > gpurun_out/gn_shapes.json
timeout 300 python tools/bench_rbd_nodes.py 2>/dev/null | tail -1 > gpurun_out/rbd_nodes.json
timeout 300 python tools/bench_rbd_nodes.py 81920 2>/dev/null | tail -1 > gpurun_out/rbd_nodes_81920.json
timeout 300 python tools/bench_rbd_nodes.py 262144 2>/dev/null | tail -1 > gpurun_out/rbd_nodes_262144.json
(source tools/use_measurement_build.sh; UNGAR_AMD_RNEA_LANE_PER_NODE=1 UNGAR_AMD_CRBA_LANE_PER_NODE=1 UNGAR_AMD_CENTROIDAL_LANE_PER_NODE=1 timeout 300 python tools/bench_rbd_nodes.py 2>/dev/null | tail -1 > gpurun_out/rbd_nodes_lane_per_node.json)
rm -rf gpurun_out/rbd_prof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rbd_prof -o rbd -- python tools/bench_rbd_nodes.py > /dev/null 2>&1
for m in anymal quadrotor rc_car; do timeout 300 python tools/bench_layouts.py $m $([ $m = anymal ] && echo 81920 || ([ $m = quadrotor ] && echo 524288 || echo 3276800)) 2>/dev/null | tail -1; done > gpurun_out/layouts_all.log
tools/_bin/valu_f64_peak 2>&1 | tail -5 | tee gpurun_out/fp64_peaks.log; tools/_bin/mfma_f64_peak 2>&1 | tail -1 | tee -a gpurun_out/fp64_peaks.log
timeout 300 python tools/bench_layouts.py 2>&1 | tail -3 > gpurun_out/layouts.log
timeout 300 python tools/bench_ocp_step.py 2>&1 | tail -1 > gpurun_out/ocp_step_srbd.json
echo "== batched SQP"
bash tools/gpu_sqp_profile.sh 2>&1 | tail -12
timeout 300 python tools/bench_sqp_anymal.py 2>/dev/null | tail -1 | tee gpurun_out/sqp_anymal.json | cut -c1-400
timeout 300 python tools/bench_sqp_srbd.py 2>/dev/null | tail -1 | tee gpurun_out/sqp_srbd.json | cut -c1-400
echo "== Riccati solve alone (all block sizes) and the C++ batched SQP on the reference's OCPs as written"
timeout 300 python tools/bench_riccati_sizes.py 4096 2>/dev/null | grep nx | tee gpurun_out/riccati_sizes.log | cut -c1-120
[ -f build/variants/lib_riccati_clocks.so ] && UNGAR_AMD_LIBRARY=$PWD/build/variants/lib_riccati_clocks.so timeout 300 python tools/bench_riccati_sizes.py 4096 2>/dev/null | grep nx > gpurun_out/riccati_sizes_clocks.log
bash tools/gpu_batched_sqp_profile.sh 4096 2>&1 | tail -30
echo "== shooting assembly kernel: wavefront-specialised sections against the generic ones (bits, wall clock, kernel split, section clocks)"
bash tools/gpu_assemble_ab.sh 2>&1 | grep -vE "^iteration [12] instance" | tail -40 | tee gpurun_out/assemble_ab.log | cut -c1-300
echo "== one-wavefront assembly kernel: three nodes per SIMD against the previous tree's kernel (wall clock, section clocks, kernel split, SQ counters)"
bash tools/gpu_assemble_occupancy.sh 2>&1 | tee gpurun_out/assemble_occupancy.log | cut -c1-300
bash tools/gpu_pmc_assemble_wave.sh > /dev/null 2>&1; cat gpurun_out/assemble_wave_sq_counters.log
echo "== rocprofv3 kernel trace + HBM counters (separate passes)"
rm -rf gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2
B="python bench.py --no-cpu-baseline --no-sub-results"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o anymal -- $B > gpurun_out/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o anymal -- $B --steps 5 --warmup 1 > gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc2 -o anymal -- $B --steps 5 --warmup 1 > gpurun_out/pmc2.log 2>&1
head -2 gpurun_out/prof/anymal_kernel_stats.csv | cut -c1-200
echo "== SQ counters of the headline kernel"
bash tools/gpu_pmc_sq.sh anymal 2>&1 | tail -18 | tee gpurun_out/sq_counters.log
echo "== the reference's own tests and examples: exit codes and wall times"
bash tools/gpu_reference_programs.sh 2>&1 | tail -30
echo "== headline kernel: fused (product) against the split producer / consumer program at two wavefronts per SIMD -- resources and SQ wave counters"
bash tools/gpu_split_occupancy.sh 2>&1 | tail -30
echo "== user OCPs of stage sizes without prebuilt kernels through the batched driver (kernel factory routes)"
for s in 10_3_0 20_9_4; do build/batched_user_ocp_test_$s /tmp/cg_user 4096 4 2>&1 | grep -E "routes|^iteration [12]:|PASS|FAIL"; done | tee gpurun_out/user_ocp.log
echo "== single-instance host call: Function::Jacobian of the quadrotor node (resident wavefront; one launch per call with UNGAR_AMD_HOST_CALL_RESIDENT_US=0), the bare round trip"
{ for i in 1 2 3; do timeout 120 build/function_test /tmp/cg_lat latency 2>&1 | grep HOST_CALL; done; echo "UNGAR_AMD_HOST_CALL_RESIDENT_US=0:"; UNGAR_AMD_HOST_CALL_RESIDENT_US=0 timeout 120 build/function_test /tmp/cg_lat latency 2>&1 | grep HOST_CALL; } | tee gpurun_out/host_call.log
timeout 120 tools/_bin/resident_pingpong 2>&1 | tee gpurun_out/resident_pingpong.log
