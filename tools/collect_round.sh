#!/usr/bin/env bash
# Copies the judged summaries of one `tools/gpu_round.sh` run out of gpurun_out/ (scratch) into profiles/ (tracked).
# usage: tools/collect_round.sh <tag>     e.g. r02d
set -euo pipefail
tag=${1:?tag}
cd "$(dirname "$0")/.."
python tools/collect_profiles.py "$tag" anymal:4096:tiles
cp_if() { [ -s "$1" ] && cp "$1" "$2" || echo "missing: $1"; }
cp_if gpurun_out/bench.log                    profiles/bench_${tag}_anymal.json
cp_if gpurun_out/bench_2rank_gloo.log         profiles/bench_${tag}_config5_2rank_gloo_1gpu.json
cp_if gpurun_out/bench_config5_1gpu.log       profiles/bench_${tag}_config5_1gpu.json
cp_if gpurun_out/bench_config5_shard_of_8.log profiles/bench_${tag}_config5_shard_of_8.json
cp_if gpurun_out/bench_anymal_unit_fastest.log profiles/bench_${tag}_anymal_unit_fastest.json
cp_if gpurun_out/tiles_gather.json            profiles/${tag}_tiles_gather.json
cp_if gpurun_out/bench_srbd.log               profiles/bench_${tag}_srbd.json
cp_if gpurun_out/bench_anymal_sparse.log      profiles/bench_${tag}_anymal_sparse.json
cp_if gpurun_out/bench_anymal_reg.log         profiles/bench_${tag}_anymal_reg.json
cp_if gpurun_out/bench_anymal_ad.log          profiles/bench_${tag}_anymal_ad.json
cp_if gpurun_out/gn_full.json                 profiles/${tag}_gn_hessian_full.json
cp_if gpurun_out/gn_upper.json                profiles/${tag}_gn_hessian_upper.json
cp_if gpurun_out/gn_lanes.json                profiles/${tag}_gn_hessian_lanes.json
cp_if gpurun_out/layouts.log                  profiles/${tag}_layouts.log
cp_if gpurun_out/ocp_step_srbd.json           profiles/${tag}_ocp_step_srbd.json
cp_if gpurun_out/sqp_bench.log                profiles/${tag}_sqp_quadrotor_timing.json
cp_if gpurun_out/sqp_prof/sqp_kernel_stats.csv profiles/${tag}_sqp_kernel_stats.csv
cp_if gpurun_out/sq_counters.log              profiles/${tag}_sq_counters.log
cp_if gpurun_out/pytest_gpu.log               profiles/pytest_gpu_${tag}.log
cp_if gpurun_out/smoke.log                    profiles/smoke_${tag}.log
cp_if gpurun_out/gn_chain.json                profiles/${tag}_gn_chain.json
cp_if gpurun_out/gn_chain_natural_stride.json profiles/${tag}_gn_chain_natural_stride.json
cp_if gpurun_out/gn_tiles_counters.json       profiles/${tag}_gn_tiles_counters.json
cp_if gpurun_out/gnt_prof/gnt_kernel_stats.csv profiles/${tag}_gn_chain_kernel_stats.csv
cp_if gpurun_out/gn_tiles_ablation.log        profiles/${tag}_gn_tiles_ablation.log
cp_if gpurun_out/fp64_peaks.log               profiles/${tag}_fp64_peaks.log
cp_if gpurun_out/prewarm.log                  profiles/${tag}_prewarm.log
cp_if gpurun_out/gn_shapes.json                profiles/${tag}_gn_shapes.json
cp_if gpurun_out/rbd_nodes.json                profiles/${tag}_rbd_nodes.json
cp_if gpurun_out/rbd_nodes_81920.json          profiles/${tag}_rbd_nodes_81920.json
cp_if gpurun_out/rbd_nodes_262144.json         profiles/${tag}_rbd_nodes_262144.json
cp_if gpurun_out/rbd_nodes_lane_per_node.json  profiles/${tag}_rbd_nodes_lane_per_node.json
cp_if "$(ls gpurun_out/rbd_prof/*/rbd_kernel_stats.csv gpurun_out/rbd_prof/rbd_kernel_stats.csv 2>/dev/null | head -1)" profiles/${tag}_rbd_nodes_kernel_stats.csv
cp_if gpurun_out/layouts_all.log              profiles/${tag}_layouts_all.log
cp_if gpurun_out/sqp_anymal.json              profiles/${tag}_sqp_anymal_timing.json
cp_if gpurun_out/sqp_bench.log                profiles/${tag}_sqp_quadrotor_timing.json
cp_if gpurun_out/sqp_srbd.json                profiles/${tag}_sqp_srbd_timing.json
cp_if gpurun_out/riccati_sizes.log            profiles/${tag}_riccati_sizes.log
cp_if gpurun_out/assemble_ab.log              profiles/${tag}_assemble_ab.log
cp_if gpurun_out/assemble_occupancy.log       profiles/${tag}_assemble_occupancy.log
cp_if gpurun_out/assemble_occupancy_kernel_stats.csv profiles/${tag}_assemble_occupancy_kernel_stats.csv
cp_if gpurun_out/assemble_wave_sq_counters.log profiles/${tag}_assemble_wave_sq_counters.log
cp_if gpurun_out/assemble_ab_kernel_stats.csv profiles/${tag}_assemble_ab_kernel_stats.csv
cp_if gpurun_out/riccati_sizes_clocks.log     profiles/${tag}_riccati_phase_clocks.log
cp_if gpurun_out/batched_quadrotor_kernel_stats.csv profiles/${tag}_batched_quadrotor_kernel_stats.csv
cp_if gpurun_out/batched_quadruped_kernel_stats.csv profiles/${tag}_batched_quadruped_kernel_stats.csv
cp_if gpurun_out/batched_rc_car_kernel_stats.csv profiles/${tag}_batched_rc_car_kernel_stats.csv
for p in quadrotor rc_car quadruped; do grep -HE "timing" gpurun_out/batched_${p}_timing.log 2>/dev/null; done > profiles/${tag}_batched_sqp_timing.log || true
cp_if gpurun_out/split_occupancy.log        profiles/${tag}_split_occupancy.log
cp_if gpurun_out/user_ocp.log               profiles/${tag}_user_ocp_routes.log
cp_if gpurun_out/reference_programs.log     profiles/${tag}_reference_programs.log
cp_if gpurun_out/quick_sqp.log              profiles/${tag}_quick_sqp.log
cp_if gpurun_out/host_call.log              profiles/${tag}_host_call.log
cp_if gpurun_out/resident_pingpong.log      profiles/${tag}_resident_pingpong.log
