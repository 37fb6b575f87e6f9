#!/usr/bin/env bash
# GPU box: rocprofv3 kernel stats of the sparse ANYmal kernel and of the config-3 pipeline (node Jacobians -> GN term).
set -uo pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_sparse gpurun_out/prof_pipeline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sparse -o sparse -- python bench.py --jacobian sparse --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_sparse.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipeline -o pipeline -- python tools/bench_pipeline.py > gpurun_out/prof_pipeline.log 2>&1
head -4 gpurun_out/prof_sparse/sparse_kernel_stats.csv | cut -c1-160
head -5 gpurun_out/prof_pipeline/pipeline_kernel_stats.csv | cut -c1-160
