#!/usr/bin/env bash
# Runs every build/variants/tile_* executable (tools/quad_tile_bench.hip built against one codegen / macro configuration each) on the GPU box.
set -uo pipefail
mkdir -p gpurun_out
for exe in build/variants/tile_*; do
  n=$(basename $exe)
  timeout 120 $exe $n 81920 ${TILE_MODE:-nocheck} 2>&1 | grep -E "check|wave tiles  |unit-fastest|OFF" 
done | tee gpurun_out/${TILE_LOG:-tile_variants}.log
