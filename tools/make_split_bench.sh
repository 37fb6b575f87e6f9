#!/usr/bin/env bash
# Builds tools/quad_split_bench.hip (run here, CPU): the fused lane-per-leg kernel with 8- and 16-byte stores next to the split
# kernel (two wavefronts per SIMD), against a code generation with paired sinks.   -> build/variants/quad_split_bench
#   tools/make_split_bench.sh [extra codegen flags]      then on the GPU box: build/variants/quad_split_bench <tag> [nodes] [variant substring]
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p build/variants /tmp/gen_split_bench
./build/ungar_codegen --out /tmp/gen_split_bench --anymal-robot ungar_amd/data/anymal_b.robot --model anymal --quad-pair-stores 1 "$@" 2>&1 | grep "anymal_split\|lane per leg"
hipcc --offload-arch=gfx950 -O3 -std=c++20 -w -I /tmp/gen_split_bench -Rpass-analysis=kernel-resource-usage -o build/variants/quad_split_bench tools/quad_split_bench.hip 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/\[-R.*//' | cut -c1-160
