#!/usr/bin/env bash
# Builds one timing executable per codegen configuration of the lane-per-leg program (run here, CPU):
#   tools/make_variants.sh "name|codegen flags[|extra hipcc flags]" ...        -> build/variants/<name>
# then on the GPU box: tools/run_variants.sh
set -uo pipefail
cd "$(dirname "$0")/.."
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; flags=${rest%%|*}; cc=""; [ "$rest" != "$flags" ] && cc=${rest#*|}
  ( dir=/tmp/gen_v/$name; mkdir -p $dir
    ./build/ungar_codegen --out $dir --anymal-robot ungar_amd/data/anymal_b.robot --model anymal $flags > $dir/log 2>&1
    hipcc --offload-arch=gfx950 -O3 -std=c++20 -I $dir -Rpass-analysis=kernel-resource-usage $cc -o build/variants/$name tools/quad_bench.hip > $dir/cc.log 2>&1
    echo "$name: $(grep -E "ScratchSize" $dir/cc.log | head -1 | sed "s/.*remark: *//; s/\[-R.*//")" ) &
done
wait
