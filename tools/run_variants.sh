#!/usr/bin/env bash
# GPU box: time every build/variants/* executable (tools/make_variants.sh); "diag" as first argument
# also runs the per-phase timestamp / cache-resident-output diagnostics of tools/quad_bench.hip.
mkdir -p gpurun_out
for exe in build/variants/*; do timeout 60 $exe $(basename $exe); [ "${1:-}" = diag ] && for m in ${MODES:-1 2 3}; do timeout 60 $exe $(basename $exe) $m; done; done 2>&1 | tee gpurun_out/variants.log
