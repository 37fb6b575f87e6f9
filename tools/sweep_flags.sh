#!/usr/bin/env bash
# Compiler-flag sweep for the lane-per-leg kernel TU.
set -uo pipefail
cd /root/repo
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result $flags -c ungar_amd/csrc/kernels/model_anymal.hip -o build/model_anymal.o 2> build/flags_err.log || { echo "flags [$flags] failed: $(tail -1 build/flags_err.log)" | tee -a build/sweep_flags.log; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o ungar_amd/lib/libungar_amd.so build/model_quadrotor.o build/model_rc_car.o build/model_srbd.o build/model_anymal.o build/model_anymal_ad.o build/model_anymal_reg.o build/gn_hessian.o build/ocp_assembly.o build/c_api.o build/function.o
  r=$(gpurun --timeout 600 -- 'bash tools/gpu_bench_variants.sh anymal' 2>&1 | grep "NodeKernel")
  echo "flags [$flags] : $r" | tee -a build/sweep_flags.log
done
