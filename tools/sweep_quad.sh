#!/usr/bin/env bash
# Parameter sweep of the lane-per-leg kernel: regenerate, rebuild model_anymal.o, relink, bench on the GPU box.
set -uo pipefail
cd /root/repo
ROBOT=ungar_amd/data/anymal_b.robot
for cfg in "$@"; do
  IFS=, read -r slots rc rd pf cpp <<< "$cfg"
  ./build/ungar_codegen --out ungar_amd/csrc/gen --anymal-robot $ROBOT --model anymal --quad-lds-slots $slots --quad-remat $rc $rd --prefetch $pf --quad-columns-per-phase $cpp > /dev/null 2>&1
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -c ungar_amd/csrc/kernels/model_anymal.hip -o build/model_anymal.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o ungar_amd/lib/libungar_amd.so build/model_quadrotor.o build/model_rc_car.o build/model_srbd.o build/model_anymal.o build/model_anymal_ad.o build/model_anymal_reg.o build/gn_hessian.o build/ocp_assembly.o build/c_api.o build/function.o
  r=$(gpurun --timeout 600 -- 'bash tools/gpu_bench_variants.sh anymal' 2>&1 | grep "NodeKernel")
  echo "cfg slots=$slots remat=($rc,$rd) prefetch=$pf colsPerPhase=$cpp : $r" | tee -a build/sweep_quad.log
done
