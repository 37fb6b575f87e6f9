#!/usr/bin/env bash
# A/B builds of the rigid-body quantity node kernels (anymal_rnea / anymal_crba / anymal_centroidal) under different
# code-generation settings (run here, CPU; needs a finished library build for the other objects):
#   tools/make_rbd_variants.sh "name|codegen flags" ...      -> build/variants/lib_<name>.so
#   e.g.  "f160|--jac-mode 1 --rbd-lds-slots 160"  "r160|--jac-mode 2 --rbd-lds-slots 160"  "plain|--rbd-lds-slots 0"
# then on the GPU box: tools/run_rbd_variants.sh (rates + tests/test_rbd_nodes.py per library, via UNGAR_AMD_LIBRARY).
# A reverse-mode variant (--jac-mode 2) keeps the library's anymal_crba (324 outputs: reverse accumulation is not an option).
set -uo pipefail
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; flags=${rest%%|*}; ccflags=""; [[ "$rest" == *"|"* ]] && ccflags=${rest#*|}   # "name|codegen flags[|hipcc flags]"
  ( dir=/tmp/rbd_v/$name; rm -rf $dir; mkdir -p $dir/gen $dir/kernels
    models=${RBD_VARIANT_MODELS:-"anymal_rnea anymal_centroidal"}; [[ -z "${RBD_VARIANT_MODELS:-}" && "$flags" != *"--jac-mode 2"* ]] && models="$models anymal_crba"
    sel=""; for m in $models; do sel="$sel --model $m"; done
    ./build/ungar_codegen --out $dir/gen --anymal-robot ungar_amd/data/anymal_b.robot $flags $sel > $dir/log 2>&1
    cp ungar_amd/csrc/kernels/*.hpp $dir/kernels/
    objs=""; skip=""
    for m in $models; do
      cp ungar_amd/csrc/kernels/model_$m.hip $dir/kernels/
      ( cd $dir/kernels && hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I $root/include -Rpass-analysis=kernel-resource-usage $ccflags -c model_$m.hip -o ../model_$m.o > ../cc_$m.log 2>&1 )
      objs="$objs $dir/model_$m.o"; skip="$skip|model_$m.o"
      echo "$name $m: $(grep -A8 'Function Name:.*ELi2E' $dir/cc_$m.log | grep -m1 ScratchSize | sed 's/.*remark: *//; s/\[-R.*//') $(grep -A12 'Function Name:.*QuadRneaKernel.*ELb1ELb1E' $dir/cc_$m.log | grep -E 'VGPRs:|AGPRs|ScratchSize' | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' ')"
    done
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$name.so $(ls build/*.o | grep -Ev "${skip#|}") $objs ) &
done
wait
