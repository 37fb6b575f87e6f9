#!/usr/bin/env bash
# Library variant with extra flags on the batched Riccati translation unit (run here, CPU; needs a finished library build):
#   tools/make_riccati_variant.sh <name> [hipcc flags]  -> build/variants/lib_riccati_<name>.so
# GPU box: UNGAR_AMD_LIBRARY=$PWD/build/variants/lib_riccati_<name>.so python tools/bench_riccati_sizes.py
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result "$@" -c ungar_amd/csrc/kernels/ocp_riccati.hip -o build/variants/ocp_riccati_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_riccati_$name.so $(ls build/*.o | grep -v "/ocp_riccati.o") build/variants/ocp_riccati_$name.o
rm -f build/variants/ocp_riccati_$name.o
echo built build/variants/lib_riccati_$name.so
