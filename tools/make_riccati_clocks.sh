#!/usr/bin/env bash
# Diagnostic library with phase clocks in the batched Riccati kernel (run here, CPU; needs a finished library build):
#   tools/make_riccati_clocks.sh [extra hipcc flags]  -> build/variants/lib_riccati_clocks.so
# GPU box: UNGAR_AMD_LIBRARY=$PWD/build/variants/lib_riccati_clocks.so python tools/bench_riccati_phases.py
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -DUNGAR_RICCATI_CLOCKS "$@" -c ungar_amd/csrc/kernels/ocp_riccati.hip -o build/variants/ocp_riccati_clocks.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_riccati_clocks.so $(ls build/*.o | grep -v "/ocp_riccati.o") build/variants/ocp_riccati_clocks.o
rm -f build/variants/ocp_riccati_clocks.o
echo built build/variants/lib_riccati_clocks.so
