#!/usr/bin/env bash
# Diagnostic library with section clocks in the shooting assembly kernel (run here, CPU; needs a finished library build):
#   tools/make_shooting_clocks.sh  -> build/variants/lib_shooting_clocks.so     (two nodes print "[assemble clocks] ..." per launch)
# GPU box: LD_PRELOAD-free: point the test program at it with LD_LIBRARY_PATH=build/variants/shooting_clocks (the file is named libungar_amd.so there)
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p build/variants/shooting_clocks
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -DUNGAR_SHOOTING_CLOCKS -c ungar_amd/csrc/kernels/ocp_shooting.hip -o build/variants/ocp_shooting_clocks.o
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/shooting_clocks/libungar_amd.so $(ls build/*.o | grep -v "/ocp_shooting.o") build/variants/ocp_shooting_clocks.o
rm -f build/variants/ocp_shooting_clocks.o
echo built build/variants/shooting_clocks/libungar_amd.so
