#!/usr/bin/env bash
# Measurement variants of the shooting assembly kernel (run here, CPU; needs a finished library build):
#   build/variants/shooting_<name>/libungar_amd.so for name in  entry_lanes (barrier terms one lane per entry),  mirror_one_wave,  both (= the first
#   wavefront-specialised version),  clocks (section clocks).  GPU box: LD_LIBRARY_PATH=build/variants/shooting_<name> build/batched_quadruped_test ...
set -euo pipefail
cd "$(dirname "$0")/.."
variant() {  # name, flags
  local name=$1; shift
  mkdir -p build/variants/shooting_$name
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result "$@" -c ungar_amd/csrc/kernels/ocp_shooting.hip -o build/variants/ocp_shooting_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/shooting_$name/libungar_amd.so $(ls build/*.o | grep -v "/ocp_shooting.o") build/variants/ocp_shooting_$name.o
  rm -f build/variants/ocp_shooting_$name.o
  echo built build/variants/shooting_$name/libungar_amd.so
}
variant entry_lanes -DUNGAR_ASSEMBLE_ENTRY_LANES
variant mirror_one_wave -DUNGAR_ASSEMBLE_MIRROR_ONE_WAVE
variant both -DUNGAR_ASSEMBLE_ENTRY_LANES -DUNGAR_ASSEMBLE_MIRROR_ONE_WAVE
variant clocks -DUNGAR_SHOOTING_CLOCKS
variant eu3_k4 -DUNGAR_ASSEMBLE_WAVES_PER_EU=3 -DUNGAR_ASSEMBLE_K_GROUP=4
variant eu4_k4 -DUNGAR_ASSEMBLE_WAVES_PER_EU=4 -DUNGAR_ASSEMBLE_K_GROUP=4
variant eu3_k2 -DUNGAR_ASSEMBLE_WAVES_PER_EU=3 -DUNGAR_ASSEMBLE_K_GROUP=2
variant wave_eu2 -DUNGAR_ASSEMBLE_WAVE_EU=2  # the one-wavefront kernel held at two nodes per SIMD (its code at three: 10.9 KB of LDS, <= 168 registers)
# The kernel of an earlier tree for the A/B of tools/gpu_assemble_occupancy.sh / gpu_pmc_assemble_wave.sh: build/variants/shooting_old/libungar_amd.so = this tree's
# objects with ocp_shooting.hip compiled (measurement flavour) from the sources of commit $1 (default a164aea: the one-wavefront assembly kernel at two nodes per SIMD).
old_commit=${1:-a164aea}
rm -rf build/exp/old && mkdir -p build/exp/old build/variants/shooting_old
git archive "$old_commit" ungar_amd/csrc include | tar -x -C build/exp/old
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -DUNGAR_AMD_MEASUREMENT -c build/exp/old/ungar_amd/csrc/kernels/ocp_shooting.hip -o build/exp/ocp_shooting_old.o
objs=$( (ls build/measurement/*.o | grep -v /ocp_shooting.o; for o in build/*.o; do [ -f build/measurement/$(basename $o) ] || echo $o; done) )
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/shooting_old/libungar_amd.so $objs build/exp/ocp_shooting_old.o
echo built build/variants/shooting_old/libungar_amd.so from $old_commit
