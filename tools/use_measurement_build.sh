# source me: makes every program of this shell resolve libungar_amd.so to the MEASUREMENT build (csrc/runtime/measurement.hpp), the only build that
# reads the A/B switches (UNGAR_AMD_ASSEMBLE_*, UNGAR_AMD_RICCATI_VARIANT, UNGAR_GN_*, *_LANE_PER_NODE, ...) these scripts set.
_ungar_root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
export UNGAR_AMD_LIBRARY="$_ungar_root/ungar_amd/lib/measurement/libungar_amd.so"
export LD_LIBRARY_PATH="$_ungar_root/ungar_amd/lib/measurement${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
[ -f "$UNGAR_AMD_LIBRARY" ] || { echo "measurement build missing: run python -c 'import __graft_entry__ as g; g.build()'" >&2; }
