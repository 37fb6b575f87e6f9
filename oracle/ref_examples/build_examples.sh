#!/usr/bin/env bash
# Compiles the reference's OWN example programs, from the sources where they lie under /root/reference,
# against ungar_amd's headers and library -- nothing of the reference is copied, no reference header is
# used (the include path holds only ungar_amd/include).  That they build UNCHANGED is the API-surface
# check of BASELINE.json's north star ("the example/mpc problems link unchanged"); running them needs a
# GPU (tests/test_reference_examples.py, -m gpu).  Outputs: oracle/_ref/<name>_example (git-ignored,
# travels to the GPU box).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"; root="$(cd "$here/../.." && pwd)"
ref=${UNGAR_REFERENCE:-/root/reference}
[ -d "$ref/example" ] || { echo "reference not present: nothing to build"; exit 0; }
mkdir -p "$root/oracle/_ref"
# Second build of every example on top of the REAL Eigen 3.4 the reference bundles (external/config/eigen/eigen-3.4.0.zip,
# unpacked into a scratch directory OUTSIDE the repository): -DUNGAR_AMD_USE_SYSTEM_EIGEN makes the facade's vector /
# quaternion / sparse types Eigen's own, i.e. what an existing Ungar installation already has in every translation unit.
eigen=""
if [ -f "$ref/external/config/eigen/eigen-3.4.0.zip" ]; then
  scratch="${TMPDIR:-/tmp}/ungar_amd_reference_eigen"
  [ -d "$scratch/eigen-3.4.0/Eigen" ] || { mkdir -p "$scratch" && unzip -q -o "$ref/external/config/eigen/eigen-3.4.0.zip" -d "$scratch"; }
  eigen="$scratch/eigen-3.4.0"
fi
# Boost.Hana: only user code needs it (variable.example.cpp defines a BOOST_HANA_DEFINE_STRUCT and logs it); the real Hana 1.84 the
# reference bundles is unpacked next to Eigen and put on the include path of every example (the facade picks it up through
# __has_include and exports the `hana` alias the reference exports).
hana=()
if [ -f "$ref/external/config/hana/hana-boost-1.84.0.zip" ]; then
  hscratch="${TMPDIR:-/tmp}/ungar_amd_reference_hana"
  [ -d "$hscratch/hana-boost-1.84.0/include/boost" ] || { mkdir -p "$hscratch" && unzip -q -o "$ref/external/config/hana/hana-boost-1.84.0.zip" -d "$hscratch"; }
  hana=(-I "$hscratch/hana-boost-1.84.0/include")
fi
for name in "$@"; do
  src="$ref/example/mpc/${name}.example.cpp"
  [ -f "$src" ] || src="$ref/example/autodiff/${name}.example.cpp"
  [ -f "$src" ] || src="$ref/example/rbd/${name}.example.cpp"
  [ -f "$src" ] || src="$ref/example/${name}.example.cpp"
  data=()
  case "$src" in */example/rbd/*)  # the rbd examples open UNGAR_DATA_FOLDER "/robots/anymal_b_description/robots/anymal.urdf": written from ungar_amd/data/anymal_b.robot
    python3 "$root/tools/robot_to_urdf.py" "$root/ungar_amd/data/anymal_b.robot" "$root/oracle/_ref/data/robots/anymal_b_description/robots/anymal.urdf"
    data=(-DUNGAR_DATA_FOLDER='"/root/repo/oracle/_ref/data"')
    # (they use Eigen expression forms -- toDense().leftCols(), setRandom() on a view -- the built-in algebra does not have: real Eigen only)
    if [ -n "$eigen" ]; then
      g++ -std=c++20 -O2 -DUNGAR_AMD_USE_SYSTEM_EIGEN "${data[@]}" -I "$eigen" "${hana[@]}" -I "$root/ungar_amd/include" -o "$root/oracle/_ref/${name}_example_eigen" "$src" \
          -L "$root/ungar_amd/lib" -lungar_amd -Wl,-rpath,'$ORIGIN/../../ungar_amd/lib' -Wl,-rpath,/opt/rocm/lib
      echo "built oracle/_ref/${name}_example_eigen (real Eigen 3.4)"
    fi
    continue;;
  esac
  g++ -std=c++20 -O2 "${hana[@]}" -I "$root/ungar_amd/include" -o "$root/oracle/_ref/${name}_example" "$src" \
      -L "$root/ungar_amd/lib" -lungar_amd -Wl,-rpath,'$ORIGIN/../../ungar_amd/lib' -Wl,-rpath,/opt/rocm/lib
  echo "built oracle/_ref/${name}_example"
  if [ -n "$eigen" ]; then
    g++ -std=c++20 -O2 -DUNGAR_AMD_USE_SYSTEM_EIGEN -I "$eigen" "${hana[@]}" -I "$root/ungar_amd/include" -o "$root/oracle/_ref/${name}_example_eigen" "$src" \
        -L "$root/ungar_amd/lib" -lungar_amd -Wl,-rpath,'$ORIGIN/../../ungar_amd/lib' -Wl,-rpath,/opt/rocm/lib
    echo "built oracle/_ref/${name}_example_eigen (real Eigen 3.4)"
  fi
done
