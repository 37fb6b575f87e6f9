#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Compiles the reference's OWN unit tests -- test/variable.test.cpp, test/autodiff/function.test.cpp,
# test/optimization/soft_sqp.test.cpp -- UNCHANGED, from the sources where they lie under /root/reference, against ungar_amd's
# facade headers and library.  Nothing of the reference is copied and no reference header is on the include path; GoogleTest is not
# in this image, so tests/gtest_shim/gtest/gtest.h provides the handful of macros those files use (what is under test is the
# facade, not the reference).  variable.test builds on both algebras (built-in and the real Eigen 3.4 the reference bundles);
# function.test and soft_sqp.test use Eigen expression forms the built-in algebra does not have and build on the real Eigen.
# test/utils/utils.test.cpp uses Boost.Hana in USER code (`hana::unpack`): it is built against the real Hana 1.84 the reference
# bundles (external/config/hana/hana-boost-1.84.0.zip, unpacked into a scratch directory outside the repository, like Eigen).
# test/rbd/robot.test.cpp (RobotTest.Constructor, RobotTest.Autodiff: ABA through Robot<ad_scalar_t> -> Autodiff::Function, 1024 random configurations) names a few Pinocchio
# types directly; the facade's ungar/rbd/robot.hpp answers to them on its own model (pinocchio::Model / JointModelFreeFlyer / urdf::buildModel).  It opens
# UNGAR_DATA_FOLDER "/robots/anymal_b_description/robots/anymal.urdf" at run time: a URDF with the same kinematic / inertial data is WRITTEN here from
# ungar_amd/data/anymal_b.robot (tools/robot_to_urdf.py) under oracle/_ref/data, which travels to the GPU box (/root/repo is a symlink to the snapshot there).  Outputs: oracle/_ref/ref_<name>_test[_eigen] (git-ignored, travel to the GPU box);
# run by tests/test_reference_tests.py (variable: CPU; function / soft_sqp: -m gpu).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"; root="$(cd "$here/../.." && pwd)"
ref=${UNGAR_REFERENCE:-/root/reference}
[ -d "$ref/test" ] || { echo "reference not present: nothing to build"; exit 0; }
mkdir -p "$root/oracle/_ref"
eigen=""
if [ -f "$ref/external/config/eigen/eigen-3.4.0.zip" ]; then
  scratch="${TMPDIR:-/tmp}/ungar_amd_reference_eigen"
  [ -d "$scratch/eigen-3.4.0/Eigen" ] || { mkdir -p "$scratch" && unzip -q -o "$ref/external/config/eigen/eigen-3.4.0.zip" -d "$scratch"; }
  eigen="$scratch/eigen-3.4.0"
fi
common=(-std=c++20 -O1 -I "$root/tests/gtest_shim" -I "$root/ungar_amd/include")
link=(-L "$root/ungar_amd/lib" -lungar_amd -Wl,-rpath,'$ORIGIN/../../ungar_amd/lib' -Wl,-rpath,/opt/rocm/lib)
g++ "${common[@]}" -o "$root/oracle/_ref/ref_variable_test" "$ref/test/variable.test.cpp"
echo "built oracle/_ref/ref_variable_test"
if [ -n "$eigen" ]; then
  g++ "${common[@]}" -DUNGAR_AMD_USE_SYSTEM_EIGEN -I "$eigen" -o "$root/oracle/_ref/ref_variable_test_eigen" "$ref/test/variable.test.cpp"
  g++ "${common[@]}" -DUNGAR_AMD_USE_SYSTEM_EIGEN -I "$eigen" -o "$root/oracle/_ref/ref_function_test_eigen" "$ref/test/autodiff/function.test.cpp" "${link[@]}"
  g++ "${common[@]}" -DUNGAR_AMD_USE_SYSTEM_EIGEN -I "$eigen" -o "$root/oracle/_ref/ref_soft_sqp_test_eigen" "$ref/test/optimization/soft_sqp.test.cpp" "${link[@]}"
  echo "built oracle/_ref/ref_{variable,function,soft_sqp}_test_eigen (real Eigen 3.4)"
  python3 "$root/tools/robot_to_urdf.py" "$root/ungar_amd/data/anymal_b.robot" "$root/oracle/_ref/data/robots/anymal_b_description/robots/anymal.urdf"
  g++ "${common[@]}" -DUNGAR_AMD_USE_SYSTEM_EIGEN -DUNGAR_DATA_FOLDER='"/root/repo/oracle/_ref/data"' -I "$eigen" -o "$root/oracle/_ref/ref_robot_test_eigen" "$ref/test/rbd/robot.test.cpp" "${link[@]}"
  echo "built oracle/_ref/ref_robot_test_eigen (real Eigen 3.4; robot description oracle/_ref/data/robots/anymal_b_description/robots/anymal.urdf)"
  if [ -f "$ref/external/config/hana/hana-boost-1.84.0.zip" ]; then
    hscratch="${TMPDIR:-/tmp}/ungar_amd_reference_hana"
    [ -d "$hscratch/hana-boost-1.84.0/include/boost" ] || { mkdir -p "$hscratch" && unzip -q -o "$ref/external/config/hana/hana-boost-1.84.0.zip" -d "$hscratch"; }
    g++ "${common[@]}" -DUNGAR_AMD_USE_SYSTEM_EIGEN -I "$eigen" -I "$hscratch/hana-boost-1.84.0/include" -o "$root/oracle/_ref/ref_utils_test_eigen" "$ref/test/utils/utils.test.cpp"
    echo "built oracle/_ref/ref_utils_test_eigen (real Eigen 3.4 + real Boost.Hana 1.84)"
  fi
fi
