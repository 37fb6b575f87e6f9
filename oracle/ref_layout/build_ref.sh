#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Builds oracle/_ref/layout_dump from the reference's own headers WHERE THEY
# LIE under /root/reference (nothing is copied into the repo; scratch lives in $SCRATCH outside
# it) and regenerates tests/golden/layout_*.txt.  Only runs where /root/reference exists (this
# container); the GPU box uses the committed fixtures and the prebuilt binary.
#
# Only the layout half of the reference is buildable offline: the autodiff /
# rbd / optimization headers need CppAD, CppADCodeGen, Pinocchio, OSQP which are fetched from the
# network by the reference's CMake and are absent here -> derivative half is "parity unpinned".
set -euo pipefail
REF=${UNGAR_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
OUT="$REPO/oracle/_ref"
SCRATCH=${UNGAR_REF_SCRATCH:-/tmp/ungar_ref_build}
[ -d "$REF/include/ungar" ] || { echo "reference not present, skipping" >&2; exit 0; }
mkdir -p "$OUT" "$SCRATCH"
cd "$SCRATCH"
[ -d eigen-3.4.0 ] || unzip -q "$REF/external/config/eigen/eigen-3.4.0.zip"
[ -d hana-boost-1.84.0 ] || unzip -q "$REF/external/config/hana/hana-boost-1.84.0.zip"
[ -d preprocessor-1.84.0-ungar ] || unzip -q "$REF/external/config/preprocessor/preprocessor-1.84.0-ungar.zip"
# Declaration blocks, cut from the reference files into scratch (never into the repo).
sed -n '51,117p'  "$REF/example/mpc/quadrotor.example.cpp" > quadrotor_vars.inc
sed -n '49,122p'  "$REF/example/mpc/rc_car.example.cpp"    > rc_car_vars.inc
sed -n '56,139p'  "$REF/example/mpc/quadruped.example.cpp" > srbd_vars.inc
sed -n '37,84p'   "$REF/test/rbd/robot.test.cpp"           > anymal_mvars.inc
g++ -std=c++20 -O1 -w -I"$REF/include" -I eigen-3.4.0 -I hana-boost-1.84.0/include \
    -I preprocessor-1.84.0-ungar/include -I "$SCRATCH" \
    -fconstexpr-depth=2147483647 -fconstexpr-loop-limit=2147483647 \
    -fconstexpr-cache-depth=2147483647 -fconstexpr-ops-limit=2147483647 \
    "$HERE/layout_dump.cpp" -o "$OUT/layout_dump"
for w in quadrotor rc_car srbd anymal; do
    "$OUT/layout_dump" "$w" > "$REPO/tests/golden/layout_$w.txt"
done
echo "built $OUT/layout_dump; fixtures refreshed"
