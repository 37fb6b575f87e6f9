#!/usr/bin/env bash
# Probes for the reference's third-party stack (CppAD + CppADCodeGen headers) and, if it is installed, builds the genuine-reference
# timing program against the reference's headers at $UNGAR_REFERENCE_INCLUDE (default /root/reference/include; on another box: the
# include directory of an installed Ungar).  Exit code 3 and one line on stdout when the stack is absent -- the normal case in
# this image (CppAD, CppADCodeGen, Pinocchio, OSQP are fetched from the network by the reference's CMake).
set -uo pipefail
here="$(cd "$(dirname "$0")" && pwd)"; root="$(cd "$here/../.." && pwd)"
inc=${UNGAR_REFERENCE_INCLUDE:-/root/reference/include}
if ! echo '#include <cppad/cg.hpp>' | g++ -std=c++20 -E -x c++ - >/dev/null 2>&1; then
  echo "unavailable: cppad/cg.hpp not found by g++ on this box"; exit 3
fi
if [ ! -f "$inc/ungar/autodiff/function.hpp" ]; then
  echo "unavailable: reference headers not found at $inc"; exit 3
fi
mkdir -p "$root/oracle/_ref"
g++ -std=c++20 -O3 -march=native -DUNGAR_CONFIG_ENABLE_AUTODIFF -I "$inc" ${UNGAR_REFERENCE_EXTRA_FLAGS:-} \
    -o "$root/oracle/_ref/time_reference_function" "$here/time_reference_function.cpp" -ldl || { echo "unavailable: the genuine-reference program did not compile"; exit 3; }
echo "built oracle/_ref/time_reference_function"
