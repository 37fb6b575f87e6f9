// ungar_amd :: kernels for the built-in 'anymal' (ANYmal B full-body) shooting-node model:
// structured implicit differentiation (CRBA + RNEA tangents + U D U^T solves), phased body whose
// cross-phase state lives in per-lane LDS slots.  One wavefront per workgroup,
// one workgroup per CU (all 160 KiB of LDS).
#include "../gen/anymal_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal, 64)
