// ungar_amd :: kernels for the built-in 'anymal' (ANYmal B full-body) shooting-node model.
// Register-heavy straight-line body: one wavefront per workgroup.
#include "../gen/anymal_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal, 64)
