// ungar_amd :: 'anymal_lds' -- the structured ANYmal B node with the phased body: values that live
// across Jacobian columns (factorisation, per-leg state) are homed in per-lane LDS slots instead of
// being left to the register allocator.  One wavefront per workgroup.
#include "../gen/anymal_lds_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_lds, 64)
