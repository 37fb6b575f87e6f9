// ungar_amd :: kernels for the built-in 'srbd' shooting-node model (body generated from the tape).
#include "../gen/srbd_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(srbd, 128)
