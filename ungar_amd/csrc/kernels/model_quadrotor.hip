// ungar_amd :: kernels for the built-in 'quadrotor' shooting-node model (body generated from the tape).
#include "../gen/quadrotor_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(quadrotor, 128)
