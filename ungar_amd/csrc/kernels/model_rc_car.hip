// ungar_amd :: kernels for the built-in 'rc_car' shooting-node model (body generated from the tape).
#include "../gen/rc_car_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(rc_car, 128)
