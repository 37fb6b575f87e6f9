// ungar_amd :: built-in node 'srbd_feet': world positions p + q * r_i of the four feet of the single-rigid-body quadruped and
// their Jacobian -- the node-local part of the foot-contact equality rows of quadruped.example.cpp:279-304.
#include "../gen/srbd_feet_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(srbd_feet, 128)
