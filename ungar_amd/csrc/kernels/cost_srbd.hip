// ungar_amd :: built-in scalar stage-cost node 'srbd_cost' (quadruped.example.cpp:209-245 per knot).
#include "../gen/srbd_cost_gen.hpp"
#include "cost_kernel.hpp"

UNGAR_AMD_DEFINE_COST_MODEL(srbd_cost)
