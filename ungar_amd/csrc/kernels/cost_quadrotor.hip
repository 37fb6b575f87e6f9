// ungar_amd :: built-in scalar stage-cost node 'quadrotor_cost' (quadrotor.example.cpp:196-236 per knot).
#include "../gen/quadrotor_cost_gen.hpp"
#include "cost_kernel.hpp"

UNGAR_AMD_DEFINE_COST_MODEL(quadrotor_cost)
