// ungar_amd :: built-in scalar stage-cost node 'anymal_cost' (full-body quadruped tracking cost, csrc/models/nodes.hpp).
#include "../gen/anymal_cost_gen.hpp"
#include "cost_kernel.hpp"

UNGAR_AMD_DEFINE_COST_MODEL(anymal_cost)
