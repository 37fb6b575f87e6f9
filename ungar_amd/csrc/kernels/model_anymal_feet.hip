// ungar_amd :: built-in rigid-body quantity node 'anymal_feet' (SURVEY.md section 8(f) N4): world placements [position(3); rotation(9)] of the four foot frames of ANYmal B (rbd/quantities/frames.hpp:42-43) and their Jacobian w.r.t. q,
// one lane per configuration, whole batch per launch (body lowered from the tape of csrc/models/rbd_nodes.hpp).
#include "../gen/anymal_feet_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_feet, 64)
