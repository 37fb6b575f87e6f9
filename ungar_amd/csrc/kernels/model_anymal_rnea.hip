// ungar_amd :: built-in rigid-body quantity node 'anymal_rnea' (SURVEY.md section 8(f) N4): joint torques tau = RNEA(q, v, a) of ANYmal B (rbd/quantities/joint_torques.hpp:42-43) and d tau / d (q, v, a),
// one lane per configuration, whole batch per launch (body lowered from the tape of csrc/models/rbd_nodes.hpp).
#include "../gen/anymal_rnea_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_rnea, 64)
