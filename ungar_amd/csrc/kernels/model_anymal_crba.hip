// ungar_amd :: built-in rigid-body quantity node 'anymal_crba' (SURVEY.md section 8(f) N4): joint-space inertia matrix M(q) of ANYmal B, 18 x 18 row-major (rbd/quantities/joint_space_inertia_matrix.hpp:42-43) and d M / d q,
// one lane per configuration, whole batch per launch (body lowered from the tape of csrc/models/rbd_nodes.hpp).
#include "../gen/anymal_crba_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_crba, 64)
