// ungar_amd :: built-in rigid-body quantity node 'anymal_minv' (SURVEY.md section 8(f) N4): inverse joint-space inertia matrix M(q)^-1 of ANYmal B, 18 x 18 row-major (rbd/quantities/joint_space_inertia_matrix_inverse.hpp:42-43); value only,
// one lane per configuration, whole batch per launch (body lowered from the tape of csrc/models/rbd_nodes.hpp).
#include "../gen/anymal_minv_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_minv, 64)
