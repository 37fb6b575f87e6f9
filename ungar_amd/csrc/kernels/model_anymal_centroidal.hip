// ungar_amd :: built-in rigid-body quantity node 'anymal_centroidal' (SURVEY.md section 8(f) N4): centroidal momentum h_G(q, v) of ANYmal B (rbd/quantities/centroidal_momentum.hpp:42-43) and d h_G / d (q, v),
// one lane per configuration, whole batch per launch (body lowered from the tape of csrc/models/rbd_nodes.hpp).
#include "../gen/anymal_centroidal_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_centroidal, 64)
