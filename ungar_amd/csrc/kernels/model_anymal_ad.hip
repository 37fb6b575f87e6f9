// ungar_amd :: 'anymal_ad' -- the ANYmal B node with derivatives obtained by TAPING the articulated-body
// algorithm (exactly what the reference does, test/rbd/robot.test.cpp:124-135).  Kept as the
// cross-check of the structured 'anymal' kernel; not the fast path.
#include "../gen/anymal_ad_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_ad, 64)
