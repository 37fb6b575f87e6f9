// ungar_amd :: built-in inequality node 'srbd_ineq': the 12 friction-cone / unilateral-contact / reach rows of one
// knot of the single-rigid-body quadruped OCP (quadruped.example.cpp:321-335) and their Jacobian w.r.t. (x, u);
// feeds the Gauss-Newton barrier term J^T diag(b''(-h)) J (soft_sqp.hpp:257-264, ungar_gn_hessian*).
#include "../gen/srbd_ineq_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(srbd_ineq, 128)
