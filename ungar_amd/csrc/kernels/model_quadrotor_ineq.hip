// ungar_amd :: built-in inequality node 'quadrotor_ineq': the 8 rotor-speed bound rows of one knot of the quadrotor OCP
// (quadrotor.example.cpp:280-288) and their (constant) Jacobian w.r.t. (x, u); feeds the barrier terms of the batched SQP.
#include "../gen/quadrotor_ineq_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(quadrotor_ineq, 128)
