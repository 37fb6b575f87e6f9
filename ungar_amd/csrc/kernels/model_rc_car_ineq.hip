// ungar_amd :: built-in inequality node 'rc_car_ineq': input bounds and minimum forward velocity of one knot of the RC-car OCP
// (rc_car.example.cpp:271-282) and their Jacobian w.r.t. (x, u).
#include "../gen/rc_car_ineq_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(rc_car_ineq, 128)
