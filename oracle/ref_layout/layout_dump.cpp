// TEST INFRASTRUCTURE (oracle/_ref driver) -- never linked into the product.
//
// Compiles the *reference's own* layout engine (/root/reference/include/ungar/variable.hpp,
// variable_map.hpp, mvariable_lazy_map.hpp) where it lies and dumps, for each workload's
// variable hierarchy, one line per (sub-)variable:
//
//     <workload> <name> <index> <size> <kind>
//
// The hierarchies themselves are NOT restated here: `build_ref.sh` cuts the declaration blocks
// out of the reference files into a scratch directory outside the repo
//   quadrotor  : example/mpc/quadrotor.example.cpp  PART I  (lines 51-117)
//   rc_car     : example/mpc/rc_car.example.cpp     PART I  (lines 49-122)
//   srbd       : example/mpc/quadruped.example.cpp  PART I  (lines 56-139)
//   anymal     : test/rbd/robot.test.cpp            namespace ANYmalB (lines 37-84)
// and this driver #includes them.  The dump is committed as tests/golden/layout_*.txt, the
// bit-exact fixture for SURVEY.md §8 rows A1-A4.
#include <cstdio>
#include <string>

#include "ungar/mvariable_lazy_map.hpp"
#include "ungar/variable_map.hpp"

using namespace Ungar;

namespace {

template <typename V>
void DumpVariable(const char* workload, const V& root) {
    root.ForEach([&](Concepts::Variable auto var) {
        const char* kind = var.IsScalar()       ? "scalar"
                           : var.IsQuaternion() ? "quaternion"
                           : var.IsVector()     ? "vector"
                                                : "branch";
        std::printf("%s %s %lld %lld %s\n",
                    workload,
                    var.Name().c_str(),
                    static_cast<long long>(var.Index()),
                    static_cast<long long>(var.Size()),
                    kind);
    });
}

void Quadrotor() {
#include "quadrotor_vars.inc"
    DumpVariable("quadrotor", variables);
}

void RcCar() {
#include "rc_car_vars.inc"
    DumpVariable("rc_car", variables);
}

void Srbd() {
#include "srbd_vars.inc"
    DumpVariable("srbd", variables);
}

}  // namespace

#include "anymal_mvars.inc"

namespace {

#define DUMP_M(expr, label)                                                        \
    std::printf("anymal %s %lld %lld mvariable\n",                                 \
                label,                                                             \
                static_cast<long long>((expr).Index()),                            \
                static_cast<long long>((expr).Size()))

void Anymal() {
    namespace vs = ANYmalB::Variables;
    DUMP_M(vs::qvtau, "qvtau");
    DUMP_M(vs::qvtau.Get(vs::q), "q");
    DUMP_M(vs::qvtau.Get(vs::q, vs::base_pose), "q.base_pose");
    DUMP_M(vs::qvtau.Get(vs::q, vs::base_pose, vs::position), "q.base_pose.position");
    DUMP_M(vs::qvtau.Get(vs::q, vs::base_pose, vs::orientation), "q.base_pose.orientation");
    DUMP_M(vs::qvtau.Get(vs::q, vs::joint_coords), "q.joint_coords");
    DUMP_M(vs::qvtau.Get(vs::v), "v");
    DUMP_M(vs::qvtau.Get(vs::v, vs::base_twist), "v.base_twist");
    DUMP_M(vs::qvtau.Get(vs::v, vs::base_twist, vs::b_linear_velocity),
           "v.base_twist.b_linear_velocity");
    DUMP_M(vs::qvtau.Get(vs::v, vs::base_twist, vs::b_angular_velocity),
           "v.base_twist.b_angular_velocity");
    DUMP_M(vs::qvtau.Get(vs::v, vs::joint_vels), "v.joint_vels");
    DUMP_M(vs::qvtau.Get(vs::tau), "tau");
    DUMP_M(vs::qvtau.Get(vs::tau, vs::base_wrench), "tau.base_wrench");
    DUMP_M(vs::qvtau.Get(vs::tau, vs::base_wrench, vs::b_generalized_force),
           "tau.base_wrench.b_generalized_force");
    DUMP_M(vs::qvtau.Get(vs::tau, vs::base_wrench, vs::b_generalized_torque),
           "tau.base_wrench.b_generalized_torque");
    DUMP_M(vs::qvtau.Get(vs::tau, vs::joint_torques), "tau.joint_torques");
    for (index_t leg = 0; leg < vs::NUM_LEGS; ++leg) {
        const std::string l = std::to_string(leg);
        DUMP_M(vs::qvtau.Get(vs::q, vs::joint_coords, vs::leg_joint_coords, leg),
               ("q.joint_coords.leg_joint_coords[" + l + "]").c_str());
        DUMP_M(vs::qvtau.Get(vs::q, vs::joint_coords, vs::leg_joint_coords, leg, vs::hip_aa),
               ("q.joint_coords.leg_joint_coords[" + l + "].hip_aa").c_str());
        DUMP_M(vs::qvtau.Get(vs::q, vs::joint_coords, vs::leg_joint_coords, leg, vs::hip_fe),
               ("q.joint_coords.leg_joint_coords[" + l + "].hip_fe").c_str());
        DUMP_M(vs::qvtau.Get(vs::q, vs::joint_coords, vs::leg_joint_coords, leg, vs::knee_fe),
               ("q.joint_coords.leg_joint_coords[" + l + "].knee_fe").c_str());
        DUMP_M(vs::qvtau.Get(vs::v, vs::joint_vels, vs::leg_joint_vels, leg),
               ("v.joint_vels.leg_joint_vels[" + l + "]").c_str());
        DUMP_M(vs::qvtau.Get(vs::v, vs::joint_vels, vs::leg_joint_vels, leg, vs::knee_fe),
               ("v.joint_vels.leg_joint_vels[" + l + "].knee_fe").c_str());
        DUMP_M(vs::qvtau.Get(vs::tau, vs::joint_torques, vs::leg_joint_torques, leg),
               ("tau.joint_torques.leg_joint_torques[" + l + "]").c_str());
        DUMP_M(vs::qvtau.Get(vs::tau, vs::joint_torques, vs::leg_joint_torques, leg, vs::hip_fe),
               ("tau.joint_torques.leg_joint_torques[" + l + "].hip_fe").c_str());
    }
}

}  // namespace

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "all";
    if (which == "all" || which == "quadrotor") Quadrotor();
    if (which == "all" || which == "rc_car") RcCar();
    if (which == "all" || which == "srbd") Srbd();
    if (which == "all" || which == "anymal") Anymal();
    return 0;
}
