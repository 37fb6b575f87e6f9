// Dumps the index/size table of the four workload hierarchies built with ungar_amd's own variable
// system, in the format of oracle/ref_layout/layout_dump.cpp, so that tests/test_layout.py can diff
// it bit-exact against the fixtures produced by the reference's engine (tests/golden/layout_*.txt).
// The hierarchies are the workload definitions of BASELINE.json (names and sizes are data):
// quadrotor / rc_car / SRBD quadruped MPC variables and the ANYmal B q-v-tau m-variables.
#include <cstdio>
#include <string>

#include "ungar/mvariable_lazy_map.hpp"
#include "ungar/variable_map.hpp"

using namespace Ungar;

template <class V>
void Dump(const char* workload, const V& root) {
    root.ForEach([&](auto var) {
        const char* kind = var.IsScalar() ? "scalar" : var.IsQuaternion() ? "quaternion" : var.IsVector() ? "vector" : "branch";
        std::printf("%s %s %lld %lld %s\n", workload, var.Name().c_str(), static_cast<long long>(var.Index()),
                    static_cast<long long>(var.Size()), kind);
    });
}

namespace quadrotor {
constexpr auto N = 30_c;
constexpr auto ROTORS = 4_c;
UNGAR_VARIABLE(position, 3);
UNGAR_VARIABLE(orientation, Q);
UNGAR_VARIABLE(linear_velocity, 3);
UNGAR_VARIABLE(b_angular_velocity, 3);
UNGAR_VARIABLE(x) <<= (position, orientation, linear_velocity, b_angular_velocity);
UNGAR_VARIABLE(X) <<= (N + 1_c) * x;
UNGAR_VARIABLE(rotor_speed, 1);
UNGAR_VARIABLE(u) <<= ROTORS * rotor_speed;
UNGAR_VARIABLE(U) <<= N * u;
UNGAR_VARIABLE(step_size, 1);
UNGAR_VARIABLE(mass, 1);
UNGAR_VARIABLE(b_moi_diagonal, 3);
UNGAR_VARIABLE(b_propeller_position, 3);
UNGAR_VARIABLE(standard_gravity, 1);
UNGAR_VARIABLE(thrust_constant, 1);
UNGAR_VARIABLE(drag_constant, 1);
UNGAR_VARIABLE(max_rotor_speed, 1);
UNGAR_VARIABLE(reference_position, 3);
UNGAR_VARIABLE(reference_orientation, Q);
UNGAR_VARIABLE(reference_linear_velocity, 3);
UNGAR_VARIABLE(b_reference_angular_velocity, 3);
UNGAR_VARIABLE(measured_position, 3);
UNGAR_VARIABLE(measured_orientation, Q);
UNGAR_VARIABLE(measured_linear_velocity, 3);
UNGAR_VARIABLE(b_measured_angular_velocity, 3);
UNGAR_VARIABLE(measured_state) <<= (measured_position, measured_orientation, measured_linear_velocity, b_measured_angular_velocity);
UNGAR_VARIABLE(decision_variables) <<= (X, U);
UNGAR_VARIABLE(parameters) <<= (step_size, mass, b_moi_diagonal, ROTORS * b_propeller_position, standard_gravity, thrust_constant,
                                drag_constant, max_rotor_speed, (N + 1_c) * reference_position, (N + 1_c) * reference_orientation,
                                (N + 1_c) * reference_linear_velocity, (N + 1_c) * b_reference_angular_velocity, measured_state);
UNGAR_VARIABLE(variables) <<= (decision_variables, parameters);
// offsets quoted in SURVEY.md §8(a) A1 / reference README.md:35-39
static_assert(x.Size() == 13 && X.Size() == 403 && U.Size() == 120 && decision_variables.Size() == 523 && parameters.Size() == 437);
static_assert(variables(x, 1).Index() == 13 && variables(u, 0).Index() == 403 && variables(u, 29).Index() == 519);
static_assert(variables(step_size).Index() == 523 && variables(reference_position, 0).Index() == 544);
static_assert(variables(reference_orientation, 0).Index() == 637 && variables(measured_state).Index() == 947);
static_assert(variables(rotor_speed, 3, 2).Index() == 403 + 3 * 4 + 2);
static_assert(X.At<"x">(2).Index() == X(x, 2).Index());
}  // namespace quadrotor

namespace rc_car {
constexpr auto N = 30_c;
UNGAR_VARIABLE(position, 2);
UNGAR_VARIABLE(yaw, 1);
UNGAR_VARIABLE(b_linear_velocity, 2);
UNGAR_VARIABLE(yaw_rate, 1);
UNGAR_VARIABLE(x) <<= (position, yaw, b_linear_velocity, yaw_rate);
UNGAR_VARIABLE(X) <<= (N + 1_c) * x;
UNGAR_VARIABLE(pwm_duty_cycle, 1);
UNGAR_VARIABLE(steering_angle, 1);
UNGAR_VARIABLE(u) <<= (pwm_duty_cycle, steering_angle);
UNGAR_VARIABLE(U) <<= N * u;
UNGAR_VARIABLE(step_size, 1);
UNGAR_VARIABLE(mass, 1);
UNGAR_VARIABLE(b_moi, 1);
UNGAR_VARIABLE(front_wheel_distance, 1);
UNGAR_VARIABLE(rear_wheel_distance, 1);
UNGAR_VARIABLE(ptm_front_b, 1);
UNGAR_VARIABLE(ptm_front_c, 1);
UNGAR_VARIABLE(ptm_front_d, 1);
UNGAR_VARIABLE(ptm_rear_b, 1);
UNGAR_VARIABLE(ptm_rear_c, 1);
UNGAR_VARIABLE(ptm_rear_d, 1);
UNGAR_VARIABLE(ptm_cm1, 1);
UNGAR_VARIABLE(ptm_cm2, 1);
UNGAR_VARIABLE(ptm_cr0, 1);
UNGAR_VARIABLE(ptm_cr2, 1);
UNGAR_VARIABLE(reference_position, 2);
UNGAR_VARIABLE(reference_trajectory) <<= (N + 1_c) * reference_position;
UNGAR_VARIABLE(measured_position, 2);
UNGAR_VARIABLE(measured_yaw, 1);
UNGAR_VARIABLE(b_measured_linear_velocity, 2);
UNGAR_VARIABLE(measured_yaw_rate, 1);
UNGAR_VARIABLE(measured_state) <<= (measured_position, measured_yaw, b_measured_linear_velocity, measured_yaw_rate);
UNGAR_VARIABLE(decision_variables) <<= (X, U);
UNGAR_VARIABLE(parameters) <<= (step_size, mass, b_moi, front_wheel_distance, rear_wheel_distance, ptm_front_b, ptm_front_c, ptm_front_d,
                                ptm_rear_b, ptm_rear_c, ptm_rear_d, ptm_cm1, ptm_cm2, ptm_cr0, ptm_cr2, reference_trajectory, measured_state);
UNGAR_VARIABLE(variables) <<= (decision_variables, parameters);
static_assert(decision_variables.Size() == 246 && parameters.Size() == 83);
}  // namespace rc_car

namespace srbd {
constexpr auto N = 30_c;
constexpr auto LEGS = 4_c;
UNGAR_VARIABLE(position, 3);
UNGAR_VARIABLE(orientation, Q);
UNGAR_VARIABLE(linear_velocity, 3);
UNGAR_VARIABLE(b_angular_velocity, 3);
UNGAR_VARIABLE(x) <<= (position, orientation, linear_velocity, b_angular_velocity);
UNGAR_VARIABLE(X) <<= (N + 1_c) * x;
UNGAR_VARIABLE(ground_reaction_force, 3);
UNGAR_VARIABLE(b_foot_position, 3);
UNGAR_VARIABLE(leg_input) <<= (ground_reaction_force, b_foot_position);
UNGAR_VARIABLE(u) <<= LEGS * leg_input;
UNGAR_VARIABLE(U) <<= N * u;
UNGAR_VARIABLE(reference_position, 3);
UNGAR_VARIABLE(reference_orientation, Q);
UNGAR_VARIABLE(reference_linear_velocity, 3);
UNGAR_VARIABLE(b_reference_angular_velocity, 3);
UNGAR_VARIABLE(reference_state) <<= (reference_position, reference_orientation, reference_linear_velocity, b_reference_angular_velocity);
UNGAR_VARIABLE(reference_contact_state, 1);
UNGAR_VARIABLE(b_reference_foot_position, 3);
UNGAR_VARIABLE(reference_leg_state) <<= (reference_contact_state, b_reference_foot_position);
UNGAR_VARIABLE(p) <<= (reference_state, LEGS * reference_leg_state);
UNGAR_VARIABLE(P) <<= (N + 1_c) * p;
UNGAR_VARIABLE(step_size, 1);
UNGAR_VARIABLE(mass, 1);
UNGAR_VARIABLE(b_moi_diagonal, 3);
UNGAR_VARIABLE(inertial_properties) <<= (mass, b_moi_diagonal);
UNGAR_VARIABLE(b_hip_position, 3);
UNGAR_VARIABLE(leg_length, 1);
UNGAR_VARIABLE(geometric_data) <<= (LEGS * b_hip_position, leg_length);
UNGAR_VARIABLE(standard_gravity, 1);
UNGAR_VARIABLE(friction_coefficient, 1);
UNGAR_VARIABLE(physical_constants) <<= (standard_gravity, friction_coefficient);
UNGAR_VARIABLE(measured_position, 3);
UNGAR_VARIABLE(measured_orientation, Q);
UNGAR_VARIABLE(measured_linear_velocity, 3);
UNGAR_VARIABLE(b_measured_angular_velocity, 3);
UNGAR_VARIABLE(measured_state) <<= (measured_position, measured_orientation, measured_linear_velocity, b_measured_angular_velocity);
UNGAR_VARIABLE(measured_contact_state, 1);
UNGAR_VARIABLE(measured_foot_position, 3);
UNGAR_VARIABLE(measured_leg_state) <<= (measured_contact_state, measured_foot_position);
UNGAR_VARIABLE(Rho) <<= (step_size, inertial_properties, geometric_data, physical_constants, measured_state, LEGS * measured_leg_state);
UNGAR_VARIABLE(decision_variables) <<= (X, U);
UNGAR_VARIABLE(parameters) <<= (P, Rho);
UNGAR_VARIABLE(variables) <<= (decision_variables, parameters);
static_assert(decision_variables.Size() == 1123 && parameters.Size() == 948 && p.Size() == 29 && Rho.Size() == 49);
// nested-array lookups of reference example/variable.example.cpp:73-90
static_assert(variables(u, 0, leg_input, 1).Index() == X.Size() + 6 && variables(u, 1, leg_input, 3).Index() == X.Size() + u.Size() + 18);
static_assert(variables(u, 0, leg_input, 2).Index() == variables(leg_input, 0, 2).Index());
static_assert(variables(u, 4, leg_input, 1, ground_reaction_force).Index() == variables(ground_reaction_force, 4, 1).Index());
}  // namespace srbd

namespace anymal {
inline constexpr auto NUM_LEGS = 4_idx;
UNGAR_LEAF_MVARIABLE(position, 3);
UNGAR_LEAF_MVARIABLE(orientation, Q);
UNGAR_BRANCH_MVARIABLE(base_pose, position, orientation);
UNGAR_LEAF_MVARIABLE(hip_aa, 1);
UNGAR_LEAF_MVARIABLE(hip_fe, 1);
UNGAR_LEAF_MVARIABLE(knee_fe, 1);
UNGAR_BRANCH_MVARIABLE(leg_joint_coords, hip_aa, hip_fe, knee_fe);
UNGAR_MVARIABLE_ARRAY(joint_coords, leg_joint_coords, NUM_LEGS);
UNGAR_BRANCH_MVARIABLE(q, base_pose, joint_coords);
UNGAR_LEAF_MVARIABLE(b_linear_velocity, 3);
UNGAR_LEAF_MVARIABLE(b_angular_velocity, 3);
UNGAR_BRANCH_MVARIABLE(base_twist, b_linear_velocity, b_angular_velocity);
UNGAR_BRANCH_MVARIABLE(leg_joint_vels, hip_aa, hip_fe, knee_fe);
UNGAR_MVARIABLE_ARRAY(joint_vels, leg_joint_vels, NUM_LEGS);
UNGAR_BRANCH_MVARIABLE(v, base_twist, joint_vels);
UNGAR_LEAF_MVARIABLE(b_generalized_force, 3);
UNGAR_LEAF_MVARIABLE(b_generalized_torque, 3);
UNGAR_BRANCH_MVARIABLE(base_wrench, b_generalized_force, b_generalized_torque);
UNGAR_BRANCH_MVARIABLE(leg_joint_torques, hip_aa, hip_fe, knee_fe);
UNGAR_MVARIABLE_ARRAY(joint_torques, leg_joint_torques, NUM_LEGS);
UNGAR_BRANCH_MVARIABLE(tau, base_wrench, joint_torques);
UNGAR_BRANCH_MVARIABLE(qvtau, q, v, tau);
static_assert(q.Size() == 19 && v.Size() == 18 && tau.Size() == 18 && qvtau.Size() == 55);  // test/rbd/robot.test.cpp:103-106
// optional lookup (reference mvariable.hpp:90-111): engaged for a variable of the hierarchy, empty otherwise, never an error
UNGAR_LEAF_MVARIABLE(not_part_of_the_robot, 2);
static_assert(qvtau.GetOpt(q).has_value() && qvtau.GetOpt(q)->get().Size() == 19 && qvtau.GetOpt(v).value().get().Index() == 19);
static_assert(qvtau.GetOpt(joint_torques, 2)->get().Index() == qvtau.Get(tau, joint_torques, 2).Index() && qvtau.GetOpt(qvtau)->get().Size() == 55);
static_assert(!qvtau.GetOpt(not_part_of_the_robot).has_value() && !q.GetOpt(base_twist) && !position.GetOpt(orientation).has_value());
}  // namespace anymal

#define DUMP_M(expr, label) \
    std::printf("anymal %s %lld %lld mvariable\n", label, static_cast<long long>((expr).Index()), static_cast<long long>((expr).Size()))

void DumpAnymal() {
    using namespace anymal;
    DUMP_M(qvtau, "qvtau");
    DUMP_M(qvtau.Get(q), "q");
    DUMP_M(qvtau.Get(q, base_pose), "q.base_pose");
    DUMP_M(qvtau.Get(q, base_pose, position), "q.base_pose.position");
    DUMP_M(qvtau.Get(q, base_pose, orientation), "q.base_pose.orientation");
    DUMP_M(qvtau.Get(q, joint_coords), "q.joint_coords");
    DUMP_M(qvtau.Get(v), "v");
    DUMP_M(qvtau.Get(v, base_twist), "v.base_twist");
    DUMP_M(qvtau.Get(v, base_twist, b_linear_velocity), "v.base_twist.b_linear_velocity");
    DUMP_M(qvtau.Get(v, base_twist, b_angular_velocity), "v.base_twist.b_angular_velocity");
    DUMP_M(qvtau.Get(v, joint_vels), "v.joint_vels");
    DUMP_M(qvtau.Get(tau), "tau");
    DUMP_M(qvtau.Get(tau, base_wrench), "tau.base_wrench");
    DUMP_M(qvtau.Get(tau, base_wrench, b_generalized_force), "tau.base_wrench.b_generalized_force");
    DUMP_M(qvtau.Get(tau, base_wrench, b_generalized_torque), "tau.base_wrench.b_generalized_torque");
    DUMP_M(qvtau.Get(tau, joint_torques), "tau.joint_torques");
    for (index_t leg = 0; leg < NUM_LEGS; ++leg) {
        const std::string l = std::to_string(leg);
        DUMP_M(qvtau.Get(q, joint_coords, leg_joint_coords, leg), ("q.joint_coords.leg_joint_coords[" + l + "]").c_str());
        DUMP_M(qvtau.Get(q, joint_coords, leg_joint_coords, leg, hip_aa), ("q.joint_coords.leg_joint_coords[" + l + "].hip_aa").c_str());
        DUMP_M(qvtau.Get(q, joint_coords, leg_joint_coords, leg, hip_fe), ("q.joint_coords.leg_joint_coords[" + l + "].hip_fe").c_str());
        DUMP_M(qvtau.Get(q, joint_coords, leg_joint_coords, leg, knee_fe), ("q.joint_coords.leg_joint_coords[" + l + "].knee_fe").c_str());
        DUMP_M(qvtau.Get(v, joint_vels, leg_joint_vels, leg), ("v.joint_vels.leg_joint_vels[" + l + "]").c_str());
        DUMP_M(qvtau.Get(v, joint_vels, leg_joint_vels, leg, knee_fe), ("v.joint_vels.leg_joint_vels[" + l + "].knee_fe").c_str());
        DUMP_M(qvtau.Get(tau, joint_torques, leg_joint_torques, leg), ("tau.joint_torques.leg_joint_torques[" + l + "]").c_str());
        DUMP_M(qvtau.Get(tau, joint_torques, leg_joint_torques, leg, hip_fe), ("tau.joint_torques.leg_joint_torques[" + l + "].hip_fe").c_str());
    }
}

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "all";
    if (which == "all" || which == "quadrotor") Dump("quadrotor", quadrotor::variables);
    if (which == "all" || which == "rc_car") Dump("rc_car", rc_car::variables);
    if (which == "all" || which == "srbd") Dump("srbd", srbd::variables);
    if (which == "all" || which == "anymal") DumpAnymal();
    return 0;
}
