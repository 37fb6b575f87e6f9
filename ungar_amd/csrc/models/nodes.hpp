// ungar_amd :: the four built-in shooting-node functions  x+ = f(x, u; w, p).
//
// Each restates, operation for operation, the inner dynamics lambda the reference inlines N times
// into its whole-horizon equality-constraint tape:
//   quadrotor  example/mpc/quadrotor.example.cpp:126-190   nx=13 nu=4   np=20
//   rc_car     example/mpc/rc_car.example.cpp:131-185      nx=6  nu=2   np=15
//   srbd       example/mpc/quadruped.example.cpp:148-203   nx=13 nu=24  nw=4 (contact flags) np=6
//   anymal     ABA of test/rbd/robot.test.cpp:109-135 (q19,v18) + the same Lie-group
//              semi-implicit Euler step (quadruped.example.cpp:197-200); nx=37 nu=12 np=1.
//              The reference has no MPC example for the full-body robot: this
//              node function is DEFINED here and documented in DESIGN.md.
// Argument blocks: x, u are differentiated; w = per-node parameters, p = per-instance parameters
// (both are "parameters" in the sense of function.hpp:529-550 -- trimmed from the Jacobian).
#pragma once

#include <cmath>
#include <limits>
#include <vector>

#include "../rbd/aba.hpp"
#include "small_math.hpp"

namespace ungar_amd::models {

struct NodeDims {
    const char* name;
    int nx, nu, nw, np;
    int ny = -1;  // outputs per node; -1: nx (a dynamics node x+ = f(x, u))
    constexpr int Ny() const {
        return ny < 0 ? nx : ny;
    }
};

inline constexpr NodeDims kQuadrotorDims{"quadrotor", 13, 4, 0, 20};
inline constexpr NodeDims kRcCarDims{"rc_car", 6, 2, 0, 15};
/// Inequality rows of one knot of the single-rigid-body quadruped OCP, h(u; w, p) <= 0, three per leg
/// (12 outputs): p = [friction_coefficient, 4 x b_hip_position(3), leg_length].
inline constexpr NodeDims kSrbdIneqDims{"srbd_ineq", 13, 24, 4, 14, 12};
/// Scalar stage cost of the quadruped OCP: p = [p_ref(3), q_ref(4), v_ref(3), omega_ref(3), 4 x b_reference_foot_position(3)].
inline constexpr NodeDims kSrbdCostDims{"srbd_cost", 13, 24, 0, 25};
/// Scalar stage cost of the quadrotor OCP (one output): p = [p_ref(3), q_ref(4), v_ref(3), omega_ref(3)].
inline constexpr NodeDims kQuadrotorCostDims{"quadrotor_cost", 13, 4, 0, 13};
inline constexpr NodeDims kSrbdDims{"srbd", 13, 24, 4, 6};
/// Scalar stage cost of the RC-car OCP (rc_car.example.cpp:204-222 per knot): p = [reference_position(2)].
inline constexpr NodeDims kRcCarCostDims{"rc_car_cost", 6, 2, 0, 2};
inline constexpr NodeDims kAnymalCostDims{"anymal_cost", 37, 12, 0, 42};
/// Inequality rows of one knot of the RC-car OCP (rc_car.example.cpp:271-282): [|d| - 15, |delta| - 15, 0.3 - v_x].
inline constexpr NodeDims kRcCarIneqDims{"rc_car_ineq", 6, 2, 0, 0, 3};
/// World positions of the four feet, p + q * r_i (12 outputs): the node-local part of the quadruped OCP's foot-contact
/// equality rows (quadruped.example.cpp:279-304), which combine the feet of knots k and k - 1 with the contact flags.
inline constexpr NodeDims kSrbdFeetDims{"srbd_feet", 13, 24, 0, 0, 12};
/// Inequality rows of one knot of the quadrotor OCP, h(u; p) <= 0: per rotor [r - r_max, -r] (8 outputs); p = [max_rotor_speed].
inline constexpr NodeDims kQuadrotorIneqDims{"quadrotor_ineq", 13, 4, 0, 1, 8};
inline constexpr NodeDims kAnymalDims{"anymal", 37, 12, 0, 1};

/// p = parameters[0:20] of the example's `parameters` variable: step_size, mass,
/// b_moi_diagonal(3), 4 x b_propeller_position(3), standard_gravity, thrust_constant,
/// drag_constant  (quadrotor.example.cpp:103-116).
template <class S>
void QuadrotorNode(const S* x, const S* u, const S* /*w*/, const S* p, S* xn) {
    const S dt = p[0], m = p[1];
    const Vec3<S> moi{p[2], p[3], p[4]};
    const S g0 = p[17], b = p[18], d = p[19];
    const Vec3<S> pos{x[0], x[1], x[2]};
    const Quat<S> q{x[3], x[4], x[5], x[6]};
    const Vec3<S> pDot{x[7], x[8], x[9]};
    const Vec3<S> om{x[10], x[11], x[12]};

    // quadrotor.example.cpp:149-166 -- per-rotor thrust force/moment and drag moment, summed
    // left to right from zero (std::accumulate).
    Vec3<S> sumF{S{0.0}, S{0.0}, S{0.0}}, sumM{S{0.0}, S{0.0}, S{0.0}}, sumD{S{0.0}, S{0.0}, S{0.0}};
    for (int i = 0; i < 4; ++i) {
        const S r2 = u[i] * u[i];  // Utils::Pow(r, 2) = CppAD::pow(r, int 2)
        const Vec3<S> thrust{S{0.0}, S{0.0}, b * r2};
        const Vec3<S> pP{p[5 + 3 * i], p[6 + 3 * i], p[7 + 3 * i]};
        sumF = Add(sumF, thrust);
        sumM = Add(sumM, Cross(pP, thrust));
        sumD = Add(sumD, Vec3<S>{S{0.0}, S{0.0}, d * r2 * ((i % 2) ? -1.0 : 1.0)});
    }
    // :167-170
    const Vec3<S> qF = Rotate(q, sumF);
    const Vec3<S> pDotDot{qF[0] / m, qF[1] / m, (qF[2] - m * g0) / m};
    const Vec3<S> Iw{moi[0] * om[0], moi[1] * om[1], moi[2] * om[2]};
    const Vec3<S> gyro = Cross(om, Iw);
    Vec3<S> omDot;
    for (std::size_t k = 0; k < 3; ++k) omDot[k] = (S{1.0} / moi[k]) * (sumM[k] + sumD[k] - gyro[k]);
    // :184-187
    Vec3<S> pDotN, omN, posN;
    for (std::size_t k = 0; k < 3; ++k) {
        pDotN[k] = pDot[k] + dt * pDotDot[k];
        omN[k] = om[k] + dt * omDot[k];
        posN[k] = pos[k] + dt * pDotN[k];
    }
    const Quat<S> qN = QuatMul(q, ApproximateExponentialMap(Scale(dt, omN)));
    for (std::size_t k = 0; k < 3; ++k) {
        xn[k] = posN[k];
        xn[7 + k] = pDotN[k];
        xn[10 + k] = omN[k];
    }
    for (std::size_t k = 0; k < 4; ++k) xn[3 + k] = qN[k];
}

/// p = parameters[0:15]: step_size, mass, b_moi, front_wheel_distance, rear_wheel_distance,
/// ptm_front_{b,c,d}, ptm_rear_{b,c,d}, ptm_cm1, ptm_cm2, ptm_cr0, ptm_cr2
/// (rc_car.example.cpp:105-121).
template <class S>
void RcCarNode(const S* x, const S* u, const S* /*w*/, const S* p, S* xn) {
    using std::atan;
    using std::cos;
    using std::sin;
    const S dt = p[0], m = p[1], moi = p[2], lf = p[3], lr = p[4];
    const S Bf = p[5], Cf = p[6], Df = p[7], Br = p[8], Cr = p[9], Dr = p[10];
    const S Cm1 = p[11], Cm2 = p[12], Cr0 = p[13], Cr2 = p[14];
    const S px = x[0], py = x[1], phi = x[2], vx = x[3], vy = x[4], om = x[5];
    const S d = u[0], delta = u[1];
    const double eps = std::numeric_limits<double>::epsilon();
    // rc_car.example.cpp:158-165
    const S alphaf = -atan((om * lf + vy) / (vx + eps)) + delta;
    const S alphar = atan((om * lr - vy) / (vx + eps));
    const S Ffy = Df * sin(Cf * atan(Bf * alphaf));
    const S Fry = Dr * sin(Cr * atan(Br * alphar));
    const S Frx = (Cm1 - Cm2 * vx) * d - Cr0 - Cr2 * (vx * vx);
    // :168-170
    const S vxDot = (Frx - Ffy * sin(delta) + m * vy * om) / m;
    const S vyDot = (Fry + Ffy * cos(delta) - m * vx * om) / m;
    const S omDot = (Ffy * lf * cos(delta) - Fry * lr) / moi;
    // :178-182
    const S vxN = vx + dt * vxDot, vyN = vy + dt * vyDot, omN = om + dt * omDot;
    xn[0] = px + dt * (vxN * cos(phi) - vyN * sin(phi));
    xn[1] = py + dt * (vxN * sin(phi) + vyN * cos(phi));
    xn[2] = phi + dt * omN;
    xn[3] = vxN;
    xn[4] = vyN;
    xn[5] = omN;
}

/// w = the four reference_contact_state flags of knot k; p = [step_size, mass,
/// b_moi_diagonal(3), standard_gravity]  (quadruped.example.cpp:160-176).
template <class S>
void SrbdNode(const S* x, const S* u, const S* w, const S* p, S* xn) {
    const S dt = p[0], m = p[1];
    const Vec3<S> moi{p[2], p[3], p[4]};
    const S g0 = p[5];
    const Vec3<S> pos{x[0], x[1], x[2]};
    const Quat<S> q{x[3], x[4], x[5], x[6]};
    const Vec3<S> pDot{x[7], x[8], x[9]};
    const Vec3<S> om{x[10], x[11], x[12]};
    // :166-176
    Vec3<S> pDotDot{S{0.0}, S{0.0}, -g0};
    const Vec3<S> Iw{moi[0] * om[0], moi[1] * om[1], moi[2] * om[2]};
    const Vec3<S> gyro = Cross(om, Iw);
    Vec3<S> omDot{-gyro[0], -gyro[1], -gyro[2]};
    for (int i = 0; i < 4; ++i) {
        const Vec3<S> f{u[6 * i], u[6 * i + 1], u[6 * i + 2]};
        const Vec3<S> r{u[6 * i + 3], u[6 * i + 4], u[6 * i + 5]};
        const S& s = w[i];
        const Vec3<S> rxqf = Cross(r, Rotate(q, f));
        for (std::size_t k = 0; k < 3; ++k) {
            pDotDot[k] = pDotDot[k] + s * f[k] / m;
            omDot[k] = omDot[k] + s * rxqf[k];
        }
    }
    for (std::size_t k = 0; k < 3; ++k) omDot[k] = omDot[k] / moi[k];
    // :197-200
    Vec3<S> pDotN, omN, posN;
    for (std::size_t k = 0; k < 3; ++k) {
        pDotN[k] = pDot[k] + dt * pDotDot[k];
        omN[k] = om[k] + dt * omDot[k];
        posN[k] = pos[k] + dt * pDotN[k];
    }
    const Quat<S> qN = QuatMul(q, ApproximateExponentialMap(Scale(dt, omN)));
    for (std::size_t k = 0; k < 3; ++k) {
        xn[k] = posN[k];
        xn[7 + k] = pDotN[k];
        xn[10 + k] = omN[k];
    }
    for (std::size_t k = 0; k < 4; ++k) xn[3 + k] = qN[k];
}

/// Full-body floating-base node.  x = [q(nq); v(nv)], u = actuated joint torques (the base
/// wrench of `tau` is zero), p = [step_size].
///   a   = ABA(q, v, [0_6; u])
///   v+  = v + dt a
///   p+  = p + dt R(quat) v+_lin          (v_lin is body-frame, Pinocchio free-flyer convention)
///   quat+ = quat * ApproximateExponentialMap(dt w+)
///   qj+ = qj + dt vj+
/// Integration half of the floating-base node: (x, a, dt) -> x+ (see FloatingBaseNode).
template <class S>
void IntegrateFloatingBase(const rbd::Model& model, const S* x, const S* a, const S& dt, S* xn) {
    const std::size_t nq = static_cast<std::size_t>(model.nq), nv = static_cast<std::size_t>(model.nv);
    std::vector<S> vN(nv);
    for (std::size_t k = 0; k < nv; ++k) vN[k] = x[nq + k] + dt * a[k];
    const Quat<S> quat{x[3], x[4], x[5], x[6]};
    const Vec3<S> lin = Rotate(quat, Vec3<S>{vN[0], vN[1], vN[2]});
    for (std::size_t k = 0; k < 3; ++k) xn[k] = x[k] + dt * lin[k];
    const Quat<S> qN = QuatMul(quat, ApproximateExponentialMap(Scale(dt, Vec3<S>{vN[3], vN[4], vN[5]})));
    for (std::size_t k = 0; k < 4; ++k) xn[3 + k] = qN[k];
    for (std::size_t k = 7; k < nq; ++k) xn[k] = x[k] + dt * vN[k - 1];
    for (std::size_t k = 0; k < nv; ++k) xn[nq + k] = vN[k];
}

template <class S>
void FloatingBaseNode(const rbd::Model& model, const S* x, const S* u, const S* /*w*/, const S* p, S* xn) {
    const std::size_t nq = static_cast<std::size_t>(model.nq), nv = static_cast<std::size_t>(model.nv);
    std::vector<S> q(x, x + nq), v(x + nq, x + nq + nv), tau(nv, S{0.0});
    for (std::size_t k = 6; k < nv; ++k) tau[k] = u[k - 6];
    const std::vector<S> a = rbd::Aba(model, q, v, tau);
    IntegrateFloatingBase(model, x, a.data(), p[0], xn);
}

/// Inequality constraints of example/mpc/quadruped.example.cpp:321-335 for one knot, per leg i:
///   -s f_z <= 0                                   (unilateral contact force)
///   s |f_xy|~ - mu f_z <= 0                       (friction cone, smoothed norm)
///   s |r - hip_i|~ - leg_length <= 0              (kinematic reach)
/// with u = 4 x [f(3), r(3)] as in SrbdNode, w = contact flags s; the state does not enter.
template <class S>
void SrbdIneqNode(const S* /*x*/, const S* u, const S* w, const S* p, S* h) {
    const S mu = p[0], legLength = p[13];
    for (int i = 0; i < 4; ++i) {
        const Vec3<S> f{u[6 * i], u[6 * i + 1], u[6 * i + 2]};
        const Vec3<S> r{u[6 * i + 3], u[6 * i + 4], u[6 * i + 5]};
        const Vec3<S> hip{p[1 + 3 * i], p[2 + 3 * i], p[3 + 3 * i]};
        const S& s = w[i];
        using std::sqrt;
        const S fxy = sqrt(f[0] * f[0] + f[1] * f[1] + S{std::numeric_limits<double>::epsilon()});  // Utils::ApproximateNorm
        h[3 * i] = -s * f[2];
        h[3 * i + 1] = s * fxy - mu * f[2];
        h[3 * i + 2] = s * ApproximateNorm(Sub(r, hip)) - legLength;
    }
}

/// Inequality constraints of example/mpc/quadrotor.example.cpp:280-288 for one knot: every rotor speed within [0, r_max],
/// written as the pair  r - r_max <= 0,  -r <= 0  (rotor-minor order); the state does not enter.
template <class S>
void QuadrotorIneqNode(const S* /*x*/, const S* u, const S* /*w*/, const S* p, S* h) {
    for (int i = 0; i < 4; ++i) {
        h[2 * i] = u[i] - p[0];
        h[2 * i + 1] = -u[i];
    }
}

/// Stage cost of example/mpc/rc_car.example.cpp:204-222 for one knot: reference position tracking plus the input
/// regularisation 1e-6 |u|^2; the input-VARIATION term (:216-220) couples u_k with u_{k-1} and lives in the whole-horizon function.
template <class S>
void RcCarCostNode(const S* x, const S* u, const S* /*w*/, const S* p, S* y) {
    y[0] = (x[0] - p[0]) * (x[0] - p[0]) + (x[1] - p[1]) * (x[1] - p[1]) + 1e-6 * (u[0] * u[0] + u[1] * u[1]);
}

/// Stage cost for the full-body quadruped (x = [q(19); v(18)], u = 12 joint torques).  The reference has NO full-body MPC example
/// (its ANYmal model appears only in test/rbd/robot.test.cpp), so this is the engine's own tracking cost in
/// the style of the reference's stage costs (quadrotor.example.cpp:196-236): weighted squared distance to a reference state --
/// base position, orientation through the sign-invariant term min(|q - q_ref|^2, |q + q_ref|^2), joint angles, velocities -- plus
/// input regularisation.  p = [x_ref(37), w_position, w_orientation, w_joints, w_velocity, w_torque].
template <class S>
void AnymalCostNode(const S* x, const S* u, const S* /*w*/, const S* p, S* y) {
    const S *ref = p, wPos = p[37], wQuat = p[38], wJoint = p[39], wVel = p[40], wTau = p[41];
    S pos{0.0}, qm{0.0}, qp{0.0}, joint{0.0}, vel{0.0}, tau{0.0};
    for (int i = 0; i < 3; ++i) pos = pos + (x[i] - ref[i]) * (x[i] - ref[i]);
    for (int i = 3; i < 7; ++i) {
        qm = qm + (x[i] - ref[i]) * (x[i] - ref[i]);
        qp = qp + (x[i] + ref[i]) * (x[i] + ref[i]);
    }
    for (int i = 7; i < 19; ++i) joint = joint + (x[i] - ref[i]) * (x[i] - ref[i]);
    for (int i = 19; i < 37; ++i) vel = vel + (x[i] - ref[i]) * (x[i] - ref[i]);
    for (int i = 0; i < 12; ++i) tau = tau + u[i] * u[i];
    y[0] = wPos * pos + wQuat * Min(qm, qp) + wJoint * joint + wVel * vel + wTau * tau;
}

/// Inequality constraints of example/mpc/rc_car.example.cpp:271-282 for one knot: input bounds through Utils::Abs and the
/// minimum forward velocity.
template <class S>
void RcCarIneqNode(const S* x, const S* u, const S* /*w*/, const S* /*p*/, S* h) {
    using std::abs;
    h[0] = abs(u[0]) - 15.0;
    h[1] = abs(u[1]) - 15.0;
    h[2] = 0.3 - x[3];
}

/// pFoot_i = p + q * r_i for the four legs (quadruped.example.cpp:288-291), u = 4 x [f(3), r(3)] as in SrbdNode.
template <class S>
void SrbdFeetNode(const S* x, const S* u, const S* /*w*/, const S* /*p*/, S* y) {
    const Vec3<S> pos{x[0], x[1], x[2]};
    const Quat<S> q{x[3], x[4], x[5], x[6]};
    for (int i = 0; i < 4; ++i) {
        const Vec3<S> foot = Add(pos, Rotate(q, Vec3<S>{u[6 * i + 3], u[6 * i + 4], u[6 * i + 5]}));
        for (std::size_t k = 0; k < 3; ++k) y[3 * i + k] = foot[k];
    }
}

/// Stage cost of example/mpc/quadrotor.example.cpp:196-236 for one knot: reference tracking with the
/// sign-invariant quaternion term  min(|q - q_ref|^2, |q + q_ref|^2)  (:215-218) plus the input
/// regularisation 1e-6 |u|^2 (:229-231).  The input-RATE term (:223-227) couples u_k with u_{k-1} and
/// therefore lives in the whole-horizon function, not in the per-knot block.
template <class S>
void QuadrotorCostNode(const S* x, const S* u, const S* /*w*/, const S* p, S* y) {
    S track{0.0};
    for (int i = 0; i < 3; ++i) track = track + (x[i] - p[i]) * (x[i] - p[i]);
    S minus{0.0}, plus{0.0};
    for (int i = 0; i < 4; ++i) {
        minus = minus + (x[3 + i] - p[3 + i]) * (x[3 + i] - p[3 + i]);
        plus = plus + (x[3 + i] + p[3 + i]) * (x[3 + i] + p[3 + i]);
    }
    track = track + Min(minus, plus);
    for (int i = 0; i < 6; ++i) track = track + (x[7 + i] - p[7 + i]) * (x[7 + i] - p[7 + i]);
    S reg{0.0};
    for (int i = 0; i < 4; ++i) reg = reg + u[i] * u[i];
    y[0] = track + 1e-6 * reg;
}

/// Stage cost of example/mpc/quadruped.example.cpp:215-245 for one knot: weighted position tracking
/// |diag(0.1, 0.1, 10) (p - p_ref)|^2, sign-invariant quaternion term, velocity tracking, foothold tracking
/// |r_i - r_ref_i|^2 and force regularisation 1e-8 |f_i|^2; u = 4 x [f(3), r(3)] as in SrbdNode.
template <class S>
void SrbdCostNode(const S* x, const S* u, const S* /*w*/, const S* p, S* y) {
    const double weight[3] = {0.1, 0.1, 10.0};
    S value{0.0};
    for (int i = 0; i < 3; ++i) {
        const S e = weight[i] * (x[i] - p[i]);
        value = value + e * e;
    }
    S minus{0.0}, plus{0.0};
    for (int i = 0; i < 4; ++i) {
        minus = minus + (x[3 + i] - p[3 + i]) * (x[3 + i] - p[3 + i]);
        plus = plus + (x[3 + i] + p[3 + i]) * (x[3 + i] + p[3 + i]);
    }
    value = value + Min(minus, plus);
    for (int i = 0; i < 6; ++i) value = value + (x[7 + i] - p[7 + i]) * (x[7 + i] - p[7 + i]);
    for (int leg = 0; leg < 4; ++leg) {
        S foothold{0.0}, force{0.0};
        for (int k = 0; k < 3; ++k) {
            const S e = u[6 * leg + 3 + k] - p[13 + 3 * leg + k];
            foothold = foothold + e * e;
            force = force + u[6 * leg + k] * u[6 * leg + k];
        }
        value = value + foothold + 1e-8 * force;
    }
    y[0] = value;
}

}  // namespace ungar_amd::models
