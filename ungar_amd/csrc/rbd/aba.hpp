// ungar_amd :: articulated-body algorithm, templated on the scalar.
//
// Hot-path row A7: the reference evaluates forward dynamics through
//   include/ungar/rbd/quantities/generalized_accelerations.hpp:42-43   pinocchio::aba(model, data, q, v, tau) -> data.ddq
//   include/ungar/rbd/evaluator.hpp:45-58
// with Scalar = ad_scalar_t so that the whole algorithm is unrolled onto the tape
// (test/rbd/robot.test.cpp:124-135).  Pinocchio is not vendored; this is a from-scratch
// statement of Featherstone's ABA in Pinocchio's published conventions:
//   * spatial vectors ordered (linear, angular), expressed in the local joint frame;
//   * free-flyer q = [p(3), quaternion xyzw(4)], v = body-frame (linear, angular) velocity,
//     rotation = Eigen's toRotationMatrix() of the quaternion as given (no normalisation);
//   * liMi = placement * joint transform; pass 1 velocities/bias, pass 2 articulated inertias,
//     pass 3 accelerations with a_gf[0] = -gravity;
//   * the 6x6 free-flyer block D = Y^A is inverted by an unpivoted LDL^T (it is SPD).
#pragma once

#include <array>
#include <cmath>
#include <vector>

#include "model.hpp"

namespace ungar_amd::rbd {

template <class S>
using Vec6 = std::array<S, 6>;
template <class S>
using Mat6 = std::array<std::array<S, 6>, 6>;
template <class S>
using Rot = std::array<std::array<S, 3>, 3>;

template <class S>
struct Xform {  // x_parent = R x_child + p
    Rot<S> R;
    std::array<S, 3> p;
};

namespace detail {

template <class S>
inline std::array<S, 3> Cross3(const std::array<S, 3>& a, const std::array<S, 3>& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
template <class S>
inline std::array<S, 3> RotMul(const Rot<S>& R, const std::array<S, 3>& v) {
    return {R[0][0] * v[0] + R[0][1] * v[1] + R[0][2] * v[2],
            R[1][0] * v[0] + R[1][1] * v[1] + R[1][2] * v[2],
            R[2][0] * v[0] + R[2][1] * v[1] + R[2][2] * v[2]};
}
template <class S>
inline std::array<S, 3> RotTMul(const Rot<S>& R, const std::array<S, 3>& v) {
    return {R[0][0] * v[0] + R[1][0] * v[1] + R[2][0] * v[2],
            R[0][1] * v[0] + R[1][1] * v[1] + R[2][1] * v[2],
            R[0][2] * v[0] + R[1][2] * v[1] + R[2][2] * v[2]};
}

/// SE3::actInv on a motion: (R^T (lin - p x ang), R^T ang).
template <class S>
inline Vec6<S> ActInvMotion(const Xform<S>& M, const Vec6<S>& m) {
    const std::array<S, 3> lin{m[0], m[1], m[2]}, ang{m[3], m[4], m[5]};
    const std::array<S, 3> pxw = Cross3(M.p, ang);
    const std::array<S, 3> l = RotTMul(M.R, std::array<S, 3>{lin[0] - pxw[0], lin[1] - pxw[1], lin[2] - pxw[2]});
    const std::array<S, 3> a = RotTMul(M.R, ang);
    return {l[0], l[1], l[2], a[0], a[1], a[2]};
}

/// SE3::act on a force: (R f, R n + p x (R f)).
template <class S>
inline Vec6<S> ActForce(const Xform<S>& M, const Vec6<S>& f) {
    const std::array<S, 3> l = RotMul(M.R, std::array<S, 3>{f[0], f[1], f[2]});
    const std::array<S, 3> a = RotMul(M.R, std::array<S, 3>{f[3], f[4], f[5]});
    const std::array<S, 3> pxl = Cross3(M.p, l);
    return {l[0], l[1], l[2], a[0] + pxl[0], a[1] + pxl[1], a[2] + pxl[2]};
}

/// Motion cross product v x m.
template <class S>
inline Vec6<S> CrossMotion(const Vec6<S>& v, const Vec6<S>& m) {
    const std::array<S, 3> vl{v[0], v[1], v[2]}, va{v[3], v[4], v[5]}, ml{m[0], m[1], m[2]}, ma{m[3], m[4], m[5]};
    const std::array<S, 3> a = Cross3(va, ml), b = Cross3(vl, ma), c = Cross3(va, ma);
    return {a[0] + b[0], a[1] + b[1], a[2] + b[2], c[0], c[1], c[2]};
}

/// Dual cross product v x* f.
template <class S>
inline Vec6<S> CrossForce(const Vec6<S>& v, const Vec6<S>& f) {
    const std::array<S, 3> vl{v[0], v[1], v[2]}, va{v[3], v[4], v[5]}, fl{f[0], f[1], f[2]}, fa{f[3], f[4], f[5]};
    const std::array<S, 3> a = Cross3(va, fl), b = Cross3(va, fa), c = Cross3(vl, fl);
    return {a[0], a[1], a[2], b[0] + c[0], b[1] + c[1], b[2] + c[2]};
}

template <class S>
inline Vec6<S> MatVec6(const Mat6<S>& Y, const Vec6<S>& v) {
    Vec6<S> r;
    for (int i = 0; i < 6; ++i) {
        S acc = Y[i][0] * v[0];
        for (int k = 1; k < 6; ++k) acc = acc + Y[i][k] * v[k];
        r[i] = acc;
    }
    return r;
}

/// Force-transform matrix of M:  Xf = [[R, 0], [p^ R, R]].
template <class S>
inline Mat6<S> ForceTransform(const Xform<S>& M) {
    Mat6<S> X{};
    const Rot<S> px{{{S{0.0}, -M.p[2], M.p[1]}, {M.p[2], S{0.0}, -M.p[0]}, {-M.p[1], M.p[0], S{0.0}}}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            X[i][j] = M.R[i][j];
            X[3 + i][3 + j] = M.R[i][j];
            X[i][3 + j] = S{0.0};
            S acc = px[i][0] * M.R[0][j];
            acc = acc + px[i][1] * M.R[1][j];
            acc = acc + px[i][2] * M.R[2][j];
            X[3 + i][j] = acc;
        }
    return X;
}

/// Xf * Ia * Xf^T (articulated inertia expressed in the parent frame).
template <class S>
inline Mat6<S> TransportInertia(const Xform<S>& M, const Mat6<S>& Ia) {
    const Mat6<S> X = ForceTransform(M);
    Mat6<S> T{}, R{};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            S acc{0.0};
            for (int k = 0; k < 6; ++k) acc = acc + X[i][k] * Ia[k][j];
            T[i][j] = acc;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
            S acc{0.0};
            for (int k = 0; k < 6; ++k) acc = acc + T[i][k] * X[j][k];
            R[i][j] = acc;
            R[j][i] = acc;
        }
    return R;
}

/// Solves the SPD 6x6 system A x = b by unpivoted LDL^T.
template <class S>
inline Vec6<S> SolveSpd6(const Mat6<S>& A, const Vec6<S>& b) {
    Mat6<S> L{};
    Vec6<S> D{};
    for (int j = 0; j < 6; ++j) {
        S d = A[j][j];
        for (int k = 0; k < j; ++k) d = d - L[j][k] * L[j][k] * D[k];
        D[j] = d;
        for (int i = j + 1; i < 6; ++i) {
            S s = A[i][j];
            for (int k = 0; k < j; ++k) s = s - L[i][k] * L[j][k] * D[k];
            L[i][j] = s / d;
        }
    }
    Vec6<S> y = b;
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < i; ++k) y[i] = y[i] - L[i][k] * y[k];
    for (int i = 0; i < 6; ++i) y[i] = y[i] / D[i];
    for (int i = 6; i-- > 0;)
        for (int k = i + 1; k < 6; ++k) y[i] = y[i] - L[k][i] * y[k];
    return y;
}

/// Eigen::Quaternion::toRotationMatrix for coefficients (x, y, z, w).
template <class S>
inline Rot<S> QuaternionToRotation(const S& x, const S& y, const S& z, const S& w) {
    const S tx = x + x, ty = y + y, tz = z + z;
    const S twx = tx * w, twy = ty * w, twz = tz * w;
    const S txx = tx * x, txy = ty * x, txz = tz * x;
    const S tyy = ty * y, tyz = tz * y, tzz = tz * z;
    return {{{S{1.0} - (tyy + tzz), txy - twz, txz + twy},
             {txy + twz, S{1.0} - (txx + tzz), tyz - twx},
             {txz - twy, tyz + twx, S{1.0} - (txx + tyy)}}};
}

/// Rotation by angle (c = cos, s = sin) about a unit axis (Rodrigues); for axis-aligned joints the
/// tape's constant folding reduces it to Pinocchio's JointModelRX/RY/RZ matrices.
template <class S>
inline Rot<S> AxisAngleRotation(const V3& a, const S& c, const S& s) {
    const S t = S{1.0} - c;
    Rot<S> R;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = (i == j ? c : S{0.0}) + t * (a[static_cast<std::size_t>(i)] * a[static_cast<std::size_t>(j)]);
    R[0][1] = R[0][1] - s * a[2];
    R[0][2] = R[0][2] + s * a[1];
    R[1][0] = R[1][0] + s * a[2];
    R[1][2] = R[1][2] - s * a[0];
    R[2][0] = R[2][0] - s * a[1];
    R[2][1] = R[2][1] + s * a[0];
    return R;
}

}  // namespace detail

/// ddq = ABA(q, v, tau).  q has model.nq entries, v/tau/ddq model.nv.
template <class S>
std::vector<S> Aba(const Model& model, const std::vector<S>& q, const std::vector<S>& v, const std::vector<S>& tau) {
    using namespace detail;
    using std::cos;
    using std::sin;
    const int n = model.NumJoints();
    std::vector<Xform<S>> liMi(static_cast<std::size_t>(n));
    std::vector<Vec6<S>> vel(static_cast<std::size_t>(n)), acc(static_cast<std::size_t>(n)), f(static_cast<std::size_t>(n));
    std::vector<Mat6<S>> Y(static_cast<std::size_t>(n));
    std::vector<Vec6<S>> U(static_cast<std::size_t>(n)), UDinv(static_cast<std::size_t>(n));
    std::vector<S> Dinv(static_cast<std::size_t>(n));
    std::vector<S> u = tau;
    std::vector<S> ddq(static_cast<std::size_t>(model.nv));

    for (auto& e : vel[0]) e = S{0.0};

    // Pass 1 (outward): placements, velocities, velocity-product bias, bias forces.
    for (int i = 1; i < n; ++i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), iq = static_cast<std::size_t>(J.idxQ), iv = static_cast<std::size_t>(J.idxV);
        Xform<S> Mj;
        Vec6<S> vj;
        if (J.type == JointType::FreeFlyer) {
            Mj.R = QuaternionToRotation(q[iq + 3], q[iq + 4], q[iq + 5], q[iq + 6]);
            Mj.p = {q[iq], q[iq + 1], q[iq + 2]};
            for (std::size_t k = 0; k < 6; ++k) vj[k] = v[iv + k];
        } else {
            Mj.R = AxisAngleRotation<S>(J.axis, cos(q[iq]), sin(q[iq]));
            Mj.p = {S{0.0}, S{0.0}, S{0.0}};
            vj = {S{0.0}, S{0.0}, S{0.0}, v[iv] * J.axis[0], v[iv] * J.axis[1], v[iv] * J.axis[2]};
        }
        // liMi = placement * Mj
        Xform<S>& M = liMi[si];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                S a{0.0};
                for (int k = 0; k < 3; ++k) a = a + J.placement.R[static_cast<std::size_t>(r)][static_cast<std::size_t>(k)] * Mj.R[static_cast<std::size_t>(k)][static_cast<std::size_t>(c)];
                M.R[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)] = a;
            }
            S a{J.placement.p[static_cast<std::size_t>(r)]};
            for (int k = 0; k < 3; ++k) a = a + J.placement.R[static_cast<std::size_t>(r)][static_cast<std::size_t>(k)] * Mj.p[static_cast<std::size_t>(k)];
            M.p[static_cast<std::size_t>(r)] = a;
        }
        vel[si] = vj;
        if (J.parent > 0) {
            const Vec6<S> vp = ActInvMotion(M, vel[static_cast<std::size_t>(J.parent)]);
            for (std::size_t k = 0; k < 6; ++k) vel[si][k] = vel[si][k] + vp[k];
        }
        acc[si] = CrossMotion(vel[si], vj);  // jdata.c() = 0 for free-flyer and revolute joints
        const auto Yd = J.inertia.Matrix();
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Y[si][r][c] = S{Yd[r][c]};
        f[si] = CrossForce(vel[si], MatVec6(Y[si], vel[si]));
    }

    // Pass 2 (inward): articulated inertias and bias forces.
    Vec6<S> uBase{};
    for (int i = n - 1; i >= 1; --i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), iv = static_cast<std::size_t>(J.idxV);
        if (J.type == JointType::FreeFlyer) {
            for (std::size_t k = 0; k < 6; ++k) uBase[k] = u[iv + k] - f[si][k];
            continue;  // root: solved in pass 3
        }
        const Vec6<S> Sax{S{0.0}, S{0.0}, S{0.0}, S{J.axis[0]}, S{J.axis[1]}, S{J.axis[2]}};
        S sf{0.0};
        for (std::size_t k = 3; k < 6; ++k) sf = sf + Sax[k] * f[si][k];
        u[iv] = u[iv] - sf;
        U[si] = MatVec6(Y[si], Sax);
        S D{0.0};
        for (std::size_t k = 3; k < 6; ++k) D = D + Sax[k] * U[si][k];
        Dinv[si] = S{1.0} / D;
        for (std::size_t k = 0; k < 6; ++k) UDinv[si][k] = U[si][k] * Dinv[si];
        Mat6<S> Ia = Y[si];
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Ia[r][c] = Ia[r][c] - UDinv[si][r] * U[si][c];
        Vec6<S> pa = MatVec6(Ia, acc[si]);
        for (std::size_t k = 0; k < 6; ++k) pa[k] = f[si][k] + pa[k] + UDinv[si][k] * u[iv];
        const std::size_t sp = static_cast<std::size_t>(J.parent);
        const Mat6<S> Yp = TransportInertia(liMi[si], Ia);
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Y[sp][r][c] = Y[sp][r][c] + Yp[r][c];
        const Vec6<S> fp = ActForce(liMi[si], pa);
        for (std::size_t k = 0; k < 6; ++k) f[sp][k] = f[sp][k] + fp[k];
    }

    // Pass 3 (outward): accelerations.  a_gf[0] = -gravity.
    acc[0] = {S{-model.gravity[0]}, S{-model.gravity[1]}, S{-model.gravity[2]}, S{0.0}, S{0.0}, S{0.0}};
    for (int i = 1; i < n; ++i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), iv = static_cast<std::size_t>(J.idxV);
        const Vec6<S> ap = ActInvMotion(liMi[si], acc[static_cast<std::size_t>(J.parent)]);
        for (std::size_t k = 0; k < 6; ++k) acc[si][k] = acc[si][k] + ap[k];
        if (J.type == JointType::FreeFlyer) {
            // ddq = D^-1 u - (U D^-1)^T a_gf  with U = Y^A, D = Y^A  =>  Y^A^-1 u - a_gf
            const Vec6<S> sol = SolveSpd6(Y[si], uBase);
            for (std::size_t k = 0; k < 6; ++k) {
                ddq[iv + k] = sol[k] - acc[si][k];
                acc[si][k] = acc[si][k] + ddq[iv + k];
            }
        } else {
            S ua{0.0};
            for (std::size_t k = 0; k < 6; ++k) ua = ua + UDinv[si][k] * acc[si][k];
            ddq[iv] = Dinv[si] * u[iv] - ua;
            for (std::size_t k = 0; k < 3; ++k) acc[si][3 + k] = acc[si][3 + k] + J.axis[k] * ddq[iv];
        }
    }
    return ddq;
}

}  // namespace ungar_amd::rbd
