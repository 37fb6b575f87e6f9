// ungar_amd :: recursive Newton-Euler, composite-rigid-body mass matrix and an U D U^T solver,
// templated on the scalar -- the ingredients of the *implicit* forward-dynamics derivatives used by
// the structured ANYmal node kernel:
//     M(q) a + h(q, v) = tau      =>   da/du = M^-1 S^T,   da/d(q,v) = -M^-1 d RNEA(q, v, a)/d(q,v)
// The reference differentiates pinocchio::aba by taping it (test/rbd/robot.test.cpp:124-135); the
// identity above yields the same derivatives from far less state per shooting node.  Conventions
// are those of aba.hpp (Pinocchio's: spatial vectors (linear, angular) in local joint frames,
// a_gf[0] = -gravity).
#pragma once

#include <vector>

#include "aba.hpp"

namespace ungar_amd::rbd {

/// Joint placements liMi for configuration q (shared by RNEA and CRBA).
template <class S>
std::vector<Xform<S>> JointPlacements(const Model& model, const std::vector<S>& q) {
    using namespace detail;
    using std::cos;
    using std::sin;
    const int n = model.NumJoints();
    std::vector<Xform<S>> liMi(static_cast<std::size_t>(n));
    for (int i = 1; i < n; ++i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t iq = static_cast<std::size_t>(J.idxQ);
        Xform<S> Mj;
        if (J.type == JointType::FreeFlyer) {
            Mj.R = QuaternionToRotation(q[iq + 3], q[iq + 4], q[iq + 5], q[iq + 6]);
            Mj.p = {q[iq], q[iq + 1], q[iq + 2]};
        } else {
            Mj.R = AxisAngleRotation<S>(J.axis, cos(q[iq]), sin(q[iq]));
            Mj.p = {S{0.0}, S{0.0}, S{0.0}};
        }
        Xform<S>& M = liMi[static_cast<std::size_t>(i)];
        for (std::size_t r = 0; r < 3; ++r) {
            for (std::size_t c = 0; c < 3; ++c) {
                S a{0.0};
                for (std::size_t k = 0; k < 3; ++k) a = a + J.placement.R[r][k] * Mj.R[k][c];
                M.R[r][c] = a;
            }
            S a{J.placement.p[r]};
            for (std::size_t k = 0; k < 3; ++k) a = a + J.placement.R[r][k] * Mj.p[k];
            M.p[r] = a;
        }
    }
    return liMi;
}

template <class S>
Vec6<S> JointMotion(const Joint& J, const std::vector<S>& v) {
    const std::size_t iv = static_cast<std::size_t>(J.idxV);
    if (J.type == JointType::FreeFlyer) return {v[iv], v[iv + 1], v[iv + 2], v[iv + 3], v[iv + 4], v[iv + 5]};
    return {S{0.0}, S{0.0}, S{0.0}, v[iv] * J.axis[0], v[iv] * J.axis[1], v[iv] * J.axis[2]};
}

/// tau = RNEA(q, v, a): inverse dynamics.  `withGravity` = false drops a_gf[0].
template <class S>
std::vector<S> Rnea(const Model& model, const std::vector<Xform<S>>& liMi, const std::vector<S>& v, const std::vector<S>& a,
                    bool withGravity = true) {
    using namespace detail;
    const int n = model.NumJoints();
    std::vector<Vec6<S>> vel(static_cast<std::size_t>(n)), acc(static_cast<std::size_t>(n)), f(static_cast<std::size_t>(n));
    for (auto& e : vel[0]) e = S{0.0};
    const double gsign = withGravity ? -1.0 : 0.0;
    acc[0] = {S{gsign * model.gravity[0]}, S{gsign * model.gravity[1]}, S{gsign * model.gravity[2]}, S{0.0}, S{0.0}, S{0.0}};
    for (int i = 1; i < n; ++i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), sp = static_cast<std::size_t>(J.parent);
        const Vec6<S> vj = JointMotion(J, v), aj = JointMotion(J, a);
        vel[si] = vj;
        if (J.parent > 0) {
            const Vec6<S> vp = ActInvMotion(liMi[si], vel[sp]);
            for (std::size_t k = 0; k < 6; ++k) vel[si][k] = vel[si][k] + vp[k];
        }
        const Vec6<S> ap = ActInvMotion(liMi[si], acc[sp]);
        const Vec6<S> cx = CrossMotion(vel[si], vj);
        for (std::size_t k = 0; k < 6; ++k) acc[si][k] = ap[k] + aj[k] + cx[k];
        Mat6<S> Y;
        const auto Yd = J.inertia.Matrix();
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Y[r][c] = S{Yd[r][c]};
        const Vec6<S> Ya = MatVec6(Y, acc[si]);
        const Vec6<S> vxf = CrossForce(vel[si], MatVec6(Y, vel[si]));
        for (std::size_t k = 0; k < 6; ++k) f[si][k] = Ya[k] + vxf[k];
    }
    std::vector<S> tau(static_cast<std::size_t>(model.nv));
    for (int i = n - 1; i >= 1; --i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), iv = static_cast<std::size_t>(J.idxV);
        if (J.type == JointType::FreeFlyer) {
            for (std::size_t k = 0; k < 6; ++k) tau[iv + k] = f[si][k];
        } else {
            tau[iv] = f[si][3] * J.axis[0] + f[si][4] * J.axis[1] + f[si][5] * J.axis[2];
        }
        if (J.parent > 0) {
            const Vec6<S> fp = ActForce(liMi[si], f[si]);
            for (std::size_t k = 0; k < 6; ++k) f[static_cast<std::size_t>(J.parent)][k] = f[static_cast<std::size_t>(J.parent)][k] + fp[k];
        }
    }
    return tau;
}

/// Joint-space inertia matrix by the composite-rigid-body algorithm (symmetric, dense storage;
/// entries between different branches of the tree are exact zeros).
template <class S>
std::vector<std::vector<S>> Crba(const Model& model, const std::vector<Xform<S>>& liMi) {
    using namespace detail;
    const int n = model.NumJoints();
    const std::size_t nv = static_cast<std::size_t>(model.nv);
    std::vector<Mat6<S>> Yc(static_cast<std::size_t>(n));
    for (int i = 1; i < n; ++i) {
        const auto Yd = model.joints[static_cast<std::size_t>(i)].inertia.Matrix();
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Yc[static_cast<std::size_t>(i)][r][c] = S{Yd[r][c]};
    }
    for (int i = n - 1; i >= 1; --i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        if (J.parent <= 0) continue;
        const Mat6<S> Yp = TransportInertia(liMi[static_cast<std::size_t>(i)], Yc[static_cast<std::size_t>(i)]);
        Mat6<S>& P = Yc[static_cast<std::size_t>(J.parent)];
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) P[r][c] = P[r][c] + Yp[r][c];
    }
    std::vector<std::vector<S>> M(nv, std::vector<S>(nv, S{0.0}));
    for (int i = 1; i < n; ++i) {
        const Joint& Ji = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), ivI = static_cast<std::size_t>(Ji.idxV);
        const int nvi = Ji.nv;
        for (int ci = 0; ci < nvi; ++ci) {
            // F = Yc_i * S_i(:, ci), then carried up the chain
            Vec6<S> Scol;
            for (std::size_t k = 0; k < 6; ++k) Scol[k] = S{0.0};
            if (Ji.type == JointType::FreeFlyer) Scol[static_cast<std::size_t>(ci)] = S{1.0};
            else for (std::size_t k = 0; k < 3; ++k) Scol[3 + k] = S{Ji.axis[k]};
            Vec6<S> F = MatVec6(Yc[si], Scol);
            int j = i;
            for (;;) {
                const Joint& Jj = model.joints[static_cast<std::size_t>(j)];
                const std::size_t ivJ = static_cast<std::size_t>(Jj.idxV);
                if (Jj.type == JointType::FreeFlyer) {
                    for (std::size_t k = 0; k < 6; ++k) {
                        M[ivJ + k][ivI + static_cast<std::size_t>(ci)] = F[k];
                        M[ivI + static_cast<std::size_t>(ci)][ivJ + k] = F[k];
                    }
                } else {
                    const S m = F[3] * Jj.axis[0] + F[4] * Jj.axis[1] + F[5] * Jj.axis[2];
                    M[ivJ][ivI + static_cast<std::size_t>(ci)] = m;
                    M[ivI + static_cast<std::size_t>(ci)][ivJ] = m;
                }
                if (Jj.parent <= 0) break;
                F = ActForce(liMi[static_cast<std::size_t>(j)], F);
                j = Jj.parent;
            }
        }
    }
    return M;
}

/// M = U D U^T with U unit UPPER triangular, pivots taken from the LAST index to the first.  With
/// the floating base first and the legs after it, eliminating from the bottom produces no fill-in
/// outside the base block (M is block-arrow); exact-zero entries stay literal zeros on the tape.
template <class S>
struct UdutFactor {
    std::vector<std::vector<S>> U;  // strictly upper part used
    std::vector<S> dinv;            // 1 / D_k
};

template <class S>
UdutFactor<S> FactorUdut(const std::vector<std::vector<S>>& M) {
    const std::size_t n = M.size();
    UdutFactor<S> F{std::vector<std::vector<S>>(n, std::vector<S>(n, S{0.0})), std::vector<S>(n)};
    std::vector<S> d(n);
    for (std::size_t kk = n; kk-- > 0;) {
        S dk = M[kk][kk];
        for (std::size_t j = kk + 1; j < n; ++j) dk = dk - F.U[kk][j] * F.U[kk][j] * d[j];
        d[kk] = dk;
        F.dinv[kk] = S{1.0} / dk;
        for (std::size_t i = 0; i < kk; ++i) {
            S s = M[i][kk];
            for (std::size_t j = kk + 1; j < n; ++j) s = s - F.U[i][j] * F.U[kk][j] * d[j];
            F.U[i][kk] = s * F.dinv[kk];
        }
    }
    return F;
}

/// Solves (U D U^T) y = r.
template <class S>
std::vector<S> SolveUdut(const UdutFactor<S>& F, const std::vector<S>& r) {
    const std::size_t n = r.size();
    std::vector<S> y = r;
    for (std::size_t i = n; i-- > 0;)  // z = U^-1 r  (U upper: back substitution from the bottom)
        for (std::size_t j = i + 1; j < n; ++j) y[i] = y[i] - F.U[i][j] * y[j];
    for (std::size_t i = 0; i < n; ++i) y[i] = y[i] * F.dinv[i];
    for (std::size_t i = 0; i < n; ++i)  // y = U^-T w  (U^T lower: forward substitution)
        for (std::size_t j = 0; j < i; ++j) y[i] = y[i] - F.U[j][i] * y[j];
    return y;
}

}  // namespace ungar_amd::rbd
