// ungar_amd :: rigid-body quantities as batched node models.
//
// The reference evaluates these one configuration at a time through Pinocchio
// (include/ungar/rbd/quantities/{joint_torques, joint_space_inertia_matrix, joint_space_inertia_matrix_inverse, frames,
// centroidal_momentum}.hpp:42-43: rnea / crba / computeMinverse / framesForwardKinematics / ccrba).  Here each is a node
// model  y = f(x, u)  of the same ABI as the dynamics nodes: recorded on the tape scalar from the project's own tree
// algorithms (csrc/rbd), differentiated, and lowered to a lane-per-node gfx950 kernel that evaluates value and
// Jacobian d y / d (x, u) for a whole batch of configurations in one launch.
//
//   anymal_rnea        x = [q(19); v(18)], u = a(18)   y = tau = RNEA(q, v, a)                     18 outputs
//   anymal_crba        x = q(19)                        y = M(q), 18 x 18 row-major (symmetric)     324 outputs
//   anymal_minv        x = q(19)                        y = M(q)^-1, 18 x 18 row-major              324 outputs
//   anymal_feet        x = q(19)                        y = 4 x [position(3); rotation row-major(9)] of the foot frames
//                                                           LF_FOOT, LH_FOOT, RF_FOOT, RH_FOOT in the world    48 outputs
//   anymal_centroidal  x = [q(19); v(18)]               y = h_G = [linear; angular] momentum about the centre of mass, world axes   6 outputs
// Conventions are Pinocchio's (aba.hpp): q = [p, quaternion xyzw, joints], v in the local joint frames.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../rbd/rnea_crba.hpp"
#include "nodes.hpp"

namespace ungar_amd::models {

inline constexpr NodeDims kAnymalRneaDims{"anymal_rnea", 37, 18, 0, 0, 18};
inline constexpr NodeDims kAnymalCrbaDims{"anymal_crba", 19, 0, 0, 0, 324};
inline constexpr NodeDims kAnymalMinvDims{"anymal_minv", 19, 0, 0, 0, 324};
inline constexpr NodeDims kAnymalFeetDims{"anymal_feet", 19, 0, 0, 0, 48};
inline constexpr NodeDims kAnymalCentroidalDims{"anymal_centroidal", 37, 0, 0, 0, 6};
inline constexpr const char* kFootFrames[4] = {"LF_FOOT", "LH_FOOT", "RF_FOOT", "RH_FOOT"};  // test/rbd/robot.test.cpp:49-52 (frame ids 12/22/32/42)

/// tau = RNEA(q, v, a)   (rbd/quantities/joint_torques.hpp:42-43)
template <class S>
void JointTorquesNode(const rbd::Model& model, const S* x, const S* u, S* y) {
    const std::size_t nq = static_cast<std::size_t>(model.nq), nv = static_cast<std::size_t>(model.nv);
    const std::vector<S> q(x, x + nq), v(x + nq, x + nq + nv), a(u, u + nv);
    const std::vector<S> tau = rbd::Rnea(model, rbd::JointPlacements(model, q), v, a, true);
    for (std::size_t k = 0; k < nv; ++k) y[k] = tau[k];
}

/// M(q) by the composite-rigid-body algorithm   (rbd/quantities/joint_space_inertia_matrix.hpp:42-43)
template <class S>
void InertiaMatrixNode(const rbd::Model& model, const S* x, S* y) {
    const std::size_t nq = static_cast<std::size_t>(model.nq), nv = static_cast<std::size_t>(model.nv);
    const auto M = rbd::Crba(model, rbd::JointPlacements(model, std::vector<S>(x, x + nq)));
    for (std::size_t r = 0; r < nv; ++r)
        for (std::size_t c = 0; c < nv; ++c) y[r * nv + c] = M[r][c];
}

/// M(q)^-1 column by column through the fill-free U D U^T factorisation   (rbd/quantities/joint_space_inertia_matrix_inverse.hpp:42-43)
template <class S>
void InertiaInverseNode(const rbd::Model& model, const S* x, S* y) {
    const std::size_t nq = static_cast<std::size_t>(model.nq), nv = static_cast<std::size_t>(model.nv);
    const auto F = rbd::FactorUdut(rbd::Crba(model, rbd::JointPlacements(model, std::vector<S>(x, x + nq))));
    for (std::size_t c = 0; c < nv; ++c) {
        std::vector<S> e(nv, S{0.0});
        e[c] = S{1.0};
        const std::vector<S> col = rbd::SolveUdut(F, e);
        for (std::size_t r = 0; r < nv; ++r) y[r * nv + c] = col[r];
    }
}

/// World placements oMi of every joint frame.
template <class S>
std::vector<rbd::Xform<S>> WorldPlacements(const rbd::Model& model, const std::vector<rbd::Xform<S>>& liMi) {
    const int n = model.NumJoints();
    std::vector<rbd::Xform<S>> oMi(static_cast<std::size_t>(n));
    for (int i = 1; i < n; ++i) {
        const std::size_t si = static_cast<std::size_t>(i), sp = static_cast<std::size_t>(model.joints[si].parent);
        if (model.joints[si].parent == 0) {
            oMi[si] = liMi[si];
            continue;
        }
        for (std::size_t r = 0; r < 3; ++r) {
            for (std::size_t c = 0; c < 3; ++c) {
                S acc{0.0};
                for (std::size_t k = 0; k < 3; ++k) acc = acc + oMi[sp].R[r][k] * liMi[si].R[k][c];
                oMi[si].R[r][c] = acc;
            }
            S acc = oMi[sp].p[r];
            for (std::size_t k = 0; k < 3; ++k) acc = acc + oMi[sp].R[r][k] * liMi[si].p[k];
            oMi[si].p[r] = acc;
        }
    }
    return oMi;
}

/// Forward kinematics of the four foot frames (lumped into the shanks through fixed joints)   (rbd/quantities/frames.hpp:42-43)
template <class S>
void FootFramesNode(const rbd::Model& model, const S* x, S* y) {
    const std::size_t nq = static_cast<std::size_t>(model.nq);
    const auto oMi = WorldPlacements(model, rbd::JointPlacements(model, std::vector<S>(x, x + nq)));
    for (std::size_t f = 0; f < 4; ++f) {
        const rbd::Frame* frame = nullptr;
        for (const rbd::Frame& fr : model.frames)
            if (fr.name == kFootFrames[f]) frame = &fr;
        if (!frame) throw std::runtime_error(std::string("robot description has no frame ") + kFootFrames[f]);
        const rbd::Xform<S>& W = oMi[static_cast<std::size_t>(frame->joint)];
        for (std::size_t r = 0; r < 3; ++r) {
            S acc = W.p[r];
            for (std::size_t k = 0; k < 3; ++k) acc = acc + W.R[r][k] * frame->placement.p[k];
            y[12 * f + r] = acc;
            for (std::size_t c = 0; c < 3; ++c) {
                S e{0.0};
                for (std::size_t k = 0; k < 3; ++k) e = e + W.R[r][k] * frame->placement.R[k][c];
                y[12 * f + 3 + 3 * r + c] = e;
            }
        }
    }
}

/// Centroidal momentum h_G = [linear; angular about the centre of mass] in world axes
/// (rbd/quantities/centroidal_momentum.hpp:42-43): every body's momentum  Y_i v_i  (spatial inertia about the joint
/// origin times the joint-frame twist) is carried to the world frame and summed; the angular part is then shifted from
/// the world origin to the centre of mass.
template <class S>
void CentroidalMomentumNode(const rbd::Model& model, const S* x, S* y) {
    using namespace rbd;
    const std::size_t nq = static_cast<std::size_t>(model.nq), nv = static_cast<std::size_t>(model.nv);
    const std::vector<S> q(x, x + nq), v(x + nq, x + nq + nv);
    const auto liMi = JointPlacements(model, q);
    const auto oMi = WorldPlacements(model, liMi);
    const int n = model.NumJoints();
    std::vector<Vec6<S>> vel(static_cast<std::size_t>(n));
    for (auto& e : vel[0]) e = S{0.0};
    std::array<S, 3> mc{S{0.0}, S{0.0}, S{0.0}}, lin{S{0.0}, S{0.0}, S{0.0}}, angO{S{0.0}, S{0.0}, S{0.0}};
    double mass = 0.0;
    for (int i = 1; i < n; ++i) {
        const Joint& J = model.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), sp = static_cast<std::size_t>(J.parent);
        vel[si] = JointMotion(J, v);
        if (J.parent > 0) {
            const Vec6<S> vp = detail::ActInvMotion(liMi[si], vel[sp]);
            for (std::size_t k = 0; k < 6; ++k) vel[si][k] = vel[si][k] + vp[k];
        }
        const double m = J.inertia.mass;
        if (m == 0.0) continue;
        mass += m;
        const auto Yd = J.inertia.Matrix();
        Vec6<S> mom;  // [m (v + w x c); h x v + I_origin w] in joint axes, about the joint origin
        for (std::size_t r = 0; r < 6; ++r) {
            S acc{0.0};
            for (std::size_t c = 0; c < 6; ++c)
                if (Yd[r][c] != 0.0) acc = acc + Yd[r][c] * vel[si][c];
            mom[r] = acc;
        }
        const std::array<S, 3> l = detail::RotMul(oMi[si].R, std::array<S, 3>{mom[0], mom[1], mom[2]});
        const std::array<S, 3> k0 = detail::RotMul(oMi[si].R, std::array<S, 3>{mom[3], mom[4], mom[5]});
        const std::array<S, 3> pxl = detail::Cross3(oMi[si].p, l);
        const std::array<S, 3> hW = detail::RotMul(oMi[si].R, std::array<S, 3>{S{J.inertia.h[0]}, S{J.inertia.h[1]}, S{J.inertia.h[2]}});
        for (std::size_t k = 0; k < 3; ++k) {
            mc[k] = mc[k] + m * oMi[si].p[k] + hW[k];
            lin[k] = lin[k] + l[k];
            angO[k] = angO[k] + k0[k] + pxl[k];
        }
    }
    const std::array<S, 3> cG{mc[0] / mass, mc[1] / mass, mc[2] / mass};
    const std::array<S, 3> shift = detail::Cross3(cG, lin);
    for (std::size_t k = 0; k < 3; ++k) {
        y[k] = lin[k];
        y[3 + k] = angO[k] - shift[k];
    }
}

}  // namespace ungar_amd::models
