// ungar_amd :: the rigid-body quantities available on a Robot (reference include/ungar/rbd/quantities/*.hpp).
//
//   quantity                              At(...)      reference algorithm -> data member
//   generalized_accelerations             q, v, tau    pinocchio::aba                      -> ddq    (hot-path row A7)
//   joint_torques                         q, v, a      pinocchio::rnea                     -> tau
//   nonlinear_effects                     q, v         pinocchio::nonLinearEffects         -> nle
//   generalized_gravity                   q            pinocchio::computeGeneralizedGravity-> g
//   joint_space_inertia_matrix            q            pinocchio::crba                     -> M      (full symmetric here)
//   joint_space_inertia_matrix_inverse    q            pinocchio::computeMinverse          -> Minv
//   com_position / _velocity / _acceleration  q[, v[, a]]  pinocchio::centerOfMass         -> com / vcom / acom (world frame)
//   kinetic_energy                        q, v         pinocchio::computeKineticEnergy     -> kinetic_energy
//   potential_energy                      q            pinocchio::computePotentialEnergy   -> potential_energy
//   frames                                q            pinocchio::framesForwardKinematics  -> oMf (one RBD::Pose per link, Model().frameNames)
//   centroidal_momentum                   q, v         pinocchio::ccrba                    -> hg (6-vector [linear; angular] at the CoM)
//   centroidal_momentum_matrix            q            pinocchio::crbaMinimal              -> Ag (6 x nv, h_G = A_G v)
//   composite_rigid_body_inertia          q[, v]       pinocchio::ccrba                    -> Ig (6 x 6 [m 1, 0; 0, I_G], a matrix
//                                                                                                instead of a pinocchio::Inertia)
// Return types differ where the reference hands out Pinocchio spatial-algebra objects (they return Pinocchio spatial-algebra
// objects).
#pragma once

#include "../quantity.hpp"

namespace Ungar {
namespace RBD {
namespace Quantities {
UNGAR_MAKE_QUANTITY(generalized_accelerations);
UNGAR_MAKE_QUANTITY(joint_torques);
UNGAR_MAKE_QUANTITY(nonlinear_effects);
UNGAR_MAKE_QUANTITY(generalized_gravity);
UNGAR_MAKE_QUANTITY(joint_space_inertia_matrix);
UNGAR_MAKE_QUANTITY(joint_space_inertia_matrix_inverse);
UNGAR_MAKE_QUANTITY(com_position);
UNGAR_MAKE_QUANTITY(com_velocity);
UNGAR_MAKE_QUANTITY(com_acceleration);
UNGAR_MAKE_QUANTITY(kinetic_energy);
UNGAR_MAKE_QUANTITY(potential_energy);
UNGAR_MAKE_QUANTITY(frames);
UNGAR_MAKE_QUANTITY(centroidal_momentum);
UNGAR_MAKE_QUANTITY(centroidal_momentum_matrix);
UNGAR_MAKE_QUANTITY(composite_rigid_body_inertia);
}  // namespace Quantities

UNGAR_MAKE_GETTER(generalized_accelerations, ddq);
UNGAR_MAKE_GETTER(joint_torques, tau);
UNGAR_MAKE_GETTER(nonlinear_effects, nle);
UNGAR_MAKE_GETTER(generalized_gravity, g);
UNGAR_MAKE_GETTER(joint_space_inertia_matrix, M);
UNGAR_MAKE_GETTER(joint_space_inertia_matrix_inverse, Minv);
UNGAR_MAKE_GETTER(com_position, com);
UNGAR_MAKE_GETTER(com_velocity, vcom);
UNGAR_MAKE_GETTER(com_acceleration, acom);
UNGAR_MAKE_GETTER(kinetic_energy, kinetic_energy);
UNGAR_MAKE_GETTER(potential_energy, potential_energy);
UNGAR_MAKE_GETTER(frames, oMf);
UNGAR_MAKE_GETTER(centroidal_momentum, hg);
UNGAR_MAKE_GETTER(centroidal_momentum_matrix, Ag);
UNGAR_MAKE_GETTER(composite_rigid_body_inertia, Ig);

#define UNGAR_RBD_EVALUATOR_MEMBERS      \
    const ::Ungar::RBD::ModelInfo& model; \
    ::Ungar::RBD::Data<S>& data

template <class S>
struct Evaluator<Quantities::generalized_accelerations, S> {
    void At(const auto& q, const auto& v, const auto& tau) {
        data.ddq = Internal::ToVector(::ungar_amd::rbd::Aba(model.impl, Internal::ToStd<S>(q), Internal::ToStd<S>(v), Internal::ToStd<S>(tau)));
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::joint_torques, S> {
    void At(const auto& q, const auto& v, const auto& a) {
        namespace rbd = ::ungar_amd::rbd;
        data.tau = Internal::ToVector(rbd::Rnea(model.impl, rbd::JointPlacements(model.impl, Internal::ToStd<S>(q)), Internal::ToStd<S>(v), Internal::ToStd<S>(a)));
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::nonlinear_effects, S> {
    void At(const auto& q, const auto& v) {
        namespace rbd = ::ungar_amd::rbd;
        const std::vector<S> zero(static_cast<std::size_t>(model.nv), S{0.0});
        data.nle = Internal::ToVector(rbd::Rnea(model.impl, rbd::JointPlacements(model.impl, Internal::ToStd<S>(q)), Internal::ToStd<S>(v), zero));
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::generalized_gravity, S> {
    void At(const auto& q) {
        namespace rbd = ::ungar_amd::rbd;
        const std::vector<S> zero(static_cast<std::size_t>(model.nv), S{0.0});
        data.g = Internal::ToVector(rbd::Rnea(model.impl, rbd::JointPlacements(model.impl, Internal::ToStd<S>(q)), zero, zero));
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::joint_space_inertia_matrix, S> {
    void At(const auto& q) {
        namespace rbd = ::ungar_amd::rbd;
        const auto M = rbd::Crba(model.impl, rbd::JointPlacements(model.impl, Internal::ToStd<S>(q)));
        data.M.resize(model.nv, model.nv);
        for (int r = 0; r < model.nv; ++r)
            for (int c = 0; c < model.nv; ++c) data.M(r, c) = M[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)];
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::joint_space_inertia_matrix_inverse, S> {
    void At(const auto& q) {
        namespace rbd = ::ungar_amd::rbd;
        const auto F = rbd::FactorUdut(rbd::Crba(model.impl, rbd::JointPlacements(model.impl, Internal::ToStd<S>(q))));
        data.Minv.resize(model.nv, model.nv);
        for (int c = 0; c < model.nv; ++c) {
            std::vector<S> e(static_cast<std::size_t>(model.nv), S{0.0});
            e[static_cast<std::size_t>(c)] = S{1.0};
            const std::vector<S> col = rbd::SolveUdut(F, e);
            for (int r = 0; r < model.nv; ++r) data.Minv(r, c) = col[static_cast<std::size_t>(r)];
        }
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

namespace Internal {
/// World-frame position / velocity / acceleration of the centre of mass (accelerations WITHOUT gravity,
/// as pinocchio::centerOfMass): classical point kinematics of every body's centre of mass.
template <class S>
void CenterOfMass(const ModelInfo& model, Data<S>& data, const std::vector<S>& q, const std::vector<S>* v, const std::vector<S>* a, bool momentum = false) {
    namespace rbd = ::ungar_amd::rbd;
    using V3s = std::array<S, 3>;
    const rbd::Model& m = model.impl;
    const int n = m.NumJoints();
    const auto liMi = rbd::JointPlacements(m, q);
    std::vector<rbd::Xform<S>> oMi(static_cast<std::size_t>(n));
    std::vector<rbd::Vec6<S>> vel(static_cast<std::size_t>(n)), acc(static_cast<std::size_t>(n));
    for (auto& e : vel[0]) e = S{0.0};
    for (auto& e : acc[0]) e = S{0.0};
    V3s com{S{0.0}, S{0.0}, S{0.0}}, vcom = com, acom = com;
    V3s angularAtOrigin{S{0.0}, S{0.0}, S{0.0}};  // sum_i [ c_i x m_i v_ci + R_i I_ci omega_i ], world frame, about the world origin
    const double total = m.TotalMass();
    auto cross = [](const V3s& x, const V3s& y) { return V3s{x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]}; };
    for (int i = 1; i < n; ++i) {
        const rbd::Joint& J = m.joints[static_cast<std::size_t>(i)];
        const std::size_t si = static_cast<std::size_t>(i), sp = static_cast<std::size_t>(J.parent);
        rbd::Xform<S>& W = oMi[si];
        if (J.parent == 0) W = liMi[si];
        else {
            for (std::size_t r = 0; r < 3; ++r) {
                for (std::size_t c = 0; c < 3; ++c) {
                    S acc3{0.0};
                    for (std::size_t k = 0; k < 3; ++k) acc3 = acc3 + oMi[sp].R[r][k] * liMi[si].R[k][c];
                    W.R[r][c] = acc3;
                }
                S accp = oMi[sp].p[r];
                for (std::size_t k = 0; k < 3; ++k) accp = accp + oMi[sp].R[r][k] * liMi[si].p[k];
                W.p[r] = accp;
            }
        }
        const double mass = J.inertia.mass;
        if (mass == 0.0 && !v) continue;
        const V3s c{S{mass ? J.inertia.h[0] / mass : 0.0}, S{mass ? J.inertia.h[1] / mass : 0.0}, S{mass ? J.inertia.h[2] / mass : 0.0}};
        const V3s wc = rbd::detail::RotMul(W.R, c);
        for (std::size_t k = 0; k < 3; ++k) com[k] = com[k] + (mass / total) * (W.p[k] + wc[k]);
        if (v) {
            // spatial velocity / acceleration in the local frame (RNEA's forward pass without gravity)
            const rbd::Vec6<S> vJ = rbd::JointMotion(J, *v);
            vel[si] = rbd::detail::ActInvMotion(liMi[si], vel[sp]);
            for (std::size_t k = 0; k < 6; ++k) vel[si][k] = vel[si][k] + vJ[k];
            const V3s lin{vel[si][0], vel[si][1], vel[si][2]}, ang{vel[si][3], vel[si][4], vel[si][5]};
            const V3s wxc = cross(ang, c);
            const V3s vc = rbd::detail::RotMul(W.R, V3s{lin[0] + wxc[0], lin[1] + wxc[1], lin[2] + wxc[2]});
            for (std::size_t k = 0; k < 3; ++k) vcom[k] = vcom[k] + (mass / total) * vc[k];
            if (momentum && mass > 0.0) {
                // rotational inertia about the body's centre of mass, joint frame: I_c = I_origin - m (|c|^2 1 - c c^T)
                const auto& I = J.inertia.I;
                const double cd[3] = {J.inertia.h[0] / mass, J.inertia.h[1] / mass, J.inertia.h[2] / mass};
                const double cc = cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2];
                V3s Iw{S{0.0}, S{0.0}, S{0.0}};
                for (std::size_t r = 0; r < 3; ++r)
                    for (std::size_t cidx = 0; cidx < 3; ++cidx)
                        Iw[r] = Iw[r] + (I[r][cidx] - mass * ((r == cidx ? cc : 0.0) - cd[r] * cd[cidx])) * ang[cidx];
                const V3s spin = rbd::detail::RotMul(W.R, Iw);
                const V3s cw{W.p[0] + wc[0], W.p[1] + wc[1], W.p[2] + wc[2]};
                const V3s orbital = cross(cw, V3s{mass * vc[0], mass * vc[1], mass * vc[2]});
                for (std::size_t k = 0; k < 3; ++k) angularAtOrigin[k] = angularAtOrigin[k] + spin[k] + orbital[k];
            }
            if (a) {
                const rbd::Vec6<S> aJ = rbd::JointMotion(J, *a);
                acc[si] = rbd::detail::ActInvMotion(liMi[si], acc[sp]);
                const rbd::Vec6<S> bias = rbd::detail::CrossMotion(vel[si], vJ);
                for (std::size_t k = 0; k < 6; ++k) acc[si][k] = acc[si][k] + aJ[k] + bias[k];
                // classical acceleration of the point c: a_lin + alpha x c + omega x (v_lin + omega x c)
                const V3s al{acc[si][0], acc[si][1], acc[si][2]}, aa{acc[si][3], acc[si][4], acc[si][5]};
                const V3s axc = cross(aa, c);
                const V3s wxv = cross(ang, V3s{lin[0] + wxc[0], lin[1] + wxc[1], lin[2] + wxc[2]});
                const V3s ac = rbd::detail::RotMul(W.R, V3s{al[0] + axc[0] + wxv[0], al[1] + axc[1] + wxv[1], al[2] + axc[2] + wxv[2]});
                for (std::size_t k = 0; k < 3; ++k) acom[k] = acom[k] + (mass / total) * ac[k];
            }
        }
    }
    data.com = Vector3<S>{com[0], com[1], com[2]};
    if (v) data.vcom = Vector3<S>{vcom[0], vcom[1], vcom[2]};
    if (momentum) {  // h_G = [m v_com ; L_origin - com x m v_com]: expressed at the centre of mass, world-aligned axes
        const V3s lin{total * vcom[0], total * vcom[1], total * vcom[2]};
        const V3s shift = cross(com, lin);
        data.hg.resize(6);
        for (std::size_t k = 0; k < 3; ++k) {
            data.hg[static_cast<index_t>(k)] = lin[k];
            data.hg[static_cast<index_t>(3 + k)] = angularAtOrigin[k] - shift[k];
        }
    }
    if (a) data.acom = Vector3<S>{acom[0], acom[1], acom[2]};
}
}  // namespace Internal

template <class S>
struct Evaluator<Quantities::com_position, S> {
    void At(const auto& q) {
        Internal::CenterOfMass<S>(model, data, Internal::ToStd<S>(q), nullptr, nullptr);
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};
template <class S>
struct Evaluator<Quantities::com_velocity, S> {
    void At(const auto& q, const auto& v) {
        const std::vector<S> vs = Internal::ToStd<S>(v);
        Internal::CenterOfMass<S>(model, data, Internal::ToStd<S>(q), &vs, nullptr);
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};
template <class S>
struct Evaluator<Quantities::centroidal_momentum, S> {
    /// [linear; angular] momentum about the centre of mass, world-aligned axes (pinocchio::ccrba -> data.hg).
    void At(const auto& q, const auto& v) {
        const std::vector<S> vs = Internal::ToStd<S>(v);
        Internal::CenterOfMass<S>(model, data, Internal::ToStd<S>(q), &vs, nullptr, true);
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};
template <class S>
struct Evaluator<Quantities::centroidal_momentum_matrix, S> {
    /// A_G(q), 6 x nv with h_G = A_G v (pinocchio::computeCentroidalMap / crbaMinimal -> data.Ag): the momentum is
    /// linear in v, so column j is the centroidal momentum of the unit velocity e_j.
    void At(const auto& q) {
        const std::vector<S> qs = Internal::ToStd<S>(q);
        data.Ag.resize(6, model.nv);
        for (int j = 0; j < model.nv; ++j) {
            std::vector<S> e(static_cast<std::size_t>(model.nv), S{0.0});
            e[static_cast<std::size_t>(j)] = S{1.0};
            Internal::CenterOfMass<S>(model, data, qs, &e, nullptr, true);
            for (int r = 0; r < 6; ++r) data.Ag(r, j) = data.hg[r];
        }
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};
template <class S>
struct Evaluator<Quantities::composite_rigid_body_inertia, S> {
    /// Inertia of the whole robot frozen in configuration q, about its centre of mass, world axes, as a 6 x 6
    /// spatial inertia [m 1, 0; 0, I_G] (pinocchio::ccrba -> data.Ig; the velocity argument is accepted and unused):
    /// its angular block maps a rigid angular velocity of the frozen robot to the angular centroidal momentum,
    /// i.e. it is the block of A_G that multiplies the base angular velocity, rotated to world axes.
    void At(const auto& q, const auto&... /*v*/) {
        const std::vector<S> qs = Internal::ToStd<S>(q);
        const auto R = ::ungar_amd::rbd::detail::QuaternionToRotation(qs[3], qs[4], qs[5], qs[6]);  // base orientation
        data.Ig.resize(6, 6);
        for (int j = 0; j < 3; ++j) {  // base angular velocity = R^T e_j (body frame) <=> world angular velocity e_j
            std::vector<S> v(static_cast<std::size_t>(model.nv), S{0.0});
            for (int k = 0; k < 3; ++k) v[static_cast<std::size_t>(3 + k)] = R[static_cast<std::size_t>(j)][static_cast<std::size_t>(k)];
            Internal::CenterOfMass<S>(model, data, qs, &v, nullptr, true);
            // a rotation about the world origin also translates the centre of mass; the angular momentum ABOUT the
            // centre of mass of a rigid rotation is I_G omega regardless of the pivot
            for (int r = 0; r < 3; ++r) data.Ig(3 + r, 3 + j) = data.hg[3 + r];
        }
        const double mass = model.impl.TotalMass();
        for (int r = 0; r < 3; ++r) data.Ig(r, r) = S{mass};
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};
template <class S>
struct Evaluator<Quantities::com_acceleration, S> {
    void At(const auto& q, const auto& v, const auto& a) {
        const std::vector<S> vs = Internal::ToStd<S>(v), as = Internal::ToStd<S>(a);
        Internal::CenterOfMass<S>(model, data, Internal::ToStd<S>(q), &vs, &as);
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::frames, S> {
    /// Forward kinematics of every frame: oMf = oMi(supporting joint) * placement of the frame in that joint.
    void At(const auto& q) {
        namespace rbd = ::ungar_amd::rbd;
        const rbd::Model& m = model.impl;
        const auto liMi = rbd::JointPlacements(m, Internal::ToStd<S>(q));
        const int n = m.NumJoints();
        std::vector<rbd::Xform<S>> oMi(static_cast<std::size_t>(n));
        for (std::size_t r = 0; r < 3; ++r) {  // the universe: identity (its frame is frame 0)
            oMi[0].p[r] = S{0.0};
            for (std::size_t c = 0; c < 3; ++c) oMi[0].R[r][c] = S{r == c ? 1.0 : 0.0};
        }
        for (int i = 1; i < n; ++i) {
            const std::size_t si = static_cast<std::size_t>(i), sp = static_cast<std::size_t>(m.joints[si].parent);
            if (m.joints[si].parent == 0) {
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
        data.oMf.resize(m.frames.size());
        for (std::size_t f = 0; f < m.frames.size(); ++f) {
            const rbd::Frame& fr = m.frames[f];
            const rbd::Xform<S>& W = oMi[static_cast<std::size_t>(fr.joint)];
            Pose<S>& out = data.oMf[f];
            for (std::size_t r = 0; r < 3; ++r) {
                for (std::size_t c = 0; c < 3; ++c) {
                    S acc{0.0};
                    for (std::size_t k = 0; k < 3; ++k) acc = acc + W.R[r][k] * fr.placement.R[k][c];
                    out.rotationMatrix[r][c] = acc;
                }
                S acc = W.p[r];
                for (std::size_t k = 0; k < 3; ++k) acc = acc + W.R[r][k] * fr.placement.p[k];
                out.position[static_cast<index_t>(r)] = acc;
            }
        }
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::kinetic_energy, S> {
    /// 1/2 v^T M(q) v
    void At(const auto& q, const auto& v) {
        namespace rbd = ::ungar_amd::rbd;
        const auto M = rbd::Crba(model.impl, rbd::JointPlacements(model.impl, Internal::ToStd<S>(q)));
        S e{0.0};
        for (int r = 0; r < model.nv; ++r)
            for (int c = 0; c < model.nv; ++c) e = e + 0.5 * S{v[r]} * M[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)] * S{v[c]};
        data.kinetic_energy = e;
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

template <class S>
struct Evaluator<Quantities::potential_energy, S> {
    /// -m_total g . com
    void At(const auto& q) {
        Internal::CenterOfMass<S>(model, data, Internal::ToStd<S>(q), nullptr, nullptr);
        const auto& g = model.impl.gravity;
        data.potential_energy = -model.impl.TotalMass() * (g[0] * data.com[0] + g[1] * data.com[1] + g[2] * data.com[2]);
    }
    UNGAR_RBD_EVALUATOR_MEMBERS;
};

}  // namespace RBD
}  // namespace Ungar
