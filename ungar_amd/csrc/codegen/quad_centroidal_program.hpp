// ungar_amd :: SPMD "lane per leg" program for the centroidal momentum of a floating-base quadruped and its derivative
// (rbd/quantities/centroidal_momentum.hpp:42-43):  h_G = [linear; angular about the centre of mass] in world axes,
// and d h_G / d (q, v).
//
// Momentum is additive over bodies.  In the BASE frame, about the base origin, a leg contributes  h_L = X_0* (Y_0 v_0 + X_1* (Y_1 v_1 + X_2* Y_2 v_2))
// (the backward pass of RNEA with momenta instead of forces) and a first mass moment  s_L = sum_i (m_i p_i + R_i h_i);  one lane per leg computes
// both from (v_b, q_L, v_L).  The lanes meet in ONE quad_sum of twelve numbers, T = [l; K; w; s] = base + sum_L, where the angular momentum about the
// base origin is kept as its two parts K = sum_i R_i k_i (about the bodies' own origins) and w = sum_i p_i x (R_i l_i);  then, identically in the four lanes,
//     h_G = [ R_b l ;  R_b K + cof(R_b) (w - (s / m) x l) ]     (the angular momentum about the centre of mass does not depend on the origin).
// cof(R_b) = R_b for a rotation; the node differentiates with respect to the four RAW quaternion entries, off the unit sphere R_b is not orthogonal, and
// (R a) x (R b) = cof(R) (a x b) is what the world-frame sums of the lane-per-node model (and of the oracle) amount to there.
// Derivatives: F(quaternion, T) is differentiated once with T as auxiliary inputs (6 x (4 + 12)); a column of the lane's own leg (q_L, v_L) is
// dF/dT times the lane-local partial of [h_L; s_L] -- no sum, stored by that lane --, a base-twist column is dF/dT times (base + quad_sum of the
// local partials), a quaternion column is dF/dquaternion, the position columns are zeros.  Pinned in the 4-lane simulator (tests/cpp/quad_rnea_sim.cpp).
#pragma once

#include <algorithm>

#include "quad_leg_program.hpp"

namespace ungar_amd::codegen {

/// pattern: the sparse pattern of the lane-per-node model 'anymal_centroidal' (rows 6 x columns 37).
inline QuadProgram RecordQuadCentroidalProgram(const rbd::Model& model, const tape::SparseEntries& pattern) {
    using namespace rbd;
    using namespace rbd::detail;
    CheckFloatingBaseQuadruped(model);
    constexpr int kRows = 6, kCols = 37;

    QuadProgram P;
    std::vector<int> kOf(static_cast<std::size_t>(kRows * kCols), -1);
    for (std::size_t e = 0; e < pattern.Nnz(); ++e) kOf[static_cast<std::size_t>(pattern.row[e] * kCols + pattern.col[e])] = static_cast<int>(e);
    // ---- inputs: [0,4) quaternion  [4,10) v_b  [10,13) q_L  [13,16) v_L  [16,28) T (auxiliary)  then the leg constants ---------------------------
    constexpr int kQuat = 0, kVb = 4, kQl = 10, kVl = 13, kAux = 16, kT = 12, kConst = kAux + kT;
    const std::vector<LegConstantRef> cref = CollectLegConstants(model, P.constants);
    std::vector<AD> in = tape::Independent(kConst + static_cast<int>(P.constants.size()));
    tape::Graph& g = tape::CurrentGraph();
    for (int i = 0; i < 4; ++i) P.inputNames.push_back("io.qb(" + std::to_string(3 + i) + ")");
    for (int i = 0; i < 6; ++i) P.inputNames.push_back("io.vb(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.ql(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.vl(" + std::to_string(i) + ")");
    for (int i = 0; i < kT; ++i) P.inputNames.push_back("aux_unused");
    for (std::size_t i = 0; i < P.constants.size(); ++i) P.inputNames.push_back("io.c(" + std::to_string(i) + ")");
    P.inputUniform.assign(P.inputNames.size(), 0);
    for (int i = 0; i < 10; ++i) P.inputUniform[static_cast<std::size_t>(i)] = 1;  // quaternion, v_b
    std::size_t cnext = 0;
    auto C = [&]() -> AD {
        const LegConstantRef& r = cref[cnext++];
        return r.literal ? AD{r.value} : in[static_cast<std::size_t>(kConst + r.index)];
    };
    std::array<AD, 3> ql{in[kQl], in[kQl + 1], in[kQl + 2]}, vl{in[kVl], in[kVl + 1], in[kVl + 2]};

    // ---- leg kinematics and inertias (as in the dynamics program) ---------------------------------------------------------------------------
    std::array<Xform<AD>, 3> X;
    std::array<Mat6<AD>, 3> Y;
    std::array<V3, 3> axis;
    std::array<AD, 3> mass;
    std::array<std::array<AD, 3>, 3> hvec;
    for (std::size_t j = 0; j < 3; ++j) {
        axis[j] = model.joints[2 + j].axis;
        std::array<AD, 3> t{C(), C(), C()};
        const AD m = C();
        const std::array<AD, 3> h{C(), C(), C()};
        AD I[3][3];
        for (std::size_t r = 0; r < 3; ++r)
            for (std::size_t c = r; c < 3; ++c) I[r][c] = I[c][r] = C();
        using std::cos;
        using std::sin;
        X[j].R = AxisAngleRotation<AD>(axis[j], cos(ql[j]), sin(ql[j]));
        X[j].p = t;
        mass[j] = m;
        hvec[j] = h;
        const AD hx[3][3] = {{AD{0.0}, -h[2], h[1]}, {h[2], AD{0.0}, -h[0]}, {-h[1], h[0], AD{0.0}}};
        for (std::size_t r = 0; r < 3; ++r)
            for (std::size_t c = 0; c < 3; ++c) {
                Y[j][r][c] = r == c ? m : AD{0.0};
                Y[j][r][3 + c] = -hx[r][c];
                Y[j][3 + r][c] = hx[r][c];
                Y[j][3 + r][3 + c] = I[r][c];
            }
    }
    auto add6 = [](const Vec6<AD>& a, const Vec6<AD>& b) {
        Vec6<AD> r;
        for (std::size_t k = 0; k < 6; ++k) r[k] = a[k] + b[k];
        return r;
    };
    auto scaleS = [&](std::size_t j, const AD& s) { return Vec6<AD>{AD{0.0}, AD{0.0}, AD{0.0}, s * axis[j][0], s * axis[j][1], s * axis[j][2]}; };
    const Vec6<AD> velB{in[kVb], in[kVb + 1], in[kVb + 2], in[kVb + 3], in[kVb + 4], in[kVb + 5]};

    // ---- the leg's momentum about the base origin and its first mass moment, base axes ------------------------------------------------------------
    std::array<Vec6<AD>, 3> vel, mom;
    for (std::size_t j = 0; j < 3; ++j) {
        vel[j] = add6(ActInvMotion(X[j], j == 0 ? velB : vel[j - 1]), scaleS(j, vl[j]));
        mom[j] = MatVec6(Y[j], vel[j]);
    }
    Vec6<AD> hLeg;
    std::array<AD, 3> sAcc = hvec[2], aAcc{mom[2][3], mom[2][4], mom[2][5]};  // first mass moment / angular momenta about the bodies' own origins, from body j down
    AD mAcc = mass[2];
    for (std::size_t j = 3; j-- > 0;) {
        const std::array<AD, 3> own{j > 0 ? mom[j - 1][3] : AD{0.0}, j > 0 ? mom[j - 1][4] : AD{0.0}, j > 0 ? mom[j - 1][5] : AD{0.0}};  // (read before the children are added)
        const Vec6<AD> up = ActForce(X[j], mom[j]);
        if (j > 0) mom[j - 1] = add6(mom[j - 1], up);
        else hLeg = up;
        const std::array<AD, 3> rs = RotMul(X[j].R, sAcc), ra = RotMul(X[j].R, aAcc);  // in the parent's axes (s: about the parent's origin)
        for (std::size_t k = 0; k < 3; ++k) {
            sAcc[k] = rs[k] + mAcc * X[j].p[k] + (j > 0 ? hvec[j - 1][k] : AD{0.0});
            aAcc[k] = ra[k] + own[k];
        }
        if (j > 0) mAcc = mAcc + mass[j - 1];
    }
    const auto Yb = model.joints[1].inertia.Matrix();
    Vec6<AD> momB;
    for (std::size_t r = 0; r < 6; ++r) {
        AD acc{0.0};
        for (std::size_t c = 0; c < 6; ++c)
            if (Yb[r][c] != 0.0) acc = acc + Yb[r][c] * velB[c];
        momB[r] = acc;
    }
    const double totalMass = model.TotalMass();
    std::array<AD, kT> local;  // this leg's [l; K; w; s]
    for (std::size_t k = 0; k < 3; ++k) {
        local[k] = hLeg[k];
        local[3 + k] = aAcc[k];
        local[6 + k] = hLeg[3 + k] - aAcc[k];
        local[9 + k] = sAcc[k];
    }
    std::array<AD, kT> total;  // T: the same in the four lanes (the base contributes l, K and s; its w is zero)
    for (std::size_t k = 0; k < 3; ++k) {
        total[k] = momB[k] + tape::QuadSum(local[k]);
        total[3 + k] = momB[3 + k] + tape::QuadSum(local[3 + k]);
        total[6 + k] = tape::QuadSum(local[6 + k]);
        total[9 + k] = AD{model.joints[1].inertia.h[k]} + tape::QuadSum(local[9 + k]);
    }

    // ---- F(quaternion, T) on auxiliary inputs -------------------------------------------------------------------------------------------------------
    const Rot<AD> Rb = QuaternionToRotation(in[kQuat], in[kQuat + 1], in[kQuat + 2], in[kQuat + 3]);
    auto aux3 = [&](int o) { return std::array<AD, 3>{in[static_cast<std::size_t>(kAux + o)], in[static_cast<std::size_t>(kAux + o + 1)], in[static_cast<std::size_t>(kAux + o + 2)]}; };
    const std::array<AD, 3> lA = aux3(0), KA = aux3(3), wA = aux3(6), sA = aux3(9);
    const std::array<AD, 3> cG{sA[0] / totalMass, sA[1] / totalMass, sA[2] / totalMass};
    const std::array<AD, 3> shift = Cross3(cG, lA);
    const std::array<AD, 3> wG{wA[0] - shift[0], wA[1] - shift[1], wA[2] - shift[2]};
    // cofactor matrix of R_b: columns c2 x c3, c3 x c1, c1 x c2
    auto col = [&](std::size_t c) { return std::array<AD, 3>{Rb[0][c], Rb[1][c], Rb[2][c]}; };
    const std::array<std::array<AD, 3>, 3> cof{Cross3(col(1), col(2)), Cross3(col(2), col(0)), Cross3(col(0), col(1))};
    const std::array<AD, 3> linW = RotMul(Rb, lA), rotK = RotMul(Rb, KA);
    std::array<AD, 3> angW;
    for (std::size_t r = 0; r < 3; ++r) angW[r] = rotK[r] + cof[0][r] * wG[0] + cof[1][r] * wG[1] + cof[2][r] * wG[2];
    std::vector<tape::Id> fIds{linW[0].Node(), linW[1].Node(), linW[2].Node(), angW[0].Node(), angW[1].Node(), angW[2].Node()};

    std::vector<tape::Id> inputIds;
    for (const AD& i : in) inputIds.push_back(i.Node());
    tape::Differentiator diff{g, inputIds};
    std::vector<int> gCols;
    for (int k = 0; k < 4; ++k) gCols.push_back(kQuat + k);
    for (int k = 0; k < kT; ++k) gCols.push_back(kAux + k);
    const tape::SparseEntries G = diff.Jacobian(fIds, gCols, 1);
    std::vector<tape::Id> lIds;
    for (const AD& v : local) lIds.push_back(v.Node());
    std::vector<int> dCols;  // v_b (6), q_L (3), v_L (3)
    for (int k = 0; k < 6; ++k) dCols.push_back(kVb + k);
    for (int k = 0; k < 3; ++k) dCols.push_back(kQl + k);
    for (int k = 0; k < 3; ++k) dCols.push_back(kVl + k);
    const tape::SparseEntries D = diff.Jacobian(lIds, dCols, 1);
    std::vector<std::pair<int, tape::Id>> sub;
    for (int k = 0; k < kT; ++k) sub.emplace_back(kAux + k, total[static_cast<std::size_t>(k)].Node());
    const std::vector<tape::Id> Gv = diff.Substitute(G.value, sub), fv = diff.Substitute(fIds, sub);
    AD Gm[6][4 + kT], Dm[kT][12];
    for (std::size_t e = 0; e < G.Nnz(); ++e) Gm[G.row[e]][G.col[e]] = AD::FromId(Gv[e]);
    for (std::size_t e = 0; e < D.Nnz(); ++e) Dm[D.row[e]][D.col[e]] = AD::FromId(D.value[e]);

    // ---- sinks --------------------------------------------------------------------------------------------------------------------------------
    for (int r = 0; r < 6; ++r) P.slots.push_back({fv[static_cast<std::size_t>(r)], "io.f_base(" + std::to_string(r) + ", %s);"});
    auto chain = [&](int r, const std::array<AD, kT>& dT) {
        AD v{0.0};
        for (int t = 0; t < kT; ++t) v = v + Gm[r][4 + t] * dT[static_cast<std::size_t>(t)];
        return v;
    };
    auto sharedSink = [&](int r, int col, const AD& v) {
        if (kOf[static_cast<std::size_t>(r * kCols + col)] < 0 && !(v.IsLiteral() && v.Literal() == 0.0))
            throw std::runtime_error("quad centroidal program: non-zero entry outside the sparsity pattern at (" + std::to_string(r) + "," + std::to_string(col) + ")");
        P.slots.push_back({v.Node(), "io.j_base_shared(" + std::to_string(r) + ", " + std::to_string(col) + ", " + std::to_string(kOf[static_cast<std::size_t>(r * kCols + col)]) + ", %s);"});
    };
    // node-level columns: x = [p 0..2 | quat 3..6 | q_leg 7 + 3 L + k | v_b 19..24 | v_leg 25 + 3 L + k]
    for (int kind = 0; kind < 2; ++kind)  // columns owned by this lane's leg: q_L, v_L
        for (int k = 0; k < 3; ++k) {
            P.phaseStarts.push_back(P.slots.size());
            const int lc = 6 + 3 * kind + k, colBase = (kind == 0 ? 7 : 25) + k;
            std::array<AD, kT> dT;
            for (int t = 0; t < kT; ++t) dT[static_cast<std::size_t>(t)] = Dm[t][lc];
            for (int r = 0; r < 6; ++r) {
                const AD v = chain(r, dT);
                std::string ks;
                std::array<int, 4> kk{};
                for (int L = 0; L < 4; ++L) {
                    kk[static_cast<std::size_t>(L)] = kOf[static_cast<std::size_t>(r * kCols + colBase + 3 * L)];
                    ks += (L ? ", " : "") + std::to_string(kk[static_cast<std::size_t>(L)]);
                }
                if (*std::min_element(kk.begin(), kk.end()) < 0 && !(v.IsLiteral() && v.Literal() == 0.0))
                    throw std::runtime_error("quad centroidal program: non-zero entry outside the sparsity pattern");
                if (*std::min_element(kk.begin(), kk.end()) >= 0) {
                    const std::array<int, 4> delta{0, kk[1] - kk[0], kk[2] - kk[0], kk[3] - kk[0]};
                    if (std::find(P.sparseDeltas.begin(), P.sparseDeltas.end(), delta) == P.sparseDeltas.end()) P.sparseDeltas.push_back(delta);
                }
                P.slots.push_back({v.Node(), "io.j_base_own(" + std::to_string(r) + ", " + std::to_string(colBase) + ", " + ks + ", %s);"});
            }
        }
    P.phaseStarts.push_back(P.slots.size());
    for (int k = 0; k < 3; ++k)  // position: zeros
        for (int r = 0; r < 6; ++r) sharedSink(r, k, AD{0.0});
    for (int k = 0; k < 4; ++k)  // quaternion: dF / dquaternion
        for (int r = 0; r < 6; ++r) sharedSink(r, 3 + k, Gm[r][k]);
    for (int k = 0; k < 6; ++k) {  // base twist: dF/dT (d base + quad_sum of the local partials)
        P.phaseStarts.push_back(P.slots.size());
        std::array<AD, kT> dT;
        for (int t = 0; t < kT; ++t) dT[static_cast<std::size_t>(t)] = (t < 6 ? AD{Yb[static_cast<std::size_t>(t)][static_cast<std::size_t>(k)]} : AD{0.0}) + tape::QuadSum(Dm[t][k]);
        for (int r = 0; r < 6; ++r) sharedSink(r, 19 + k, chain(r, dT));
    }

    std::vector<AD> roots;
    for (const auto& sl : P.slots) roots.push_back(AD::FromId(sl.value));
    P.tape = tape::MakeTape(roots);
    for (std::size_t i = 0; i < P.slots.size(); ++i) P.slots[i].value = P.tape.outputs[i];
    return P;
}

}  // namespace ungar_amd::codegen
