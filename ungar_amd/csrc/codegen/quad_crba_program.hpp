// ungar_amd :: SPMD "lane per leg" program for the joint-space inertia matrix of a floating-base quadruped and its derivative
// (rbd/quantities/joint_space_inertia_matrix.hpp:42-43):  M(q), 18 x 18 row-major, and d M / d q.
//
// M is block-arrow and does not depend on the base pose: the base block is Y_b + sum_L X_L* Yc_L X_L^-1 (the legs' composite inertias carried
// to the base), the base-leg blocks M_bL and the leg blocks M_LL depend on q_L alone, leg-leg blocks of different legs are zero.  One lane per
// leg computes Yc_L, M_LL, M_bL and its summand of the base block from ITS three joint angles; the lanes meet in one quad_sum (the base
// block).  d M / d q has columns of the joint angles only, and every one of them is owned by one leg: the lane differentiates its local
// function (tape differentiator, three forward columns) and stores d(M_LL), d(M_bL) (both orientations) and d(its summand of the base block)
// -- no sums.  Every sink carries the CSR index of its entry per leg (the model's pattern, 916 of 324 x 19); the dense 324 x 19 block -- 85 %
// structural zeros -- stays with the lane-per-node kernel.  Pinned in a 4-lane CPU simulator (tests/cpp/quad_crba_sim.cpp).
#pragma once

#include <algorithm>

#include "quad_leg_program.hpp"

namespace ungar_amd::codegen {

/// pattern: the sparse pattern of the lane-per-node model 'anymal_crba' (rows 324 x columns 19).
inline QuadProgram RecordQuadCrbaProgram(const rbd::Model& model, const tape::SparseEntries& pattern) {
    using namespace rbd;
    using namespace rbd::detail;
    CheckFloatingBaseQuadruped(model);
    constexpr int kOut = 324, kCols = 19;

    QuadProgram P;
    std::vector<int> kOf(static_cast<std::size_t>(kOut * kCols), -1);
    for (std::size_t e = 0; e < pattern.Nnz(); ++e) kOf[static_cast<std::size_t>(pattern.row[e] * kCols + pattern.col[e])] = static_cast<int>(e);
    const std::vector<LegConstantRef> cref = CollectLegConstants(model, P.constants);
    constexpr int kQl = 0, kConst = 3;
    std::vector<AD> in = tape::Independent(kConst + static_cast<int>(P.constants.size()));
    tape::Graph& g = tape::CurrentGraph();
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.ql(" + std::to_string(i) + ")");
    for (std::size_t i = 0; i < P.constants.size(); ++i) P.inputNames.push_back("io.c(" + std::to_string(i) + ")");
    P.inputUniform.assign(P.inputNames.size(), 0);
    std::size_t cnext = 0;
    auto C = [&]() -> AD {
        const LegConstantRef& r = cref[cnext++];
        return r.literal ? AD{r.value} : in[static_cast<std::size_t>(kConst + r.index)];
    };
    std::array<AD, 3> ql{in[kQl], in[kQl + 1], in[kQl + 2]};

    // ---- leg kinematics and inertias (as in the dynamics program) ---------------------------------------------------------------------------
    std::array<Xform<AD>, 3> X;
    std::array<Mat6<AD>, 3> Y;
    std::array<V3, 3> axis;
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
        const AD hx[3][3] = {{AD{0.0}, -h[2], h[1]}, {h[2], AD{0.0}, -h[0]}, {-h[1], h[0], AD{0.0}}};
        for (std::size_t r = 0; r < 3; ++r)
            for (std::size_t c = 0; c < 3; ++c) {
                Y[j][r][c] = r == c ? m : AD{0.0};
                Y[j][r][3 + c] = -hx[r][c];
                Y[j][3 + r][c] = hx[r][c];
                Y[j][3 + r][3 + c] = I[r][c];
            }
    }
    const auto Yb = model.joints[1].inertia.Matrix();
    auto Sjoint = [&](std::size_t j) { return Vec6<AD>{AD{0.0}, AD{0.0}, AD{0.0}, AD{axis[j][0]}, AD{axis[j][1]}, AD{axis[j][2]}}; };
    auto dotS = [&](std::size_t j, const Vec6<AD>& f) { return f[3] * axis[j][0] + f[4] * axis[j][1] + f[5] * axis[j][2]; };

    // ---- CRBA restricted to one leg ----------------------------------------------------------------------------------------------------------
    std::array<Mat6<AD>, 3> Yc = Y;
    for (std::size_t j = 2; j >= 1; --j) {
        const Mat6<AD> T = TransportInertia(X[j], Yc[j]);
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Yc[j - 1][r][c] = Yc[j - 1][r][c] + T[r][c];
    }
    const Mat6<AD> YcbLeg = TransportInertia(X[0], Yc[0]);
    AD MLL[3][3], MbL[6][3];
    for (std::size_t j = 0; j < 3; ++j) {
        Vec6<AD> F = MatVec6(Yc[j], Sjoint(j));
        MLL[j][j] = dotS(j, F);
        for (std::size_t k = j; k-- > 0;) {
            F = ActForce(X[k + 1], F);
            MLL[k][j] = MLL[j][k] = dotS(k, F);
        }
        F = ActForce(X[0], F);
        for (std::size_t r = 0; r < 6; ++r) MbL[r][j] = F[r];
    }

    // ---- sinks: an output of the local function is one entry of M per leg (or the same base entry in all four) ---------------------------------------
    // kinds: 0 base block (r, c);  1 M[r][6 + 3 L + j];  2 M[6 + 3 L + j][r];  3 M[6 + 3 L + i][6 + 3 L + j]
    struct Local {
        AD value;   // lane-local value (kind 0: this leg's summand)
        int kind, a, b;
    };
    std::vector<Local> locals;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) locals.push_back({YcbLeg[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)], 0, r, c});
    for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 3; ++j) {
            locals.push_back({MbL[r][j], 1, r, j});
            locals.push_back({MbL[r][j], 2, j, r});
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) locals.push_back({MLL[i][j], 3, i, j});
    auto outIndex = [](const Local& l, int L) {
        switch (l.kind) {
            case 0: return l.a * 18 + l.b;
            case 1: return l.a * 18 + 6 + 3 * L + l.b;
            case 2: return (6 + 3 * L + l.a) * 18 + l.b;
            default: return (6 + 3 * L + l.a) * 18 + 6 + 3 * L + l.b;
        }
    };
    // values
    for (const Local& l : locals) {
        const std::string ab = std::to_string(l.a) + ", " + std::to_string(l.b);
        if (l.kind == 0) P.slots.push_back({(AD{Yb[static_cast<std::size_t>(l.a)][static_cast<std::size_t>(l.b)]} + tape::QuadSum(l.value)).Node(), "io.f_base(" + std::to_string(l.a * 18 + l.b) + ", %s);"});
        else if (l.kind == 1) P.slots.push_back({l.value.Node(), "io.f_bl(" + ab + ", %s);"});
        else if (l.kind == 2) P.slots.push_back({l.value.Node(), "io.f_lb(" + ab + ", %s);"});
        else {
            P.slots.push_back({l.value.Node(), "io.f_ll(" + ab + ", 0, %s);"});
            for (int rot = 1; rot < 4; ++rot) P.slots.push_back({AD{0.0}.Node(), "io.f_ll(" + ab + ", " + std::to_string(rot) + ", %s);"});  // blocks of two different legs: zeros
        }
    }
    // derivatives with respect to the lane's own joint angles (columns 7 + 3 L + k of the node)
    std::vector<tape::Id> inputIds;
    for (const AD& i : in) inputIds.push_back(i.Node());
    tape::Differentiator diff{g, inputIds};
    std::vector<tape::Id> outs;
    for (const Local& l : locals) outs.push_back(l.value.Node());
    const tape::SparseEntries D = diff.Jacobian(outs, std::vector<int>{kQl, kQl + 1, kQl + 2}, 1);
    std::vector<std::array<AD, 3>> Dm(locals.size());
    for (std::size_t e = 0; e < D.Nnz(); ++e) Dm[static_cast<std::size_t>(D.row[e])][static_cast<std::size_t>(D.col[e])] = AD::FromId(D.value[e]);
    std::vector<char> written(pattern.Nnz(), 0);
    for (int k = 0; k < 3; ++k) {
        P.phaseStarts.push_back(P.slots.size());
        for (std::size_t o = 0; o < locals.size(); ++o) {
            std::array<int, 4> ks{};
            for (int L = 0; L < 4; ++L) ks[static_cast<std::size_t>(L)] = kOf[static_cast<std::size_t>(outIndex(locals[o], L) * kCols + 7 + 3 * L + k)];
            const AD v = Dm[o][static_cast<std::size_t>(k)];
            const bool literalZero = v.IsLiteral() && v.Literal() == 0.0;
            if (*std::max_element(ks.begin(), ks.end()) < 0) {
                if (!literalZero) throw std::runtime_error("quad crba program: non-zero entry outside the sparsity pattern");
                continue;
            }
            if (*std::min_element(ks.begin(), ks.end()) < 0 && !literalZero) throw std::runtime_error("quad crba program: the pattern differs between legs");
            for (int kk : ks)
                if (kk >= 0) written[static_cast<std::size_t>(kk)] = 1;
            if (*std::min_element(ks.begin(), ks.end()) >= 0) {
                const std::array<int, 4> delta{0, ks[1] - ks[0], ks[2] - ks[0], ks[3] - ks[0]};
                if (std::find(P.sparseDeltas.begin(), P.sparseDeltas.end(), delta) == P.sparseDeltas.end()) P.sparseDeltas.push_back(delta);
            }
            P.slots.push_back({v.Node(), "io.j_sparse(" + std::to_string(ks[0]) + ", " + std::to_string(ks[1]) + ", " + std::to_string(ks[2]) + ", " + std::to_string(ks[3]) + ", %s);"});
        }
    }
    for (std::size_t e = 0; e < written.size(); ++e)
        if (!written[e]) throw std::runtime_error("quad crba program: pattern entry (" + std::to_string(pattern.row[e]) + ", " + std::to_string(pattern.col[e]) + ") has no sink");

    std::vector<AD> roots;
    for (const auto& sl : P.slots) roots.push_back(AD::FromId(sl.value));
    P.tape = tape::MakeTape(roots);
    for (std::size_t i = 0; i < P.slots.size(); ++i) P.slots[i].value = P.tape.outputs[i];
    return P;
}

}  // namespace ungar_amd::codegen
