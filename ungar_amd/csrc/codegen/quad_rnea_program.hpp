// ungar_amd :: SPMD "lane per leg" program for the joint torques of a floating-base quadruped and their derivatives
// (rbd/quantities/joint_torques.hpp:42-43):  tau = RNEA(q, v, a),  d tau / d (q, v, a).
//
// The lane-per-node kernel lowered from the tape of csrc/models/rbd_nodes.hpp keeps the whole tree alive in one lane (2-4 KB of scratch,
// 0.8 TB/s written).  RNEA has the same structure as the dynamics node of quad_leg_program.hpp, without its solves: the four legs meet
// only in the base wrench.  One lane per leg:
//   forward recursion down ITS leg from the (lane-uniform) base twist and base acceleration a_b + R_b^T (-g);
//   backward recursion up to the leg's wrench on the base, f_L;   tau_base = f_own(base) + quad_sum(f_L),  tau_L = S^T f.
// Derivatives come from the tape's differentiator applied to the lane-local function (tau_L, f_L, f_own) of
// (quaternion, v_b, a_b | q_L, v_L, a_L):
//   columns of the lane's own leg:  its 3 leg rows and the 6 base rows (d f_L; no other leg depends on them: exact zeros there);
//   shared columns (quaternion, v_b, a_b):  leg rows per lane, base rows = d f_own + quad_sum(d f_L) -- the same value in the four lanes;
//   position columns: zeros.
// Every lane stores its leg's 3 rows of all 55 columns and the 6 base rows of its 9 columns; the base rows of the 19 shared columns are
// stored by all four lanes (same address, same value: merged inside the store instruction).  The program is recorded on the tape like the
// dynamics program, emitted generic over the value type, and pinned in a 4-lane CPU simulator (tests/cpp/quad_rnea_sim.cpp).
#pragma once

#include <algorithm>

#include "quad_leg_program.hpp"

namespace ungar_amd::codegen {

/// pattern: the sparse pattern of the lane-per-node model 'anymal_rnea' (rows 18 x columns 55): the per-leg CSR indices of the sinks.
/// reverse: partials by reverse accumulation (15 local outputs against 25 local inputs), sinks and phases ordered row by row instead of column by column.
inline QuadProgram RecordQuadRneaProgram(const rbd::Model& model, const tape::SparseEntries& pattern, bool reverse = false) {
    using namespace rbd;
    using namespace rbd::detail;
    CheckFloatingBaseQuadruped(model);
    constexpr int kRows = 18, kCols = 55;

    QuadProgram P;
    std::vector<int> kOf(static_cast<std::size_t>(kRows * kCols), -1);
    for (std::size_t e = 0; e < pattern.Nnz(); ++e) kOf[static_cast<std::size_t>(pattern.row[e] * kCols + pattern.col[e])] = static_cast<int>(e);
    auto kArgs = [&](int rowBase, int rowLegMul, int colBase, int colLegMul, int rot) {
        std::string s;
        std::array<int, 4> k{};
        for (int L = 0; L < 4; ++L) {
            const int r = rowBase + 3 * rowLegMul * L, c = colBase + 3 * colLegMul * ((L + rot) & 3);
            k[static_cast<std::size_t>(L)] = kOf[static_cast<std::size_t>(r * kCols + c)];
            s += (L ? ", " : "") + std::to_string(k[static_cast<std::size_t>(L)]);
        }
        if (*std::min_element(k.begin(), k.end()) >= 0) {  // per-leg index pattern k_L - k_0: one per-lane base pointer each in the sparse kernel
            const std::array<int, 4> delta{0, k[1] - k[0], k[2] - k[0], k[3] - k[0]};
            if (std::find(P.sparseDeltas.begin(), P.sparseDeltas.end(), delta) == P.sparseDeltas.end()) P.sparseDeltas.push_back(delta);
        }
        return s;
    };
    // ---- inputs: [0,4) quaternion  [4,10) v_b  [10,16) a_b  [16,19) q_L  [19,22) v_L  [22,25) a_L  then the leg constants -----------------
    constexpr int kQuat = 0, kVb = 4, kAb = 10, kQl = 16, kVl = 19, kAl = 22, kConst = 25;
    const std::vector<LegConstantRef> cref = CollectLegConstants(model, P.constants);
    const int nInputs = kConst + static_cast<int>(P.constants.size());
    std::vector<AD> in = tape::Independent(nInputs);
    tape::Graph& g = tape::CurrentGraph();
    for (int i = 0; i < 4; ++i) P.inputNames.push_back("io.qb(" + std::to_string(3 + i) + ")");
    for (int i = 0; i < 6; ++i) P.inputNames.push_back("io.vb(" + std::to_string(i) + ")");
    for (int i = 0; i < 6; ++i) P.inputNames.push_back("io.ab(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.ql(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.vl(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.al(" + std::to_string(i) + ")");
    for (std::size_t i = 0; i < P.constants.size(); ++i) P.inputNames.push_back("io.c(" + std::to_string(i) + ")");
    P.inputUniform.assign(P.inputNames.size(), 0);
    for (int i = 0; i < 16; ++i) P.inputUniform[static_cast<std::size_t>(i)] = 1;  // quaternion, v_b, a_b

    std::size_t cnext = 0;
    auto C = [&]() -> AD {
        const LegConstantRef& r = cref[cnext++];
        return r.literal ? AD{r.value} : in[static_cast<std::size_t>(kConst + r.index)];
    };
    std::array<AD, 3> ql{in[kQl], in[kQl + 1], in[kQl + 2]}, vl{in[kVl], in[kVl + 1], in[kVl + 2]}, al{in[kAl], in[kAl + 1], in[kAl + 2]};

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
    Mat6<AD> Yb;
    {
        const auto Yd = model.joints[1].inertia.Matrix();
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Yb[r][c] = AD{Yd[r][c]};
    }
    auto dotS = [&](std::size_t j, const Vec6<AD>& f) { return f[3] * axis[j][0] + f[4] * axis[j][1] + f[5] * axis[j][2]; };
    auto add6 = [](const Vec6<AD>& a, const Vec6<AD>& b) {
        Vec6<AD> r;
        for (std::size_t k = 0; k < 6; ++k) r[k] = a[k] + b[k];
        return r;
    };
    auto scaleS = [&](std::size_t j, const AD& s) { return Vec6<AD>{AD{0.0}, AD{0.0}, AD{0.0}, s * axis[j][0], s * axis[j][1], s * axis[j][2]}; };

    const Rot<AD> Rb = QuaternionToRotation(in[kQuat], in[kQuat + 1], in[kQuat + 2], in[kQuat + 3]);
    Vec6<AD> accB;  // a_b + [R_b^T (-g); 0]: gravity as a fictitious base acceleration
    for (std::size_t k = 0; k < 3; ++k) {
        AD acc{0.0};
        for (std::size_t r = 0; r < 3; ++r) acc = acc + Rb[r][k] * (-model.gravity[r]);
        accB[k] = in[static_cast<std::size_t>(kAb) + k] + acc;
        accB[3 + k] = in[static_cast<std::size_t>(kAb) + 3 + k];
    }
    const Vec6<AD> velB{in[kVb], in[kVb + 1], in[kVb + 2], in[kVb + 3], in[kVb + 4], in[kVb + 5]};

    // ---- RNEA restricted to one leg ----------------------------------------------------------------------------------------------------------
    std::array<Vec6<AD>, 3> vel, acc, f;
    for (std::size_t j = 0; j < 3; ++j) {
        const Vec6<AD> vj = scaleS(j, vl[j]);
        vel[j] = add6(ActInvMotion(X[j], j == 0 ? velB : vel[j - 1]), vj);
        acc[j] = add6(add6(ActInvMotion(X[j], j == 0 ? accB : acc[j - 1]), scaleS(j, al[j])), CrossMotion(vel[j], vj));
        f[j] = add6(MatVec6(Y[j], acc[j]), CrossForce(vel[j], MatVec6(Y[j], vel[j])));
    }
    std::array<AD, 3> tau;
    Vec6<AD> fLeg;
    for (std::size_t j = 3; j-- > 0;) {
        tau[j] = dotS(j, f[j]);
        const Vec6<AD> up = ActForce(X[j], f[j]);
        if (j > 0) f[j - 1] = add6(f[j - 1], up);
        else fLeg = up;
    }
    const Vec6<AD> fOwn = add6(MatVec6(Yb, accB), CrossForce(velB, MatVec6(Yb, velB)));

    // ---- value sinks: rows 0..5 base wrench, 6 + 3 L + k leg torques ---------------------------------------------------------------------------
    for (int r = 0; r < 6; ++r) P.slots.push_back({(fOwn[static_cast<std::size_t>(r)] + tape::QuadSum(fLeg[static_cast<std::size_t>(r)])).Node(), "io.f_base(" + std::to_string(r) + ", %s);"});
    for (int k = 0; k < 3; ++k) P.slots.push_back({tau[static_cast<std::size_t>(k)].Node(), "io.f_leg(" + std::to_string(6 + k) + ", %s);"});

    // ---- partials of the lane-local function ------------------------------------------------------------------------------------------------
    std::vector<tape::Id> inputIds;
    for (const AD& i : in) inputIds.push_back(i.Node());
    tape::Differentiator diff{g, inputIds};
    std::vector<tape::Id> tIds;  // tau_L(3) f_L(6) f_own(6)
    for (const AD& v : tau) tIds.push_back(v.Node());
    for (const AD& v : fLeg) tIds.push_back(v.Node());
    for (const AD& v : fOwn) tIds.push_back(v.Node());
    std::vector<int> dCols;
    for (int k = 0; k < kConst; ++k) dCols.push_back(k);
    const tape::SparseEntries D = diff.Jacobian(tIds, dCols, reverse ? 2 : 1);  // forward: a column's partials are born together; reverse: a row's
    AD Dm[15][kConst];
    for (std::size_t e = 0; e < D.Nnz(); ++e) Dm[D.row[e]][D.col[e]] = AD::FromId(D.value[e]);

    auto checkZero = [&](const AD& v, int rowBase, int rowLegMul, int colBase, int colLegMul, int rot) {  // an entry outside the pattern must be a literal zero
        for (int L = 0; L < 4; ++L) {
            const int r = rowBase + 3 * rowLegMul * L, c = colBase + 3 * colLegMul * ((L + rot) & 3);
            if (kOf[static_cast<std::size_t>(r * kCols + c)] < 0 && !(v.IsLiteral() && v.Literal() == 0.0))
                throw std::runtime_error("quad rnea program: non-zero entry outside the sparsity pattern at (" + std::to_string(r) + "," + std::to_string(c) + ")");
        }
    };
    auto legSink = [&](const AD& v, int k, int colBase, int colLegMul, int rot) {
        checkZero(v, 6 + k, 1, colBase, colLegMul, rot);
        P.slots.push_back({v.Node(), "io.j_leg(" + std::to_string(6 + k) + ", " + std::to_string(colBase) + ", " + std::to_string(colLegMul) + ", " + std::to_string(rot) + ", " +
                                         kArgs(6 + k, 1, colBase, colLegMul, rot) + ", %s);"});
    };
    // node-level columns: x = [p 0..2 | quat 3..6 | q_leg 7 + 3 L + k | v_b 19..24 | v_leg 25 + 3 L + k],  u = [a_b 37..42 | a_leg 43 + 3 L + k]
    const int ownLocal[3] = {kQl, kVl, kAl}, ownNode[3] = {7, 25, 43};
    struct SharedCol {
        int local, node;
    };
    std::vector<SharedCol> shared;
    for (int k = 0; k < 3; ++k) shared.push_back({-1, k});  // position: zeros
    for (int k = 0; k < 4; ++k) shared.push_back({kQuat + k, 3 + k});
    for (int k = 0; k < 6; ++k) shared.push_back({kVb + k, 19 + k});
    for (int k = 0; k < 6; ++k) shared.push_back({kAb + k, 37 + k});
    auto baseOwnSink = [&](int r, int lc, int colBase) {
        const AD v = Dm[3 + r][lc];  // (f_own does not depend on leg variables)
        checkZero(v, r, 0, colBase, 1, 0);
        P.slots.push_back({v.Node(), "io.j_base_own(" + std::to_string(r) + ", " + std::to_string(colBase) + ", " + kArgs(r, 0, colBase, 1, 0) + ", %s);"});
    };
    auto baseSharedSink = [&](int r, int lc, int col) {
        const AD v = lc >= 0 ? Dm[9 + r][lc] + tape::QuadSum(Dm[3 + r][lc]) : AD{0.0};
        checkZero(v, r, 0, col, 0, 0);
        P.slots.push_back({v.Node(), "io.j_base_shared(" + std::to_string(r) + ", " + std::to_string(col) + ", " + std::to_string(kOf[static_cast<std::size_t>(r * kCols + col)]) + ", %s);"});
    };
    if (!reverse) {
        // ---- columns owned by this lane's leg, then the shared ones: one phase per column ------------------------------------------------------
        for (int kind = 0; kind < 3; ++kind)
            for (int k = 0; k < 3; ++k) {
                P.phaseStarts.push_back(P.slots.size());
                const int lc = ownLocal[kind] + k, colBase = ownNode[kind] + k;
                for (int r = 0; r < 6; ++r) baseOwnSink(r, lc, colBase);
                for (int kk = 0; kk < 3; ++kk) legSink(Dm[kk][lc], kk, colBase, 1, 0);
                for (int rot = 1; rot < 4; ++rot)  // the other legs' versions of this column: exact zeros in this leg's rows
                    for (int kk = 0; kk < 3; ++kk) legSink(AD{0.0}, kk, colBase, 1, rot);
            }
        for (const SharedCol& sc : shared) {
            P.phaseStarts.push_back(P.slots.size());
            for (int r = 0; r < 6; ++r) baseSharedSink(r, sc.local, sc.node);
            for (int kk = 0; kk < 3; ++kk) legSink(sc.local >= 0 ? Dm[kk][sc.local] : AD{0.0}, kk, sc.node, 0, 0);
        }
    } else {
        // ---- one phase per row of the local function: the leg's torque rows (top of the leg first), then the base rows --------------------------
        for (int kk = 2; kk >= 0; --kk) {
            P.phaseStarts.push_back(P.slots.size());
            for (const SharedCol& sc : shared) legSink(sc.local >= 0 ? Dm[kk][sc.local] : AD{0.0}, kk, sc.node, 0, 0);
            for (int kind = 0; kind < 3; ++kind)
                for (int k = 0; k < 3; ++k) {
                    legSink(Dm[kk][ownLocal[kind] + k], kk, ownNode[kind] + k, 1, 0);
                    for (int rot = 1; rot < 4; ++rot) legSink(AD{0.0}, kk, ownNode[kind] + k, 1, rot);
                }
        }
        for (int r = 0; r < 6; ++r) {
            P.phaseStarts.push_back(P.slots.size());
            for (const SharedCol& sc : shared) baseSharedSink(r, sc.local, sc.node);
            for (int kind = 0; kind < 3; ++kind)
                for (int k = 0; k < 3; ++k) baseOwnSink(r, ownLocal[kind] + k, ownNode[kind] + k);
        }
    }

    std::vector<AD> roots;
    for (const auto& sl : P.slots) roots.push_back(AD::FromId(sl.value));
    P.tape = tape::MakeTape(roots);
    for (std::size_t i = 0; i < P.slots.size(); ++i) P.slots[i].value = P.tape.outputs[i];
    return P;
}

}  // namespace ungar_amd::codegen
