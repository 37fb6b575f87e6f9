// ungar_amd :: SPMD "lane per leg" program for a floating-base quadruped shooting node.
//
// The structured derivative program of DESIGN.md §4.3 keeps ~600 doubles alive per node -- more than a
// lane can hold on chip (128 ArchVGPR + 128 AGPR doubles, plus a share of the 160 KiB LDS).  The robot
// is a floating base with four structurally identical 3-joint legs that only interact through the
// base, so the node is split over the FOUR LANES OF A QUAD, one lane per leg (state per lane ~170
// doubles: registers only, four wavefronts per CU).  All four lanes run the SAME straight-line
// program; leg-specific numbers come from a constants table, and the lanes meet at a few points:
//   quad_sum : composite inertia of the base, Schur complement of the block-arrow mass matrix, base
//              bias force, shared right-hand sides;
//   quad_rot : the base part y_b of M^-1 r for a column owned by another leg (so that every lane
//              fills ITS OWN leg's rows of every Jacobian column).
// The program is recorded on the tape like everything else (RNEA tangents come from the tape's own
// differentiator, applied to the lane-local leg function with the solved accelerations held fixed
// through auxiliary inputs), and is emitted generic over the value type so that the same text runs
// on the GPU (T = double, DPP quad permutes) and in a 4-lane CPU simulator (T = Quad, tests).
//
// Math per lane (leg L), notation of csrc/rbd/*.hpp:
//   M = [[Ybb, M_bL...], [M_bL^T, M_LL]]   block-arrow;  U D U^T from the bottom:
//   M_LL = U_LL D_L U_LL^T,  U_bL = M_bL U_LL^-T D_L^-1,  S = Ybb - sum_L U_bL D_L U_bL^T  (quad_sum)
//   solve(r_b, r_L):  z_L = U_LL^-1 r_L;  y_b = S^-1 (r_b - [sum_L] U_bL z_L);  y_L = U_LL^-T (D_L^-1 z_L - U_bL^T y_b)
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../models/nodes.hpp"
#include "../rbd/rnea_crba.hpp"
#include "../tape/emit.hpp"

namespace ungar_amd::codegen {

using tape::AD;

struct QuadProgram {
    tape::Tape tape;                         // graph + inputs (outputs unused)
    std::vector<tape::OutputSlot> slots;     // in emission order
    std::vector<std::size_t> phaseStarts;    // slot indices where a new phase (Jacobian column) begins
    std::vector<std::string> inputNames;     // spelling of every tape input in the generated code
    std::vector<char> inputUniform;          // 1: same value in the four lanes of a quad (base state, dt)
    std::vector<char> inputLate;             // 1: read where it is used, in every phase that uses it (channel items of the split program)
    // distinct per-leg CSR index patterns (k_L - k_0, L = 0..3) of the per-lane sinks: the sparse kernel keeps
    // one per-lane base pointer per pattern, so that a sparse store costs what a dense one does
    std::vector<std::array<int, 4>> sparseDeltas;
    std::vector<std::array<double, 4>> constants;  // constants[k][leg]
    // tile program (tileStores): entry (row * 49 + col, -1 = padding) held by lane `leg` in store image `sink`: tileEntries[4 * sink + leg].
    // Two consecutive images leave in one 16-byte store instruction (io.t_put2), 1 KiB contiguous per wavefront.
    std::vector<int> tileEntries;
};

/// The node program as ONE wavefront (Fused), or split over the two wavefronts of a workgroup:
///   Producer: leg kinematics, bias forces, RNEA tangents -> right-hand sides of the Jacobian columns, handed over through an LDS ring;
///   Consumer: composite inertias, block-arrow factorisation, solves, integrator chain, result stores.
/// Each half keeps its state in 256 registers + its share of LDS, so that TWO wavefronts run per SIMD.
enum class QuadRole { Fused, Producer, Consumer };
inline constexpr int kQuadMessageItems = 9;   // values per lane and message: r_L(3), r_b(6)
inline constexpr int kQuadMessages = 13;      // bias forces + 12 columns whose right-hand side needs the RNEA tangents

namespace detail {

inline bool IsZeroLit(const AD& a) {
    return a.IsLiteral() && a.Literal() == 0.0;
}

}  // namespace detail

/// The lane-per-leg programs assume a free-flyer with four structurally identical 3-joint legs off the base (same joint axes, placements that
/// are pure translations): checked here, the leg-specific numbers go through a constants table.
inline void CheckFloatingBaseQuadruped(const rbd::Model& model) {
    using rbd::Joint;
    if (model.NumJoints() != 14 || model.nq != 19 || model.nv != 18) throw std::runtime_error("quad program: expected a free-flyer with 12 revolute joints");
    for (int L = 0; L < 4; ++L)
        for (int j = 0; j < 3; ++j) {
            const Joint& J = model.joints[static_cast<std::size_t>(2 + 3 * L + j)];
            const Joint& J0 = model.joints[static_cast<std::size_t>(2 + j)];
            if (J.parent != (j == 0 ? 1 : 2 + 3 * L + j - 1)) throw std::runtime_error("quad program: legs must be 3-joint chains off the base");
            if (J.axis != J0.axis) throw std::runtime_error("quad program: legs must share joint axes");
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    if (std::fabs(J.placement.R[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)] - (r == c ? 1.0 : 0.0)) > 1e-14)
                        throw std::runtime_error("quad program: joint placements must be pure translations");
        }
}

/// Leg constants in the order the programs consume them -- per joint t(3), mass, h(3), I(6 upper) -- as a literal when identical in the four
/// legs, otherwise as an index into `table` ([k][leg]).
struct LegConstantRef {
    bool literal;
    double value;
    int index;
};
inline std::vector<LegConstantRef> CollectLegConstants(const rbd::Model& model, std::vector<std::array<double, 4>>& table) {
    using rbd::Joint;
    std::vector<LegConstantRef> cref;
    auto addConst = [&](const std::array<double, 4>& v) {
        if (v[0] == v[1] && v[1] == v[2] && v[2] == v[3]) {
            cref.push_back({true, v[0], -1});
        } else {
            cref.push_back({false, 0.0, static_cast<int>(table.size())});
            table.push_back(v);
        }
    };
    for (int j = 0; j < 3; ++j) {
        auto joint = [&](int L) -> const Joint& { return model.joints[static_cast<std::size_t>(2 + 3 * L + j)]; };
        for (std::size_t k = 0; k < 3; ++k) addConst({joint(0).placement.p[k], joint(1).placement.p[k], joint(2).placement.p[k], joint(3).placement.p[k]});
        addConst({joint(0).inertia.mass, joint(1).inertia.mass, joint(2).inertia.mass, joint(3).inertia.mass});
        for (std::size_t k = 0; k < 3; ++k) addConst({joint(0).inertia.h[k], joint(1).inertia.h[k], joint(2).inertia.h[k], joint(3).inertia.h[k]});
        for (std::size_t r = 0; r < 3; ++r)
            for (std::size_t c = r; c < 3; ++c) addConst({joint(0).inertia.I[r][c], joint(1).inertia.I[r][c], joint(2).inertia.I[r][c], joint(3).inertia.I[r][c]});
    }
    return cref;
}

/// Records the leg-lane program for a free-flyer + 4 x (3 revolute) robot.
inline QuadProgram RecordQuadLegProgram(const rbd::Model& model, const tape::SparseEntries& pattern, int columnsPerPhase = 1,
                                        bool mergeSharedStores = false, QuadRole role = QuadRole::Fused, bool pairStores = false, bool tileStores = false) {
    using namespace rbd;
    using namespace rbd::detail;
    CheckFloatingBaseQuadruped(model);

    QuadProgram P;
    // CSR index of (row, col) in the node's sparse pattern, -1 for structural zeros (sparse output mode)
    std::vector<int> kOf(static_cast<std::size_t>(37 * 49), -1);
    for (std::size_t e = 0; e < pattern.Nnz(); ++e) kOf[static_cast<std::size_t>(pattern.row[e] * 49 + pattern.col[e])] = static_cast<int>(e);
    auto kArgs = [&](int rowBase, int rowLegMul, int colBase, int colLegMul, int rot) {
        std::string s;
        std::array<int, 4> k{};
        for (int L = 0; L < 4; ++L) {
            const int r = rowBase + 3 * rowLegMul * L, c = colBase + 3 * colLegMul * ((L + rot) & 3);
            k[static_cast<std::size_t>(L)] = kOf[static_cast<std::size_t>(r * 49 + c)];
            s += (L ? ", " : "") + std::to_string(k[static_cast<std::size_t>(L)]);
        }
        if (*std::min_element(k.begin(), k.end()) >= 0) {
            const std::array<int, 4> delta{0, k[1] - k[0], k[2] - k[0], k[3] - k[0]};
            if (std::find(P.sparseDeltas.begin(), P.sparseDeltas.end(), delta) == P.sparseDeltas.end()) P.sparseDeltas.push_back(delta);
        }
        return s;
    };
    // ---- inputs --------------------------------------------------------------------------------------------
    // [0,7) q_b  [7,13) v_b  [13,16) q_L  [16,19) v_L  [19,22) u_L  [22] dt  [23, 23+K) constants  then 9 aux
    constexpr int kQb = 0, kVb = 7, kQl = 13, kVl = 16, kUl = 19, kDt = 22, kConst = 23;
    const std::vector<LegConstantRef> cref = CollectLegConstants(model, P.constants);
    const int K = static_cast<int>(P.constants.size());
    const int kAux = kConst + K;  // aux: accb(6), aL(3)
    const int kRecv = kAux + 9;   // split program: items of the producer's messages as the consumer sees them
    const int nInputs = kRecv + kQuadMessages * kQuadMessageItems;
    const bool fused = role == QuadRole::Fused, producer = role == QuadRole::Producer, consumer = role == QuadRole::Consumer;
    if (pairStores && mergeSharedStores) throw std::runtime_error("quad program: paired stores and merged shared stores exclude each other");
    if (tileStores && (pairStores || mergeSharedStores || role != QuadRole::Fused)) throw std::runtime_error("quad program: tile stores are a mode of the fused program of their own");
    if (!fused && (columnsPerPhase != 1 || mergeSharedStores)) throw std::runtime_error("quad program: the split program takes one column per phase");
    std::vector<AD> in = tape::Independent(nInputs);
    tape::Graph& g = tape::CurrentGraph();
    for (int i = 0; i < 7; ++i) P.inputNames.push_back("io.qb(" + std::to_string(i) + ")");
    for (int i = 0; i < 6; ++i) P.inputNames.push_back("io.vb(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.ql(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.vl(" + std::to_string(i) + ")");
    for (int i = 0; i < 3; ++i) P.inputNames.push_back("io.ul(" + std::to_string(i) + ")");
    P.inputNames.push_back("io.dt()");
    for (int i = 0; i < K; ++i) P.inputNames.push_back("io.c(" + std::to_string(i) + ")");
    // the producer reads the solved accelerations the consumer hands back: a_b + gamma (6, the same in the four lanes), a_L (3)
    for (int i = 0; i < 9; ++i) P.inputNames.push_back(producer ? "io.acc(" + std::to_string(i) + ")" : "aux_unused");
    for (int m = 0; m < kQuadMessages; ++m)
        for (int i = 0; i < kQuadMessageItems; ++i) P.inputNames.push_back("io.recv(" + std::to_string(m) + ", " + std::to_string(i) + ")");
    P.inputUniform.assign(P.inputNames.size(), 0);
    P.inputLate.assign(P.inputNames.size(), 0);
    for (int i = kAux; i < nInputs; ++i) P.inputLate[static_cast<std::size_t>(i)] = 1;
    if (producer)
        for (int i = 0; i < 6; ++i) P.inputUniform[static_cast<std::size_t>(kAux + i)] = 1;
    for (int i = 0; i < 13; ++i) P.inputUniform[static_cast<std::size_t>(i)] = 1;  // qb, vb
    P.inputUniform[22] = 1;                                                        // dt

    std::size_t cnext = 0;
    auto C = [&]() -> AD {
        const LegConstantRef& r = cref[cnext++];
        return r.literal ? AD{r.value} : in[static_cast<std::size_t>(kConst + r.index)];
    };
    const AD dt = in[kDt];
    std::vector<AD> vb(in.begin() + kVb, in.begin() + kVb + 6);
    std::array<AD, 3> ql{in[kQl], in[kQl + 1], in[kQl + 2]}, vl{in[kVl], in[kVl + 1], in[kVl + 2]}, ul{in[kUl], in[kUl + 1], in[kUl + 2]};

    // ---- hand-over between the two halves of the split program ----------------------------------------------------------
    // channel(m, i, v): item i of message m.  Fused: the value itself.  Producer: v becomes a send sink of message m.
    // Consumer: a fresh input standing for the received item (the producer's sub-graph is then unreachable from the
    // consumer's sinks and is not emitted).  Literal items are not transmitted (both halves record the same graph).
    std::vector<std::vector<tape::OutputSlot>> sends(static_cast<std::size_t>(kQuadMessages));
    const tape::Id noValue = g.Constant(0.0);  // value of a sink that only carries a statement (wait / post / done)
    auto channel = [&](int m, int i, const AD& v, bool uniform) -> AD {
        if (fused || v.IsLiteral() || g.At(v.Node()).op == tape::Op::Const) return v;
        if (m < 0 || m >= kQuadMessages || i < 0 || i >= kQuadMessageItems) throw std::logic_error("quad program: channel item out of range");
        if (producer) {
            sends[static_cast<std::size_t>(m)].push_back({v.Node(), "io.send(" + std::to_string(m) + ", " + std::to_string(i) + ", %s);"});
            return v;
        }
        const std::size_t slot = static_cast<std::size_t>(kRecv + kQuadMessageItems * m + i);
        P.inputUniform[slot] = uniform ? 1 : 0;
        return in[slot];
    };

    // ---- leg kinematics and inertias ---------------------------------------------------------------------------
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
    auto Sjoint = [&](std::size_t j) { return Vec6<AD>{AD{0.0}, AD{0.0}, AD{0.0}, AD{axis[j][0]}, AD{axis[j][1]}, AD{axis[j][2]}}; };
    auto dotS = [&](std::size_t j, const Vec6<AD>& f) { return f[3] * axis[j][0] + f[4] * axis[j][1] + f[5] * axis[j][2]; };
    auto add6 = [](const Vec6<AD>& a, const Vec6<AD>& b) {
        Vec6<AD> r;
        for (std::size_t k = 0; k < 6; ++k) r[k] = a[k] + b[k];
        return r;
    };
    auto scaleS = [&](std::size_t j, const AD& s) { return Vec6<AD>{AD{0.0}, AD{0.0}, AD{0.0}, s * axis[j][0], s * axis[j][1], s * axis[j][2]}; };

    const Rot<AD> Rb = QuaternionToRotation(in[kQb + 3], in[kQb + 4], in[kQb + 5], in[kQb + 6]);
    std::array<AD, 3> gamma;
    for (std::size_t k = 0; k < 3; ++k) {
        AD acc{0.0};
        for (std::size_t r = 0; r < 3; ++r) acc = acc + Rb[r][k] * (-model.gravity[r]);
        gamma[k] = acc;
    }
    const Vec6<AD> velB{vb[0], vb[1], vb[2], vb[3], vb[4], vb[5]};

    // ---- leg function: RNEA restricted to one leg, for given base acceleration and joint accelerations ---------
    struct LegOut {
        std::array<AD, 3> tau;
        Vec6<AD> fb;
    };
    auto legRnea = [&](const Vec6<AD>& accB, const std::array<AD, 3>& aL) {
        std::array<Vec6<AD>, 3> vel, acc, f;
        for (std::size_t j = 0; j < 3; ++j) {
            const Vec6<AD> vj = scaleS(j, vl[j]);
            vel[j] = add6(ActInvMotion(X[j], j == 0 ? velB : vel[j - 1]), vj);
            acc[j] = add6(add6(ActInvMotion(X[j], j == 0 ? accB : acc[j - 1]), scaleS(j, aL[j])), CrossMotion(vel[j], vj));
            f[j] = add6(MatVec6(Y[j], acc[j]), CrossForce(vel[j], MatVec6(Y[j], vel[j])));
        }
        LegOut o;
        for (std::size_t j = 3; j-- > 0;) {
            o.tau[j] = dotS(j, f[j]);
            const Vec6<AD> up = ActForce(X[j], f[j]);
            if (j > 0) f[j - 1] = add6(f[j - 1], up);
            else o.fb = up;
        }
        return o;
    };
    auto baseOwnForce = [&](const Vec6<AD>& accB) { return add6(MatVec6(Yb, accB), CrossForce(velB, MatVec6(Yb, velB))); };

    // ---- CRBA: composite inertias, mass-matrix blocks -------------------------------------------------------------
    std::array<Mat6<AD>, 3> Yc = Y;
    for (std::size_t j = 2; j >= 1; --j) {
        const Mat6<AD> T = TransportInertia(X[j], Yc[j]);
        for (std::size_t r = 0; r < 6; ++r)
            for (std::size_t c = 0; c < 6; ++c) Yc[j - 1][r][c] = Yc[j - 1][r][c] + T[r][c];
    }
    const Mat6<AD> YcbLeg = TransportInertia(X[0], Yc[0]);
    Mat6<AD> Ybb;
    for (std::size_t r = 0; r < 6; ++r)
        for (std::size_t c = r; c < 6; ++c) Ybb[r][c] = Ybb[c][r] = Yb[r][c] + tape::QuadSum(YcbLeg[r][c]);
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
    // ---- block-arrow U D U^T ------------------------------------------------------------------------------------------
    AD ULL[3][3], UbL[6][3], dL[3], dinvL[3];
    for (std::size_t kk = 3; kk-- > 0;) {
        AD dk = MLL[kk][kk];
        for (std::size_t j = kk + 1; j < 3; ++j) dk = dk - ULL[kk][j] * ULL[kk][j] * dL[j];
        dL[kk] = dk;
        dinvL[kk] = AD{1.0} / dk;
        for (std::size_t i = 0; i < kk; ++i) {
            AD s = MLL[i][kk];
            for (std::size_t j = kk + 1; j < 3; ++j) s = s - ULL[i][j] * ULL[kk][j] * dL[j];
            ULL[i][kk] = s * dinvL[kk];
        }
        for (std::size_t r = 0; r < 6; ++r) {
            AD s = MbL[r][kk];
            for (std::size_t j = kk + 1; j < 3; ++j) s = s - UbL[r][j] * ULL[kk][j] * dL[j];
            UbL[r][kk] = s * dinvL[kk];
        }
    }
    std::vector<std::vector<AD>> Smat(6, std::vector<AD>(6));
    for (std::size_t r = 0; r < 6; ++r)
        for (std::size_t c = r; c < 6; ++c) {
            AD acc{0.0};
            for (std::size_t k = 0; k < 3; ++k) acc = acc + UbL[r][k] * UbL[c][k] * dL[k];
            Smat[r][c] = Smat[c][r] = Ybb[r][c] - tape::QuadSum(acc);
        }
    const UdutFactor<AD> Fb = FactorUdut(Smat);

    struct Sol {
        std::vector<AD> yb;
        std::array<AD, 3> yl;
    };
    /// shared = the right-hand side is non-zero in every leg (all lanes solve the SAME system).
    auto solve = [&](const std::vector<AD>& rb, const std::array<AD, 3>& rl, bool shared) {
        std::array<AD, 3> z = rl;
        for (std::size_t k = 3; k-- > 0;)
            for (std::size_t j = k + 1; j < 3; ++j) z[k] = z[k] - ULL[k][j] * z[j];
        std::vector<AD> zb(6);
        for (std::size_t r = 0; r < 6; ++r) {
            AD c{0.0};
            for (std::size_t k = 0; k < 3; ++k) c = c + UbL[r][k] * z[k];
            zb[r] = rb[r] - (shared ? tape::QuadSum(c) : c);
        }
        Sol s;
        s.yb = SolveUdut(Fb, zb);
        for (std::size_t k = 0; k < 3; ++k) {
            AD y = z[k] * dinvL[k];
            for (std::size_t r = 0; r < 6; ++r) y = y - UbL[r][k] * s.yb[r];
            for (std::size_t j = 0; j < k; ++j) y = y - ULL[j][k] * s.yl[j];
            s.yl[k] = y;
        }
        return s;
    };
    /// rows of THIS leg for a column whose right-hand side lives entirely in another leg / the base.
    auto foreignRows = [&](const std::vector<AD>& yb) {
        std::array<AD, 3> y;
        for (std::size_t k = 0; k < 3; ++k) {
            AD v{0.0};
            for (std::size_t r = 0; r < 6; ++r) v = v - UbL[r][k] * yb[r];
            for (std::size_t j = 0; j < k; ++j) v = v - ULL[j][k] * y[j];
            y[k] = v;
        }
        return y;
    };

    // ---- primal acceleration -------------------------------------------------------------------------------------------
    const Vec6<AD> accB0{gamma[0], gamma[1], gamma[2], AD{0.0}, AD{0.0}, AD{0.0}};
    const LegOut h = legRnea(accB0, {AD{0.0}, AD{0.0}, AD{0.0}});
    const Vec6<AD> hbOwn = baseOwnForce(accB0);
    std::vector<AD> rb0(6);
    std::array<AD, 3> rl0;
    if (fused) {
        for (std::size_t r = 0; r < 6; ++r) rb0[r] = -(hbOwn[r] + tape::QuadSum(h.fb[r]));
        rl0 = {ul[0] - h.tau[0], ul[1] - h.tau[1], ul[2] - h.tau[2]};
    } else {  // message 0: the leg's bias torques and its summed bias force on the base
        for (std::size_t r = 0; r < 6; ++r) rb0[r] = -(hbOwn[r] + channel(0, 3 + static_cast<int>(r), tape::QuadSum(h.fb[r]), true));
        for (std::size_t k = 0; k < 3; ++k) rl0[k] = ul[k] + channel(0, static_cast<int>(k), -h.tau[k], false);
    }
    const Sol acc = solve(rb0, rl0, true);

    // ---- stage functions on auxiliary inputs: integrator and RNEA at fixed accelerations -----------------------------------
    std::vector<AD> accbAux(in.begin() + kAux, in.begin() + kAux + 6);  // stands for a_b (integrator) / a_b + gamma (RNEA)
    std::array<AD, 3> alAux{in[static_cast<std::size_t>(kAux + 6)], in[static_cast<std::size_t>(kAux + 7)], in[static_cast<std::size_t>(kAux + 8)]};
    std::vector<tape::Id> inputIds;
    for (const AD& i : in) inputIds.push_back(i.Node());
    tape::Differentiator diff{g, inputIds};

    // integrator rows: base (13) then leg (6)
    using models::ApproximateExponentialMap;
    using models::QuatMul;
    using models::Rotate;
    using models::Scale;
    std::vector<AD> gOut;  // p+(3) quat+(4) v_b+(6) q_L+(3) v_L+(3)
    {
        std::array<AD, 6> vbN;
        for (std::size_t k = 0; k < 6; ++k) vbN[k] = vb[k] + dt * accbAux[k];
        const models::Quat<AD> quat{in[kQb + 3], in[kQb + 4], in[kQb + 5], in[kQb + 6]};
        const models::Vec3<AD> lin = Rotate(quat, models::Vec3<AD>{vbN[0], vbN[1], vbN[2]});
        for (std::size_t k = 0; k < 3; ++k) gOut.push_back(in[static_cast<std::size_t>(kQb) + k] + dt * lin[k]);
        const models::Quat<AD> qN = QuatMul(quat, ApproximateExponentialMap(Scale(dt, models::Vec3<AD>{vbN[3], vbN[4], vbN[5]})));
        for (std::size_t k = 0; k < 4; ++k) gOut.push_back(qN[k]);
        for (std::size_t k = 0; k < 6; ++k) gOut.push_back(vbN[k]);
        std::array<AD, 3> vlN;
        for (std::size_t k = 0; k < 3; ++k) vlN[k] = vl[k] + dt * alAux[k];
        for (std::size_t k = 0; k < 3; ++k) gOut.push_back(ql[k] + dt * vlN[k]);
        for (std::size_t k = 0; k < 3; ++k) gOut.push_back(vlN[k]);
    }
    std::vector<tape::Id> gIds;
    for (const AD& v : gOut) gIds.push_back(v.Node());
    // integrator partials w.r.t. [q_b(7) v_b(6) q_L(3) v_L(3) | a_b(6) a_L(3)]
    std::vector<int> gCols;
    for (int k = 0; k < 19; ++k) gCols.push_back(k);
    for (int k = 0; k < 9; ++k) gCols.push_back(kAux + k);
    const tape::SparseEntries G = diff.Jacobian(gIds, gCols, 1);  // forward: each column's partials are born in its own phase
    // RNEA stage
    const Vec6<AD> accBAux{accbAux[0], accbAux[1], accbAux[2], accbAux[3], accbAux[4], accbAux[5]};
    const LegOut tl = legRnea(accBAux, alAux);
    const Vec6<AD> fbOwn = baseOwnForce(accBAux);
    std::vector<tape::Id> tIds;  // tau_L(3) fb_leg(6) fb_own(6)
    for (const AD& v : tl.tau) tIds.push_back(v.Node());
    for (const AD& v : tl.fb) tIds.push_back(v.Node());
    for (const AD& v : fbOwn) tIds.push_back(v.Node());
    std::vector<int> dCols;  // q_L(3) v_L(3) v_b(6)
    for (int k = 0; k < 3; ++k) dCols.push_back(kQl + k);
    for (int k = 0; k < 3; ++k) dCols.push_back(kVl + k);
    for (int k = 0; k < 6; ++k) dCols.push_back(kVb + k);
    const tape::SparseEntries D = diff.Jacobian(tIds, dCols, 1);
    const tape::SparseEntries dGamma = diff.Jacobian({gamma[0].Node(), gamma[1].Node(), gamma[2].Node()}, std::vector<int>{3, 4, 5, 6});

    // substitutions: integrator sees a_b, RNEA sees a_b + gamma
    std::vector<std::pair<int, tape::Id>> subG, subD;
    for (std::size_t k = 0; k < 6; ++k) {
        subG.emplace_back(kAux + static_cast<int>(k), acc.yb[k].Node());
        subD.emplace_back(kAux + static_cast<int>(k), (acc.yb[k] + (k < 3 ? gamma[k] : AD{0.0})).Node());
    }
    for (std::size_t k = 0; k < 3; ++k) {
        subG.emplace_back(kAux + 6 + static_cast<int>(k), acc.yl[k].Node());
        subD.emplace_back(kAux + 6 + static_cast<int>(k), acc.yl[k].Node());
    }
    // (the producer's auxiliary inputs ARE the accelerations it receives: nothing to substitute there)
    const std::vector<tape::Id> Gv = diff.Substitute(G.value, subG), fv = diff.Substitute(gIds, subG), Dv = producer ? D.value : diff.Substitute(D.value, subD);
    // dense views of the small sparse blocks
    AD Gm[19][28], Dm[15][12], dGam[3][4];
    for (std::size_t e = 0; e < G.Nnz(); ++e) Gm[G.row[e]][G.col[e]] = AD::FromId(Gv[e]);
    for (std::size_t e = 0; e < D.Nnz(); ++e) Dm[D.row[e]][D.col[e]] = AD::FromId(Dv[e]);
    for (std::size_t e = 0; e < dGamma.Nnz(); ++e) dGam[dGamma.row[e]][dGamma.col[e]] = AD::FromId(dGamma.value[e]);

    // ---- output helpers ---------------------------------------------------------------------------------------------------
    // node-level indices: x = [p 0..2 | quat 3..6 | q_leg 7+3L+k | v_b 19..24 | v_leg 25+3L+k],  u column 37+3L+k
    auto baseRowIndex = [](int i) { return i < 7 ? i : 19 + (i - 7); };  // i in [0,13): p,quat,v_b
    auto sinkF = [&](const AD& v, const std::string& call) { P.slots.push_back({v.Node(), call}); };
    // phase 0 = kinematics + CRBA + block-arrow factorisation (its results are pinned by no-op sinks so that the
    // emitter can place them in LDS / rematerialise around them); phase 1 = bias forces, primal solve, f
    auto keep = [&](const AD& v) {
        if (!v.IsLiteral()) P.slots.push_back({v.Node(), "io.keep(%s);"});
    };
    for (std::size_t r = 0; r < 6; ++r)
        for (std::size_t k = 0; k < 3; ++k) keep(UbL[r][k]);
    for (std::size_t i = 0; i < 3; ++i) {
        keep(dinvL[i]);
        for (std::size_t j = i + 1; j < 3; ++j) keep(ULL[i][j]);
    }
    for (std::size_t i = 0; i < 6; ++i) {
        keep(Fb.dinv[i]);
        for (std::size_t j = i + 1; j < 6; ++j) keep(Fb.U[i][j]);
    }
    if (consumer) {  // the solved accelerations go back to the producer as early as possible: a phase of their own
        P.phaseStarts.push_back(P.slots.size());
        P.slots.push_back({noValue, "@begin:io.wait(0);"});
        for (std::size_t k = 0; k < 6; ++k) P.slots.push_back({(acc.yb[k] + (k < 3 ? gamma[k] : AD{0.0})).Node(), "io.send_acc(" + std::to_string(k) + ", %s);"});
        for (std::size_t k = 0; k < 3; ++k) P.slots.push_back({acc.yl[k].Node(), "io.send_acc(" + std::to_string(6 + k) + ", %s);"});
        P.slots.push_back({noValue, "io.post_acc();"});
        P.slots.push_back({noValue, "io.done(0);"});
    }
    P.phaseStarts.push_back(P.slots.size());
    for (int i = 0; i < 13; ++i) sinkF(AD::FromId(fv[static_cast<std::size_t>(i)]), "io.f_base(" + std::to_string(baseRowIndex(i)) + ", %s);");
    for (int k = 0; k < 3; ++k) sinkF(AD::FromId(fv[static_cast<std::size_t>(13 + k)]), "io.f_leg(" + std::to_string(7 + k) + ", %s);");
    for (int k = 0; k < 3; ++k) sinkF(AD::FromId(fv[static_cast<std::size_t>(16 + k)]), "io.f_leg(" + std::to_string(25 + k) + ", %s);");

    // ---- tile stores (store path of the headline kernel) ------------------------------------------------------
    // The Jacobian block of the 16 nodes of a wavefront leaves as a sequence of REGISTER IMAGES: image s holds, in the lane of leg q of every
    // node, entry tileEntries[4 s + q] of that node.  A per-lane result (leg rows, base rows of a column owned by the lane's leg) is an image
    // as it stands; the base rows of the shared columns are identical in the four lanes of a node, so four of them share one image (lane q
    // keeps the q-th: io.sel4).  Images leave two at a time (io.t_put2: partner nodes exchange one value each, then ONE 16-byte store
    // instruction writes 1 KiB contiguous).
    // Every phase is SELF-CONTAINED -- an even number of complete images -- so that the phases of the program can be emitted in any order
    // (EmitQuadProgram: phaseOrder; wavefronts running different orders keep the chip's store traffic even).  What a phase lacks is filled
    // from a pool of LITERAL entries: the three position columns are the identity in the position rows and zero elsewhere (111 entries);
    // the rest of the pool (+ 3 padding slots) is stored by statements of its own at the top of the program.
    struct TileImage {
        std::vector<AD> values;  // 1: per-lane value;  4: lane q keeps values[q]
    };
    bool tilePooling = false;  // true while the position columns are recorded: their entries go to the pool
    std::vector<std::pair<int, double>> tilePool;  // (entry, literal value)
    bool tileHavePending = false;
    TileImage tilePending;
    std::vector<std::pair<int, AD>> tileShared;  // base-row entries of shared columns waiting for an image
    auto tileImage = [&](TileImage img, const std::array<int, 4>& entries) {
        const int sink = static_cast<int>(P.tileEntries.size() / 4);
        for (int q = 0; q < 4; ++q) P.tileEntries.push_back(entries[static_cast<std::size_t>(q)]);
        if (!tileHavePending) {
            tilePending = std::move(img);
            tileHavePending = true;
            return;
        }
        std::vector<tape::Id> ids;
        auto spell = [&](const TileImage& im) {
            auto ph = [&](const AD& v) {
                ids.push_back(v.Node());
                const std::size_t k = ids.size() - 1;
                return k == 0 ? std::string("%s") : k == 1 ? std::string("%t") : "%" + std::to_string(k);
            };
            if (im.values.size() == 1) return ph(im.values[0]);
            std::string t = "io.sel4(";
            for (std::size_t q = 0; q < 4; ++q) t += (q ? ", " : "") + ph(im.values[q]);
            return t + ")";
        };
        const std::string a = spell(tilePending), b = spell(img);
        tape::OutputSlot slot{ids[0], "io.t_put2(" + std::to_string(sink / 2) + ", " + a + ", " + b + ");", ids[1]};
        slot.more.assign(ids.begin() + 2, ids.end());
        P.slots.push_back(std::move(slot));
        tileHavePending = false;
    };
    auto tileSharedEntry = [&](int entry, const AD& v) {
        tileShared.emplace_back(entry, v);
        if (tileShared.size() < 4) return;
        TileImage img;
        std::array<int, 4> entries{};
        for (std::size_t q = 0; q < 4; ++q) {
            img.values.push_back(tileShared[q].second);
            entries[q] = tileShared[q].first;
        }
        tileShared.clear();
        tileImage(std::move(img), entries);
    };
    auto tilePoolEntry = [&]() -> std::pair<int, double> {  // next literal entry, or a padding slot once the pool is empty
        if (tilePool.empty()) return {-1, 0.0};
        const std::pair<int, double> e = tilePool.back();
        tilePool.pop_back();
        return e;
    };
    /// Completes the images of the phase that ends here with literal entries.
    auto tileClosePhase = [&]() {
        if (!tileStores) return;
        while (!tileShared.empty()) {
            const auto [e, c] = tilePoolEntry();
            tileSharedEntry(e, AD{c});
        }
        if (tileHavePending)
            for (int q = 0; q < 4; ++q) {
                const auto [e, c] = tilePoolEntry();
                tileSharedEntry(e, AD{c});
            }
        if (tileHavePending || !tileShared.empty()) throw std::logic_error("quad program: a phase of the tile program is not self-contained");
    };

    /// Emits the Jacobian entries this lane is responsible for, for one column.
    ///   gLocal: index of the column among the lane-local integrator inputs (0..18) or -1
    ///   yb, yl: base / own-leg rows of da/dz for the column;  ownKind: this lane owns the column
    ///   colExpr: how the column index is spelled; baseRows: emit the 13 base rows too
    int columnCounter = 0;
    int pendingWait = -1;  // consumer: message the next column phase starts by waiting for
    int sharedGroups = 0;  // merged base-row stores of shared columns emitted so far (unique names)
    auto emitColumn = [&](int gLocal, const std::vector<AD>& yb, const std::array<AD, 3>& yl, int colBase, int colLegMul, int rot, bool baseRows,
                          bool sharedColumn) {
        const std::string colArgs = std::to_string(colBase) + ", " + std::to_string(colLegMul) + ", " + std::to_string(rot);
        auto checkZero = [&](const AD& v, int rowBase, int rowLegMul) {  // an entry outside the pattern must be a literal zero
            for (int L = 0; L < 4; ++L) {
                const int r = rowBase + 3 * rowLegMul * L, c = colBase + 3 * colLegMul * ((L + rot) & 3);
                if (kOf[static_cast<std::size_t>(r * 49 + c)] < 0 && !(v.IsLiteral() && v.Literal() == 0.0))
                    throw std::runtime_error("quad program: non-zero entry outside the sparsity pattern at (" + std::to_string(r) + "," + std::to_string(c) + ")");
            }
        };
        if (baseRows && !tilePooling && (columnCounter++ % columnsPerPhase) == 0) P.phaseStarts.push_back(P.slots.size());  // a column + its rotated copies
        if (baseRows && consumer && pendingWait >= 0) {
            P.slots.push_back({noValue, "@begin:io.wait(" + std::to_string(pendingWait) + ");"});
            pendingWait = -1;
        }
        auto entry = [&](int gr) {  // integrator row gr (0..18) of this column
            AD v = gLocal >= 0 ? Gm[gr][gLocal] : AD{0.0};
            for (int k = 0; k < 6; ++k) v = v + Gm[gr][19 + k] * yb[static_cast<std::size_t>(k)];
            for (int k = 0; k < 3; ++k) v = v + Gm[gr][25 + k] * yl[static_cast<std::size_t>(k)];
            return v;
        };
        if (tileStores) {
            if (baseRows)
                for (int i = 0; i < 13; ++i) {
                    const AD v = entry(i);
                    const int row = baseRowIndex(i);
                    checkZero(v, row, 0);
                    if (tilePooling) {
                        if (!v.IsLiteral() && g.At(v.Node()).op != tape::Op::Const) throw std::logic_error("quad program: the position columns are expected to be literals");
                        tilePool.emplace_back(row * 49 + colBase, v.IsLiteral() ? v.Literal() : g.At(v.Node()).value);
                    } else if (sharedColumn) {
                        tileSharedEntry(row * 49 + colBase, v);
                    } else {  // column owned by the lane's leg: lane q writes (row, colBase + 3 q)
                        std::array<int, 4> e{};
                        for (int q = 0; q < 4; ++q) e[static_cast<std::size_t>(q)] = row * 49 + colBase + 3 * colLegMul * ((q + rot) & 3);
                        tileImage(TileImage{{v}}, e);
                    }
                }
            for (int half = 0; half < 2; ++half)
                for (int k = 0; k < 3; ++k) {
                    const int rowBase = (half ? 25 : 7) + k;
                    const AD v = entry(13 + 3 * half + k);
                    checkZero(v, rowBase, 1);
                    std::array<int, 4> e{};
                    for (int q = 0; q < 4; ++q) e[static_cast<std::size_t>(q)] = (rowBase + 3 * q) * 49 + colBase + 3 * colLegMul * ((q + rot) & 3);
                    if (tilePooling) {
                        if (!v.IsLiteral() && g.At(v.Node()).op != tape::Op::Const) throw std::logic_error("quad program: the position columns are expected to be literals");
                        for (int q = 0; q < 4; ++q) tilePool.emplace_back(e[static_cast<std::size_t>(q)], v.IsLiteral() ? v.Literal() : g.At(v.Node()).value);
                    } else {
                        tileImage(TileImage{{v}}, e);
                    }
                }
            return;
        }
        // pairStores: two entries of a column leave in ONE 16-byte store after a pairwise exchange between the lanes of
        // neighbouring nodes (quad_kernel.hpp: BufPut2) -- base rows (0,1) (2,3) (4,5) (19,20) (21,22) (23,24), row 6 alone;
        // leg rows (7 + k, 25 + k).  The sinks carry both entries; an I/O policy without paired stores writes them one by one.
        if (baseRows && pairStores) {
            for (int i = 0; i < 13; ++i) {
                const AD v = entry(i);
                checkZero(v, baseRowIndex(i), 0);
                const std::string head = sharedColumn ? "io.j_base_shared" : "io.j_base_own";
                if (i == 6) {
                    P.slots.push_back({v.Node(), head + "(" + std::to_string(baseRowIndex(i)) + ", " + colArgs + ", " + kArgs(baseRowIndex(i), 0, colBase, colLegMul, rot) + ", %s);"});
                    continue;
                }
                const AD v2 = entry(i + 1);
                checkZero(v2, baseRowIndex(i + 1), 0);
                P.slots.push_back({v.Node(),
                                   head + "2(" + std::to_string(baseRowIndex(i)) + ", " + std::to_string(baseRowIndex(i + 1)) + ", " + colArgs + ", " +
                                       kArgs(baseRowIndex(i), 0, colBase, colLegMul, rot) + ", " + kArgs(baseRowIndex(i + 1), 0, colBase, colLegMul, rot) + ", %s, %t);",
                                   v2.Node()});
                ++i;
            }
        }
        if (baseRows && !pairStores)
            for (int i = 0; i < 13; ++i) {
                const AD v = entry(i);
                checkZero(v, baseRowIndex(i), 0);
                // a shared column's base rows hold the same value in all four lanes of a node: four entries
                // are merged into ONE store instruction, lane of leg g writing entry g (names sh<g>_<group>
                // carry the first three values to the sink of the fourth)
                if (mergeSharedStores && sharedColumn && i < 12) {
                    const int g = i & 3, group = sharedGroups + (i >> 2);
                    if (g < 3) {
                        P.slots.push_back({v.Node(), "const T sh" + std::to_string(g) + "_" + std::to_string(group) + " = %s;"});
                    } else {
                        std::string rows, ks, vals;
                        for (int j = 0; j < 4; ++j) {
                            const int r = baseRowIndex(i - 3 + j);
                            rows += std::to_string(r) + ", ";
                            ks += std::to_string(kOf[static_cast<std::size_t>(r * 49 + colBase)]) + ", ";
                            if (j < 3) vals += "sh" + std::to_string(j) + "_" + std::to_string(group) + ", ";
                        }
                        P.slots.push_back({v.Node(), "io.j_base_shared4(" + rows + std::to_string(colBase) + ", " + ks + vals + "%s);"});
                    }
                    continue;
                }
                P.slots.push_back({v.Node(), std::string(sharedColumn ? "io.j_base_shared(" : "io.j_base_own(") + std::to_string(baseRowIndex(i)) + ", " + colArgs +
                                                 ", " + kArgs(baseRowIndex(i), 0, colBase, colLegMul, rot) + ", %s);"});
            }
        if (baseRows && sharedColumn) sharedGroups += 3;
        if (pairStores) {
            for (int k = 0; k < 3; ++k) {
                const AD v = entry(13 + k), v2 = entry(16 + k);
                checkZero(v, 7 + k, 1);
                checkZero(v2, 25 + k, 1);
                P.slots.push_back({v.Node(),
                                   "io.j_leg2(" + std::to_string(7 + k) + ", " + std::to_string(25 + k) + ", " + colArgs + ", " + kArgs(7 + k, 1, colBase, colLegMul, rot) + ", " +
                                       kArgs(25 + k, 1, colBase, colLegMul, rot) + ", %s, %t);",
                                   v2.Node()});
            }
            return;
        }
        for (int half = 0; half < 2; ++half)
            for (int k = 0; k < 3; ++k) {
                const int rowBase = (half ? 25 : 7) + k;
                const AD v = entry(13 + 3 * half + k);
                checkZero(v, rowBase, 1);
                P.slots.push_back({v.Node(), "io.j_leg(" + std::to_string(rowBase) + ", " + colArgs + ", " + kArgs(rowBase, 1, colBase, colLegMul, rot) + ", %s);"});
            }
    };
    const std::vector<AD> zero6(6, AD{0.0});
    const std::array<AD, 3> zero3{AD{0.0}, AD{0.0}, AD{0.0}};

    // ---- columns owned by this lane's leg: q_L, v_L (via D) and u_L ------------------------------------------------------------
    int nextMessage = 1;
    std::vector<int> accMessages;  // producer: messages whose right-hand side depends on the solved accelerations
    auto doneWith = [&](int message) {
        if (consumer && message >= 0) P.slots.push_back({noValue, "io.done(" + std::to_string(message) + ");"});
    };
    auto ownColumn = [&](int kind, int k) {  // 0: q_L, 1: v_L, 2: u_L
        std::vector<AD> rb(6, AD{0.0});
        std::array<AD, 3> rl = zero3;
        int message = -1;
        if (kind < 2) {
            const int dc = 3 * kind + k;
            if (!fused) message = nextMessage++;
            if (kind == 0) accMessages.push_back(message);
            for (int r = 0; r < 3; ++r) rl[static_cast<std::size_t>(r)] = channel(message, r, -Dm[r][dc], false);
            for (int r = 0; r < 6; ++r) rb[static_cast<std::size_t>(r)] = channel(message, 3 + r, -Dm[3 + r][dc], false);  // fb_own does not depend on leg variables
        } else {
            rl[static_cast<std::size_t>(k)] = AD{1.0};
        }
        pendingWait = message;
        const Sol s = solve(rb, rl, false);
        const int colBase = (kind == 0 ? 7 : kind == 1 ? 25 : 37) + k;
        const int gLocal = kind == 0 ? 13 + k : kind == 1 ? 16 + k : -1;
        emitColumn(gLocal, s.yb, s.yl, colBase, 1, 0, true, false);
        // the other three legs' versions of this column: their y_b arrives by rotation, only leg rows are ours
        for (int rot = 1; rot < 4; ++rot) {
            std::vector<AD> ybr(6);
            for (std::size_t r = 0; r < 6; ++r) ybr[r] = tape::QuadRot(s.yb[r], rot);
            emitColumn(-1, ybr, foreignRows(ybr), colBase, 1, rot, false, false);
        }
        doneWith(message);
        tileClosePhase();
    };
    // ---- shared columns: base twist (via D, summed over legs), quaternion (closed form), position (none) -------------------------
    auto twistColumn = [&](int k) {
        const int dc = 6 + k;
        const int message = fused ? -1 : nextMessage++;
        std::vector<AD> rb(6);
        for (int r = 0; r < 6; ++r) rb[static_cast<std::size_t>(r)] = channel(message, 3 + r, -(tape::QuadSum(Dm[3 + r][dc]) + Dm[9 + r][dc]), true);
        std::array<AD, 3> rl;
        for (int r = 0; r < 3; ++r) rl[static_cast<std::size_t>(r)] = channel(message, r, -Dm[r][dc], false);
        pendingWait = message;
        const Sol s = solve(rb, rl, true);
        emitColumn(7 + k, s.yb, s.yl, 19 + k, 0, 0, true, true);
        doneWith(message);
        tileClosePhase();
    };
    auto quaternionColumn = [&](int k) {
        std::vector<AD> yb(6, AD{0.0});
        for (int r = 0; r < 3; ++r) yb[static_cast<std::size_t>(r)] = -dGam[r][k];
        emitColumn(3 + k, yb, zero3, 3 + k, 0, 0, true, true);
        tileClosePhase();
    };
    auto positionColumn = [&](int k) { emitColumn(k, zero6, zero3, k, 0, 0, true, true); };
    if (fused && !tileStores) {
        for (int kind = 0; kind < 3; ++kind)
            for (int k = 0; k < 3; ++k) ownColumn(kind, k);
        for (int k = 0; k < 6; ++k) twistColumn(k);
        for (int k = 0; k < 4; ++k) quaternionColumn(k);
        for (int k = 0; k < 3; ++k) positionColumn(k);
    } else if (fused) {
        // tile program: the literal position columns first, into the pool the other phases complete their images from
        tilePooling = true;
        for (int k = 0; k < 3; ++k) positionColumn(k);
        tilePooling = false;
        for (int kind = 0; kind < 3; ++kind)
            for (int k = 0; k < 3; ++k) ownColumn(kind, k);
        for (int k = 0; k < 6; ++k) twistColumn(k);
        for (int k = 0; k < 4; ++k) quaternionColumn(k);
        // What is left of the pool leaves through statements at the TOP of the program ("@begin:" sinks of the first phase): a wavefront then has
        // these bytes in flight while it computes kinematics, composite inertias and the factorisation (17 % of its statements, no other stores).
        std::vector<tape::OutputSlot> literalSinks;
        while (!tilePool.empty() || (P.tileEntries.size() / 4) % 2 != 0) {
            std::string vals[2];
            for (int half = 0; half < 2; ++half) {
                std::array<int, 4> e{};
                std::array<double, 4> c{};
                for (std::size_t q = 0; q < 4; ++q) std::tie(e[q], c[q]) = tilePoolEntry();
                for (int q = 0; q < 4; ++q) P.tileEntries.push_back(e[static_cast<std::size_t>(q)]);
                auto lit = [](double v) {
                    char buf[40];
                    std::snprintf(buf, sizeof buf, "%.17g", v);
                    std::string t = buf;
                    if (t.find_first_of(".en") == std::string::npos) t += ".0";
                    return t;
                };
                vals[half] = c[0] == c[1] && c[1] == c[2] && c[2] == c[3] ? lit(c[0]) : "io.sel4(" + lit(c[0]) + ", " + lit(c[1]) + ", " + lit(c[2]) + ", " + lit(c[3]) + ")";
            }
            literalSinks.push_back({noValue, "@begin:io.t_put2(" + std::to_string(P.tileEntries.size() / 8 - 1) + ", " + vals[0] + ", " + vals[1] + ");"});
        }
        P.slots.insert(P.slots.begin(), literalSinks.begin(), literalSinks.end());
        for (std::size_t& st : P.phaseStarts) st += literalSinks.size();
        std::vector<char> seen(37 * 49, 0);
        for (int e : P.tileEntries)
            if (e >= 0) {
                if (seen[static_cast<std::size_t>(e)]) throw std::logic_error("quad program: entry stored twice in the tile program");
                seen[static_cast<std::size_t>(e)] = 1;
            }
        for (char c : seen)
            if (!c) throw std::logic_error("quad program: entry missing from the tile program");
    } else {
        // Order of the hand-over: the velocity columns first (their tangents do not involve the solved accelerations, so the
        // producer starts on them while the consumer still factorises), the joint-angle columns last.  The consumer
        // interleaves the columns it can do on its own, so that it stays a column or two behind the producer.
        for (int k = 0; k < 3; ++k) {
            ownColumn(1, k);
            ownColumn(2, k);
        }
        for (int k = 0; k < 6; ++k) {
            twistColumn(k);
            if (k < 4) quaternionColumn(k);
            if (k == 4)
                for (int j = 0; j < 3; ++j) positionColumn(j);
        }
        for (int k = 0; k < 3; ++k) ownColumn(0, k);
        if (nextMessage != kQuadMessages) throw std::logic_error("quad program: message count");
    }
    if (producer) {  // one phase per message: [wait for the ring slot / the accelerations] items... post
        P.slots.clear();
        P.phaseStarts.clear();
        for (int m = 0; m < kQuadMessages; ++m) {
            if (m) P.phaseStarts.push_back(P.slots.size());
            P.slots.push_back({noValue, "@begin:io.wait_free(" + std::to_string(m) + ");"});
            if (!accMessages.empty() && m == accMessages.front()) P.slots.push_back({noValue, "@begin:io.wait_acc();"});
            for (const auto& sl : sends[static_cast<std::size_t>(m)]) P.slots.push_back(sl);
            P.slots.push_back({noValue, "io.post(" + std::to_string(m) + ");"});
        }
    }

    // ---- package ---------------------------------------------------------------------------------------------------------------------
    std::vector<AD> roots;
    for (const auto& sl : P.slots) roots.push_back(AD::FromId(sl.value));
    for (const auto& sl : P.slots)
        if (sl.value2 != tape::kNoId) roots.push_back(AD::FromId(sl.value2));
    for (const auto& sl : P.slots)
        for (tape::Id m : sl.more) roots.push_back(AD::FromId(m));
    P.tape = tape::MakeTape(roots);
    std::size_t second = P.slots.size();
    for (std::size_t i = 0; i < P.slots.size(); ++i) {
        P.slots[i].value = P.tape.outputs[i];
        if (P.slots[i].value2 != tape::kNoId) P.slots[i].value2 = P.tape.outputs[second++];
    }
    for (auto& sl : P.slots)
        for (tape::Id& m : sl.more) m = P.tape.outputs[second++];
    return P;
}

/// Emits `template <class T, class IO> void <fn>(IO& io)` with all values of type T.
inline std::string EmitQuadProgram(const QuadProgram& P, const std::string& fnName, tape::EmitStats* stats = nullptr, bool usePhases = true,
                                   int ldsSlots = 0, int* ldsSlotsUsed = nullptr, int rematConsumers = 0, int rematDepth = 0, int prefetch = 0,
                                   int uniformSlots = 0, int* uniformSlotsUsed = nullptr, bool prefetchAcrossPhases = false, bool interleaveSinks = false,
                                   const std::vector<int>* phaseOrder = nullptr, bool explicitFma = false) {
    if (explicitFma) {  // contraction decided here, not by the compiler (tape::FuseMultiplyAdd): the function is compiled with `fp contract(off)`
        QuadProgram fusedProgram = P;
        std::vector<tape::Id> roots;
        for (const auto& sl : P.slots) {
            roots.push_back(sl.value);
            if (sl.value2 != tape::kNoId) roots.push_back(sl.value2);
            for (tape::Id m : sl.more) roots.push_back(m);
        }
        tape::FuseMultiplyAdd(P.tape.graph, roots, fusedProgram.tape.graph);
        std::size_t next = 0;
        for (auto& sl : fusedProgram.slots) {
            sl.value = roots[next++];
            if (sl.value2 != tape::kNoId) sl.value2 = roots[next++];
            for (tape::Id& m : sl.more) m = roots[next++];
        }
        std::string text = EmitQuadProgram(fusedProgram, fnName, stats, usePhases, ldsSlots, ldsSlotsUsed, rematConsumers, rematDepth, prefetch, uniformSlots, uniformSlotsUsed,
                                           prefetchAcrossPhases, interleaveSinks, phaseOrder, false);
        const std::string head = "(IO& io) {\n";
        const std::size_t at = text.find(head);
        if (at == std::string::npos) throw std::logic_error("quad program: function head not found");
        text.insert(at + head.size(), "#pragma clang fp contract(off)\n");
        return text;
    }
    // inputs are read ONCE into locals (an accessor call per use would be re-issued as a memory load
    // after every store, since the compiler cannot prove the output buffers do not alias them)
    std::vector<char> used(P.inputNames.size(), 0);
    {
        const tape::Graph& g = P.tape.graph;
        std::vector<char> live(g.Size(), 0);
        for (const auto& sl : P.slots) {
            live[static_cast<std::size_t>(sl.value)] = 1;
            if (sl.value2 != tape::kNoId) live[static_cast<std::size_t>(sl.value2)] = 1;
            for (tape::Id m : sl.more) live[static_cast<std::size_t>(m)] = 1;
        }
        for (std::size_t i = g.Size(); i-- > 0;) {
            if (!live[i]) continue;
            const tape::Node& nd = g.At(static_cast<tape::Id>(i));
            if (nd.op == tape::Op::Input) used[static_cast<std::size_t>(nd.a)] = 1;
            if (tape::Arity(nd.op) == 0) continue;
            for (tape::Id o : {nd.a, nd.b, nd.c, nd.d})
                if (o != tape::kNoId) live[static_cast<std::size_t>(o)] = 1;
        }
    }
    std::vector<std::string> names(P.inputNames.size());
    std::string prologue;
    for (std::size_t i = 0; i < names.size(); ++i) {
        const bool late = i < P.inputLate.size() && P.inputLate[i];
        names[i] = late ? P.inputNames[i] : "in" + std::to_string(i);
        if (used[i] && !late) prologue += "    const T " + names[i] + " = " + P.inputNames[i] + ";\n";
    }
    tape::Emitter em{P.tape.graph, names};
    em.SetRereadInputs(P.inputLate);
    em.SetInterleaveSinks(interleaveSinks);
    // phases (primal + one per owned / shared column) separated by scheduling barriers, no LDS home:
    // per-lane state fits the register file, the barriers only stop the scheduler from interleaving columns
    std::vector<std::vector<tape::OutputSlot>> phases;
    {
        std::size_t next = 0;
        std::vector<std::size_t> starts = P.phaseStarts;
        starts.push_back(P.slots.size());
        for (std::size_t st : starts) {
            if (st > next) phases.emplace_back(P.slots.begin() + static_cast<std::ptrdiff_t>(next), P.slots.begin() + static_cast<std::ptrdiff_t>(st));
            next = st;
        }
    }
    if (phaseOrder) {  // the phases in another order (programs whose phases are self-contained: tile stores)
        if (phaseOrder->size() != phases.size()) throw std::logic_error("quad program: phase order of " + std::to_string(phaseOrder->size()) + " entries for " + std::to_string(phases.size()) + " phases");
        std::vector<std::vector<tape::OutputSlot>> permuted;
        for (int ph : *phaseOrder) permuted.push_back(phases.at(static_cast<std::size_t>(ph)));
        phases = std::move(permuted);
    }
    int slotsUsed = 0;
    std::string body = usePhases ? em.EmitPhased(phases, ldsSlots, slotsUsed, rematConsumers, rematDepth, prefetch, "    ", &P.inputUniform, uniformSlots, uniformSlotsUsed, prefetchAcrossPhases)
                                 : em.Emit(P.slots);
    if (ldsSlotsUsed) *ldsSlotsUsed = slotsUsed;
    // the straight-line emitter declares `const double vN`; make the value type generic
    std::string out;
    out.reserve(body.size() + 256);
    const std::string from = "const double v";
    std::size_t pos = 0;
    for (;;) {
        const std::size_t hit = body.find(from, pos);
        if (hit == std::string::npos) break;
        out.append(body, pos, hit - pos);
        out += "const T v";
        pos = hit + from.size();
    }
    out.append(body, pos, std::string::npos);
    if (stats) *stats = em.Stats();
    return "template <class T, class IO>\n__host__ __device__ inline void " + fnName + "(IO& io) {\n" + prologue + out + "}\n";
}

}  // namespace ungar_amd::codegen
