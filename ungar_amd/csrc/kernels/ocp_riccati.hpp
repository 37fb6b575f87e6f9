// ungar_amd :: batched solve of the SQP's quadratic subproblem for optimal-control structure.
//
// The reference hands every QP of its soft SQP to OSQP as one sparse problem (include/ungar/optimization/
// soft_sqp.hpp:143-158):   min_d 1/2 d^T H d + g^T d   s.t.  J_g d = -g(z),   H = hess f + J_h^T diag(b'') J_h + 1e-6 I.
// For a shooting problem with stage-wise cost and soft inequalities, z = [x_0..x_N | u_0..u_{N-1}],
//     g = [x_0 - x_m ; x_{k+1} - f(x_k, u_k)]   (example/mpc/quadrotor.example.cpp:246-266),
// H is block diagonal over the knots and J_g block bidiagonal, so the KKT system is solved EXACTLY by the discrete
// Riccati recursion -- O(N (nx + nu)^3) per instance, no fill-in, no iteration count to tune:
//     dx_0 given,  dx_{k+1} = A_k dx_k + B_k du_k + b_k                    (A|B = node Jacobian, b_k = f_k - x_{k+1})
//     P_N = W_N, p_N = w_N;   for k = N-1 .. 0, with AB = [A_k | B_k]:
//       H = W_k + AB^T P AB,   h = w_k + AB^T (P b_k + p)                  (n x n, n = nx + nu)
//       R = H_uu = L L^T,  [K | kff] = -R^-1 [H_ux | h_u],   P <- H_xx + H_ux^T K,   p <- h_x + H_ux^T kff
//     forward:  du_k = K_k dx_k + kff_k,  dx_{k+1} = AB [dx_k; du_k] + b_k.
// One workgroup (one wavefront for the small blocks, four for 13 + 24 and 37 + 12) owns one MPC instance: every matrix of the recursion lives in LDS, the lanes share the entries of
// each product (Exec::ForEach) and meet at workgroup barriers; thousands of instances fill the device.
//
// The recursion is written once, generic over an execution policy: DeviceExec (ocp_riccati.hip) strides the lanes of a
// workgroup over the index range; a sequential host policy exists ONLY in tests/cpp/riccati_host.cpp so that the very same
// source can be pinned against a dense KKT solve without a GPU (test infrastructure, like tests/cpp/quad_sim.cpp).
//
// Execution policy:
//   ForEach(n, f)            f(i) for i in [0, n), spread over the lanes, followed by a workgroup barrier on the scratch memory
//   GlobalSync()             workgroup barrier that also orders the workgroup's global writes before its later global reads
//   ForEachNoSync(n, f)      the same without the closing barrier (followed by other loops and one Barrier())
//   kLanes                   lanes that share a ForEach (device: the workgroup size; picks the register tiles of the two large products)
//   kPrefetch                whether the policy stages a knot's operands in registers (below); without it they are read in place
//   kAhead                   with kPrefetch: stage the NEXT knot's operands while this one is processed (else: this knot's, at its top)
//   Stage<SLOTS>             per-lane registers for a strided global read of up to 64 * SLOTS doubles
//   Fetch(n, f, stage)       issues the loads stage <- f(i); no barrier, nothing waits for the data
//   Commit(n, stage, dst)    dst[i] <- stage (no barrier);  Barrier() synchronises the scratch memory
//   kDma (optional)          the policy copies global memory into the scratch memory asynchronously, without registers:
//     DmaFetch(n, addr, dst)        every lane group takes part: dst[i] <- *addr(i), i < n; returns at once, nothing waits
//     DmaWait()                     this lane group's copies have landed (a Barrier() then publishes them to the workgroup)
//     kDmaOwners, DmaFetchOne(o, n, addr, dst), DmaWaitOne(o)   the same issued / awaited by ONE of kDmaOwners lane groups
//   kDmaSelf (optional)      one lane group per instance that issues its own copies and waits for them by COUNT:
//     DmaFetchSelf(n, addr, dst), DmaWaitSelf<YOUNGER>()   all copies but the youngest YOUNGER copy instructions (32 doubles each) have landed
//   A knot's operands are then requested the moment their destination dies in the PREVIOUS knot ([A|B] after the H phase, b and the
//   packed stage Hessian after the P [A|B] phase into the retired cost-to-go buffer), and the forward pass runs kDmaOwners knots ahead.
// With prefetching, the operands of knot k - 1 ([A|B], W, w, b: one HBM round trip each) are in flight while knot k is
// factorised, instead of stalling every phase that touches them (0.75 -> see DESIGN.md for 4096 x 30 quadrotor knots).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UNGAR_HD __host__ __device__
#else
#define UNGAR_HD
#endif

#include <cmath>

#ifndef UNGAR_RICCATI_TILE_MIN_NX
#define UNGAR_RICCATI_TILE_MIN_NX 6  // register tiles in the two large products for compile-time sizes from this nx on
#endif

#ifndef UNGAR_RICCATI_MFMA_MIN_NX
#define UNGAR_RICCATI_MFMA_MIN_NX 24  // state dimension from which the four-wavefront kernels run their two large products on the matrix cores
#endif

#ifndef UNGAR_RICCATI_BLOCKED_MIN_NX
#define UNGAR_RICCATI_BLOCKED_MIN_NX 24  // blocked Cholesky from this state dimension on: such blocks fit two workgroups per CU by their LDS anyway, so its registers cost
#endif                                 // no occupancy (13 + 24 runs five workgroups per CU on the rank-one phases: 4.8 ms against 5.7 blocked, r03e)

#ifndef UNGAR_RICCATI_BLOCKED_LDLT_MIN_NU
#define UNGAR_RICCATI_BLOCKED_LDLT_MIN_NU 16  // input dimension from which the four-wavefront kernels factorise R by the blocked L D L^T with matrix-core trailing updates
#endif
#ifndef UNGAR_RICCATI_BLOCK
#define UNGAR_RICCATI_BLOCK 8  // diagonal block of the blocked Cholesky for input dimensions above 12 (factorised in registers by every lane)
#endif

namespace ungar_amd::kernels {

/// Strided view (instance, knot, element) of one operand, in doubles.
struct RiccatiView {
    double* base;
    long long bs, ks, es;
    UNGAR_HD double& at(long long b, long long k, long long e) const {
        return base[b * bs + k * ks + e * es];
    }
};

struct RiccatiArgs {
    int nx, nu, N;
    long long batch;
    RiccatiView jac;    // dense nx x (nx+nu) row-major block per node (b, k), k < N   -- what ungar_model_dense_jacobian writes
    RiccatiView b;      // nx per node: affine term of the linearised dynamics
    RiccatiView hess;   // (nx+nu)^2 per node, row-major; only entries with row <= col are read (upper triangle, function.hpp:232-235)
    RiccatiView grad;   // nx+nu per node
    RiccatiView hessN;  // terminal: nx^2 per instance (upper triangle read) -- base may be null: zero terminal cost
    RiccatiView gradN;  // terminal: nx per instance -- base may be null
    RiccatiView dx0;    // nx per instance
    RiccatiView dX;     // out: (N+1) nx per instance, element e of knot k at (b, k, e)
    RiccatiView dU;     // out: N nu per instance
    double* gains;      // workspace: batch * N * nu * (nx + 1) doubles (feedback gains and feed-forward terms)
    double regularization;  // added to the diagonal of every H and of P_N (the reference's 1e-6 I, soft_sqp.hpp:149-151)
    int* status;        // per instance: 0 ok, k+1 = R_k not positive definite at knot k (may be null)
    // Stage equality rows  E_k [dx_k; du_k] + e_k = 0  (k < N; the foot-contact rows of example/mpc/quadruped.example.cpp:279-304 enter the
    // reference's QP as hard equalities next to the dynamics, soft_sqp.hpp:155-157).  ne rows per node, dense row-major ne x (nx+nu); a row
    // that is identically zero (an inactive contact) is skipped; the input parts of the other rows must be linearly independent.
    int ne = 0;
    RiccatiView eq{nullptr, 0, 0, 0};   // ne x (nx+nu) per node
    RiccatiView eqv{nullptr, 0, 0, 0};  // ne per node
    int hessNld = 0;    // leading dimension of hessN (0: nx) -- lets the terminal block be the leading nx x nx part of a wider stage block
};

/// Doubles of LDS (or host scratch) one instance needs.
UNGAR_HD inline int RiccatiScratchDoubles(int nx, int nu, int ne = 0) {
    const int n = nx + nu;
    return nx * n /*AB*/ + n * n /*H*/ + 2 * nx * nx /*P, Pn*/ + nx * n /*PAB*/ + 2 * nx /*p, pn*/ + nx /*t*/ + n /*h*/ + nx /*bk*/ + (nu + ne) * (nx + 1) /*K|kff (and the multiplier feedback)*/ +
           nx + nu /*dx, du*/ + nx /*dxn*/ + nu + ne /*pivots*/ + n /*next knot's stage gradient (kDma)*/ +
           (((n + 1) / 2) * (n + 1) > nx * nx ? ((n + 1) / 2) * (n + 1) : 0) /*next knot's folded stage Hessian where the retired cost-to-go buffer is too small (kDma)*/ +
           ne * (n + ne) + ne /*equality rows [C | D | 0] and their values*/;
}

/// Register tile of the two large products: the smallest one with which all tiles of a rows x cols product fit the lanes of the
/// workgroup ONCE (the accompanying dot products t = P b + p, h += [A|B]^T t follow as one item each: a short second pass at most).
/// The fixed 2 x 4 / 2 x 2 tiles needed a second pass of TILES for H: 374 items on 256 lanes for the 37 + 12 block, 98 on 64 for 13 + 4.
/// (Measured and dropped: folding the dot products onto the spare lanes of the first pass, several per lane -- as consecutive chains
/// they outlast the tiles, as interleaved chains the 37 + 12 phase still ran 40 % longer than with the short second pass.)
struct RiccatiTile {
    int ti, tc;
};
constexpr RiccatiTile RiccatiChooseTile(int rows, int cols, int lanes) {
    constexpr RiccatiTile candidates[] = {{2, 2}, {2, 3}, {2, 4}, {3, 4}, {4, 4}};
    for (const RiccatiTile t : candidates)
        if (((rows + t.ti - 1) / t.ti) * ((cols + t.tc - 1) / t.tc) <= lanes) return t;
    return rows >= 24 ? RiccatiTile{2, 4} : RiccatiTile{2, 2};  // more tiles than lanes either way: several passes
}

template <class Exec>
constexpr bool RiccatiExecHasDma() {
    if constexpr (requires { Exec::kDma; }) return Exec::kDma;
    else return false;
}

template <class Exec>
constexpr bool RiccatiExecHasMatrixCores() {
    if constexpr (requires { Exec::kMatrixCores; }) return Exec::kMatrixCores;
    else return false;
}

/// State dimension from which the policy runs the two large products on the matrix cores (0: never).
template <class Exec>
constexpr int RiccatiMatrixCoresFrom() {
    if constexpr (RiccatiExecHasMatrixCores<Exec>()) {
        if constexpr (requires { Exec::kMatrixCoresMinNx; }) return Exec::kMatrixCoresMinNx;
        else return UNGAR_RICCATI_MFMA_MIN_NX;
    } else {
        return 0;
    }
}

/// R = H_uu by the BLOCKED L D L^T with trailing updates on the matrix cores (policies with a FactorGainsBlocked member: the four-wavefront device
/// kernels), for input dimensions that are a multiple of 4 from UNGAR_RICCATI_BLOCKED_LDLT_MIN_NU on.
template <class Exec, int NX, int NU, int NE>
constexpr bool RiccatiFactorBlockedOnMatrixCores() {
    if constexpr (requires { Exec::kFactorBlocked; })
        return Exec::kFactorBlocked && NX > 0 && NE == 0 && NU >= UNGAR_RICCATI_BLOCKED_LDLT_MIN_NU && NU % 4 == 0 && NU <= 32 && (NU + 15) / 16 * 16 + NX + 1 <= 64;
    else
        return false;
}

/// Lanes that share one inner product of the forward pass (policies with a SplitDots member; 1: one lane per row).
template <class Exec, int NX, int NU>
constexpr int RiccatiSplitDotParts() {
    if constexpr (requires { Exec::kSplitDots; }) {
        constexpr int rows = NX > NU ? NX : NU;
        return !Exec::kSplitDots ? 1 : Exec::kLanes >= 4 * rows ? 4 : Exec::kLanes >= 2 * rows ? 2 : 1;
    } else {
        return 1;
    }
}

template <class Exec>
constexpr bool RiccatiExecHasSelfDma() {
    if constexpr (requires { Exec::kDmaSelf; }) return Exec::kDmaSelf;
    else return false;
}

/// Upper triangle (r <= c) of a symmetric n x n matrix folded into a ((n + 1) / 2) x (n + 1) rectangle: row q of the rectangle holds
/// row q of the triangle (n - q entries) followed by row n - 1 - q (q + 1 entries) -- decodable with one division by n + 1.
UNGAR_HD inline int RiccatiFoldedIndex(int n, int r, int c) {
    // row q = min(r, n - 1 - r) of the rectangle; the second half of a row starts r + 1 words in (branch-free: selects, no jumps)
    const int mirrored = n - 1 - r, q = r <= mirrored ? r : mirrored;
    return q * (n + 1) + (c - r) + (r <= mirrored ? 0 : r + 1);
}
/// Inverse: index in the rectangle -> r * n + c of the entry it holds, or -1 for the unused tail of the middle row (n odd).
UNGAR_HD inline int RiccatiFoldedSource(int n, int i) {
    const int q = i / (n + 1), j = i - q * (n + 1);
    if (j < n - q) return q * n + q + j;
    const int r = n - 1 - q;
    if (r == q) return -1;
    return r * n + r + (j - (n - q));
}

/// Phase clocks (diagnostic builds only: a policy with a Mark(id) member accumulates the cycles since its previous call under `id`).
template <class Exec>
UNGAR_HD inline void RiccatiMark(Exec& ex, int id) {
    if constexpr (requires { ex.Mark(id); }) ex.Mark(id);
}

/// The whole recursion for instance `inst`; `scratch` holds RiccatiScratchDoubles(nx, nu) doubles private to the workgroup.
/// NX / NU > 0 fix the sizes at compile time (the index arithmetic -- a division and a remainder by nx + nu per matrix entry --
/// and the short inner products then cost a fraction of the generic code, which is issue-bound on them); 0 = read them from `a`.
/// NE: equality rows per knot for the compile-time sizes (with NX == 0 they are read from `a` as well).
template <class Exec, int NX = 0, int NU = 0, int NE = 0>
UNGAR_HD void RiccatiInstance(const RiccatiArgs& a, long long inst, double* scratch, Exec& ex) {
    const int nx = NX > 0 ? NX : a.nx, nu = NU > 0 ? NU : a.nu, n = nx + nu, N = a.N, nk = nx + 1;
    const int ne = NX > 0 ? NE : a.ne, nuE = nu + ne, ldq = n + ne;  // (u, lambda) block of the stage KKT matrix: rows / columns nx .. nx + nuE - 1
    double* AB = scratch;
    double* H = AB + nx * n;
    double* P = H + n * n;
    double* Pn = P + nx * nx;
    double* PAB = Pn + nx * nx;
    double* p = PAB + nx * n;
    double* pn = p + nx;
    double* t = pn + nx;
    double* h = t + nx;
    double* bk = h + n;
    double* K = bk + nx;  // nu x (nx + 1): [K | kff]
    double* dx = K + nuE * nk;
    double* du = dx + nx;
    double* dxn = du + nu;
    double* piv = dxn + nx;  // diagonal of the Cholesky factor of R
    double* wn = piv + nuE;  // kDma: stage gradient of the next knot
    double* wf = wn + n;     // kDma: folded stage Hessian of the next knot, for sizes where it does not fit the retired cost-to-go buffer
    double* Eq = wf + (((n + 1) / 2) * (n + 1) > nx * nx ? ((n + 1) / 2) * (n + 1) : 0);  // ne x (n + ne): [C | D | 0], eliminated in place with R
    double* ev = Eq + ne * ldq;
    // Row i of the (u, lambda) block and of its couplings: entry (i, j) at rowOf(i)[nx + j], the state coupling (H_ux or C) at rowOf(i)[c], c < nx.
    auto rowOf = [&](int i) -> double* {
        if constexpr (NX > 0 && NE == 0) return H + (nx + i) * n;
        else return i < nu ? H + (nx + i) * n : Eq + (i - nu) * ldq;
    };
    auto rhsOf = [&](int i) -> double {  // affine term of row i: h_u or e
        if constexpr (NX > 0 && NE == 0) return h[nx + i];
        else return i < nu ? h[nx + i] : ev[i - nu];
    };
    // Pivot j of the L D L^T: positive in the input block; negative in the multiplier block (minus a Schur complement D R^-1 D^T),
    // or EXACTLY zero for a row that is identically zero -- such a row is skipped (reciprocal 0: it neither updates nor is updated).
    auto pivotReciprocal = [&](double d, int j) -> double {
        if (j < nu) return 1.0 / (d > 0.0 ? d : 1.0);
        return d < 0.0 ? 1.0 / d : 0.0;
    };
    auto pivotBad = [&](double d, int j) -> bool { return j < nu ? !(d > 0.0) : !(d <= 0.0); };
    auto loadEqualityRows = [&](int k) {  // (no closing barrier)
        if (ne > 0)
            ex.ForEachNoSync(ne * ldq + ne, [&](int idx) {
                if (idx < ne * ldq) {
                    const int j = idx / ldq, c = idx - j * ldq;
                    Eq[idx] = c < n ? a.eq.at(inst, k, j * n + c) : 0.0;
                } else {
                    ev[idx - ne * ldq] = a.eqv.at(inst, k, idx - ne * ldq);
                }
            });
    };
    double* gains = a.gains + inst * static_cast<long long>(N) * nu * nk;
    int failed = 0;
    // Asynchronous operand copies (policies with kDma, compile-time sizes).  The stage Hessian of the next knot is parked, folded,
    // in the cost-to-go buffer that retired after the P [A|B] phase (requested right after that phase) -- when it fits there;
    // otherwise in a buffer of its own, requested once the H phase has consumed the current one.
    constexpr bool dma = NX >= UNGAR_RICCATI_TILE_MIN_NX && RiccatiExecHasDma<Exec>();
    constexpr int nFold = ((NX + NU + 1) / 2) * (NX + NU + 1);
    constexpr bool foldW = dma, foldInP = dma && nFold <= NX * NX;
    auto foldedSource = [&](int k) {
        return [&a, inst, k, n](int i) {
            // RiccatiFoldedSource without branches (this runs once per copy instruction of every wavefront, on the critical path of the phase that issues
            // the copies): a word of the rectangle belongs to row q or to row n - 1 - q; the unused tail of the middle row (n odd) copies valid upper entries
            // of that row, which nobody reads
            const int q = i / (n + 1), j = i - q * (n + 1);
            const bool first = j < n - q;
            const int r = first ? q : n - 1 - q, off = first ? j : j - (n - q);
            return &a.hess.at(inst, k, r * (n + 1) + off);
        };
    };

    // terminal cost-to-go
    ex.ForEach(nx * nx, [&](int idx) {
        const int i = idx / nx, j = idx % nx;
        double v = 0.0;
        const int ldN = a.hessNld > 0 ? a.hessNld : nx;
        if (a.hessN.base) v = i <= j ? a.hessN.at(inst, 0, i * ldN + j) : a.hessN.at(inst, 0, j * ldN + i);
        P[idx] = v + (i == j ? a.regularization : 0.0);
    });
    ex.ForEach(nx, [&](int i) { p[i] = a.gradN.base ? a.gradN.at(inst, 0, i) : 0.0; });

    typename Exec::StageAB sAB;
    typename Exec::StageW sW;
    typename Exec::StageV sw, sb;
    // The stage Hessian W_k is brought straight into H and the stage gradient w_k into h: both are updated in place.
    auto fetchKnot = [&](int k) {
        ex.Fetch(nx * n, [&](int idx) { return a.jac.at(inst, k, idx); }, sAB);
        ex.Fetch(n * n, [&](int idx) { return a.hess.at(inst, k, idx); }, sW);
        ex.Fetch(n, [&](int c) { return a.grad.at(inst, k, c); }, sw);
        ex.Fetch(nx, [&](int i) { return a.b.at(inst, k, i); }, sb);
    };
    auto commitKnot = [&] {
        ex.Commit(nx * n, sAB, AB);
        ex.Commit(n * n, sW, H);
        ex.Commit(n, sw, h);
        ex.Commit(nx, sb, bk);
        ex.Barrier();
    };
    // kPrefetch: operands go through register stages (all loads of a knot issued back to back).  kAhead: the stage is filled one
    // knot early and committed after the knot (latency hidden, registers held across the knot); otherwise filled and committed at
    // the top of the knot (one exposed round trip per knot instead of one per load, no registers held).
    constexpr bool ahead = Exec::kPrefetch && Exec::kAhead;
    if constexpr (ahead) {
        fetchKnot(N - 1);
        commitKnot();
    }
    [[maybe_unused]] auto dmaAfterPab = [&](int k, double* retired) {  // b_k's and (folded) W_k's destinations are free once P [A|B] and t exist
        if constexpr (dma) {
            ex.DmaFetch(nx, [&](int i) { return &a.b.at(inst, k, i); }, bk);
            if constexpr (foldInP) ex.DmaFetch(nFold, foldedSource(k), retired);
        }
    };
    [[maybe_unused]] auto dmaAfterH = [&](int k) {  // [A|B]'s once H exists; the gradient (and a folded W outside P) has its own buffer but one copy in flight
        if constexpr (dma) {
            if constexpr (requires { ex.DmaFetchContiguous(0, static_cast<const double*>(nullptr), AB); }) {
                // contiguous [A|B] block (element stride 1, what the stage-QP kernels write): 16 bytes per lane and copy instruction, a quarter of the instructions
                if (a.jac.es == 1) ex.DmaFetchContiguous(nx * n, &a.jac.at(inst, k, 0), AB);
                else ex.DmaFetch(nx * n, [&](int idx) { return &a.jac.at(inst, k, idx); }, AB);
            } else {
                ex.DmaFetch(nx * n, [&](int idx) { return &a.jac.at(inst, k, idx); }, AB);
            }
            ex.DmaFetch(n, [&](int c) { return &a.grad.at(inst, k, c); }, wn);
            if constexpr (!foldInP) ex.DmaFetch(nFold, foldedSource(k), wf);
        }
    };
    if constexpr (dma) {
        dmaAfterPab(N - 1, Pn);
        dmaAfterH(N - 1);
    }
    // Policies with SetCopyWaves (the four-wavefront device kernels): ALL copies of the next knot are requested after the H phase, by the wavefronts
    // other than the first -- the factorisation that follows keeps the first one busy (its lanes factorise R in registers, or run the panel
    // phases of the blocked route) while the others wait at a barrier, so the issue slots of the copies (address generation, M0, the instruction:
    // ~10 per copy, ~100 copies per knot) come off the critical path.  The destinations are free by then and the data is not needed before the next knot.
    // (Only where the factorisation is the register one: the blocked route uses all wavefronts in its update phases and measured 3 % SLOWER with deferred copies.)
    constexpr bool deferCopies = dma && NU > 0 && NU <= 12 && NE == 0 && requires { ex.SetCopyWaves(true); };
    RiccatiMark(ex, 0);
    for (int k = N - 1; k >= 0; --k) {
        if constexpr (dma) {
            ex.DmaWait();
            if constexpr (!foldW) {
                ex.ForEachNoSync(n * n, [&](int idx) { H[idx] = a.hess.at(inst, k, idx); });
                ex.ForEachNoSync(n, [&](int idx) { h[idx] = a.grad.at(inst, k, idx); });
            }
            loadEqualityRows(k);
            ex.Barrier();
        } else if constexpr (ahead) {
            if (k > 0) fetchKnot(k - 1);  // in flight while knot k is processed
            if (ne > 0) {
                loadEqualityRows(k);
                ex.Barrier();
            }
        } else if constexpr (Exec::kPrefetch) {
            fetchKnot(k);
            loadEqualityRows(k);
            commitKnot();
        } else {
            // one loop per operand, ONE barrier: inside a loop every iteration is `LDS[i] = global[i]`, which the compiler unrolls into
            // a batch of loads followed by the stores; a single loop over all four operands with an if-chain serialised them
            // (17 dependent round trips per lane for the 37 + 12 block)
            ex.ForEachNoSync(nx * n, [&](int idx) { AB[idx] = a.jac.at(inst, k, idx); });
            ex.ForEachNoSync(n * n, [&](int idx) { H[idx] = a.hess.at(inst, k, idx); });
            ex.ForEachNoSync(n, [&](int idx) { h[idx] = a.grad.at(inst, k, idx); });
            ex.ForEachNoSync(nx, [&](int idx) { bk[idx] = a.b.at(inst, k, idx); });
            loadEqualityRows(k);
            ex.Barrier();
        }
        RiccatiMark(ex, 1);  // operands of the knot
        if constexpr (RiccatiMatrixCoresFrom<Exec>() > 0 && NX >= RiccatiMatrixCoresFrom<Exec>()) {
            // The two large products on the FP64 matrix cores (policies that have them: the four-wavefront device kernels).  What this
            // buys is not arithmetic rate -- v_mfma_f64_16x16x4_f64 and v_fma_f64 peak alike on gfx950 -- but operand traffic: a wavefront
            // reads two LDS words per lane for 1024 multiply-adds instead of (TI + TC) per TI x TC (these phases were bound by LDS issue,
            // 47 % of the 37 + 12 recursion), and no index arithmetic is left in the inner loop.  b rides along as column n of [A|B]
            // (t = P b + p) and t as column n of P [A|B] (h = w + [A|B]^T t): both fall into tiles that are computed anyway.
            ex.template ProductPab<NX, NU>(P, AB, bk, p, PAB, t);
            RiccatiMark(ex, 2);  // P [A|B]
            if constexpr (dma) {
                if constexpr (!deferCopies) {
                    if (k > 0) dmaAfterPab(k - 1, P);
                }
            }
            const double* Wsrc = foldW ? (foldInP ? Pn : wf) : nullptr;
            ex.template ProductH<NX, NU, foldW>(AB, PAB, t, Wsrc, foldW ? wn : h, a.regularization, H, h);
        } else if constexpr (NX >= UNGAR_RICCATI_TILE_MIN_NX) {
            // Sizes fixed at compile time: TI x TC register tiles (RiccatiChooseTile; 2 x 4: eight multiply-adds per six LDS reads instead of
            // per sixteen) and no bounds checks inside the product (a tile on the edge reads past its row / matrix into the neighbouring
            // scratch arrays, which is harmless: only the stores are guarded).  A tile's columns are INTERLEAVED (tc, tc + tilesC, ...):
            // neighbouring lanes then read neighbouring LDS words.  With four contiguous columns per lane the lanes of a read were 32
            // bytes apart and, paired into ds_read2_b64 (32-bank mode), collided four ways: SQ_LDS_BANK_CONFLICT was 4x the LDS issue cycles.
            constexpr RiccatiTile tileP = RiccatiChooseTile(NX, NX + NU, Exec::kLanes);
            constexpr int TI = tileP.ti, TC = tileP.tc, tilesI = (NX + TI - 1) / TI, tilesC = (NX + NU + TC - 1) / TC, tilesP = tilesI * tilesC;
            ex.ForEach(tilesP + nx, [&](int idx) {
                if (idx < tilesP) {
                    const int i0 = idx / tilesC, c0 = idx % tilesC;  // rows i0 + a2 tilesI, columns c0 + b2 tilesC
                    double acc[TI][TC] = {};
#pragma unroll 4
                    for (int m = 0; m < nx; ++m) {
                        double pv[TI], bv[TC];
                        for (int a2 = 0; a2 < TI; ++a2) pv[a2] = P[(i0 + a2 * tilesI) * nx + m];
                        for (int b2 = 0; b2 < TC; ++b2) bv[b2] = AB[m * n + c0 + b2 * tilesC];
                        for (int a2 = 0; a2 < TI; ++a2)
                            for (int b2 = 0; b2 < TC; ++b2) acc[a2][b2] += pv[a2] * bv[b2];
                    }
                    for (int a2 = 0; a2 < TI; ++a2)
                        for (int b2 = 0; b2 < TC; ++b2)
                            if (i0 + a2 * tilesI < nx && c0 + b2 * tilesC < n) PAB[(i0 + a2 * tilesI) * n + c0 + b2 * tilesC] = acc[a2][b2];
                } else {
                    const int i = idx - tilesP;
                    double acc = p[i];
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += P[i * nx + m] * bk[m];
                    t[i] = acc;
                }
            });
            RiccatiMark(ex, 2);  // P [A|B]
            if constexpr (dma) {
                if constexpr (!deferCopies) {
                    if (k > 0) dmaAfterPab(k - 1, P);  // P retired with the phase above; the folded W_k read below sits in Pn
                }
            }
            // H = W + AB^T PAB: all tiles (interleaved rows and columns straddle the diagonal); entries r <= c are updated and mirrored
            constexpr RiccatiTile tileH = RiccatiChooseTile(NX + NU, NX + NU, Exec::kLanes);
            constexpr int HI = tileH.ti, HC = tileH.tc, tilesR = (NX + NU + HI - 1) / HI, tilesD = (NX + NU + HC - 1) / HC, tilesH = tilesR * tilesD;
            ex.ForEach(tilesH + n, [&](int idx) {
                if (idx < tilesH) {
                    const int r0 = idx / tilesD, c0 = idx % tilesD;
                    double acc[HI][HC] = {};
#pragma unroll 4
                    for (int m = 0; m < nx; ++m) {
                        double lv[HI], qv[HC];
                        for (int a2 = 0; a2 < HI; ++a2) lv[a2] = AB[m * n + r0 + a2 * tilesR];
                        for (int b2 = 0; b2 < HC; ++b2) qv[b2] = PAB[m * n + c0 + b2 * tilesD];
                        for (int a2 = 0; a2 < HI; ++a2)
                            for (int b2 = 0; b2 < HC; ++b2) acc[a2][b2] += lv[a2] * qv[b2];
                    }
                    for (int a2 = 0; a2 < HI; ++a2)
                        for (int b2 = 0; b2 < HC; ++b2) {
                            const int r = r0 + a2 * tilesR, c = c0 + b2 * tilesD;
                            if (r <= c && c < n) {
                                double wv;
                                if constexpr (foldW) wv = (foldInP ? Pn : wf)[RiccatiFoldedIndex(n, r, c)];
                                else wv = H[r * n + c];
                                const double e = wv + (r == c ? a.regularization : 0.0) + acc[a2][b2];
                                H[r * n + c] = e;
                                H[c * n + r] = e;
                            }
                        }
                } else {
                    const int c = idx - tilesH;
                    double acc = foldW ? wn[c] : h[c];
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += AB[m * n + c] * t[m];
                    h[c] = acc;
                }
            });
        } else {
            // PAB = P AB;  t = P b + p
            ex.ForEach(nx * n + nx, [&](int idx) {
                if (idx < nx * n) {
                    const int i = idx / n, c = idx % n;
                    double acc = 0.0;
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += P[i * nx + m] * AB[m * n + c];
                    PAB[idx] = acc;
                } else {
                    const int i = idx - nx * n;
                    double acc = p[i];
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += P[i * nx + m] * bk[m];
                    t[i] = acc;
                }
            });
            // H = W + AB^T PAB in place (computed for r <= c, mirrored: only the upper triangle of W is read);  h = w + AB^T t in place
            ex.ForEach(n * n + n, [&](int idx) {
                if (idx < n * n) {
                    const int r = idx / n, c = idx % n;
                    if (r > c) return;
                    double acc = H[r * n + c] + (r == c ? a.regularization : 0.0);
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += AB[m * n + r] * PAB[m * n + c];
                    H[r * n + c] = acc;
                    H[c * n + r] = acc;
                } else {
                    const int c = idx - n * n;
                    double acc = h[c];
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += AB[m * n + c] * t[m];
                    h[c] = acc;
                }
            });
        }
        RiccatiMark(ex, 3);  // H (for the untiled path: both products)
        if constexpr (dma) {
            if constexpr (deferCopies) {
                if (k > 0) {
                    ex.SetCopyWaves(true);
                    dmaAfterPab(k - 1, P);
                    dmaAfterH(k - 1);
                    ex.SetCopyWaves(false);
                }
            } else {
                if (k > 0) dmaAfterH(k - 1);
            }
        }
        if constexpr (RiccatiFactorBlockedOnMatrixCores<Exec, NX, NU, NE>()) {
            // four-wavefront device kernels, 16..36 inputs: the same L D L^T in 4 x 4 blocks, trailing updates on the matrix cores (NU - 1 phases)
            if (ex.template FactorGainsBlocked<NX, NU>(H, h, K, piv, gains + static_cast<long long>(k) * nu * nk)) failed = failed ? failed : k + 1;
        } else if constexpr (NU > 0 && NU <= 12 && NE == 0) {
            // Input dimension <= 12 fixed at compile time: every lane factorises R = H_uu = L D L^T itself, in registers (NU^3 / 6
            // multiply-adds from NU (NU + 1) / 2 LDS reads), and goes straight on to its right-hand side of
            // [K | kff] = -R^-1 [H_ux | h_u] -- one phase instead of NU + 1 (a barrier and an LDS round trip per Cholesky column).
            ex.ForEach(nk, [&](int c) {
                // R = L D L^T with unit L (no square roots: a column's dependent chain is its inner product and ONE division; the Cholesky form
                // had a square root in front of every division -- 12 of them in sequence for the full-body block)
                double L[NU][NU], U[NU][NU], inv[NU];  // U = L D (unscaled columns), inv[j] = 1 / d_j
                bool bad = false;
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    double d = H[(nx + j) * n + nx + j];
#pragma unroll
                    for (int m = 0; m < j; ++m) d -= U[j][m] * L[j][m];
                    const bool neg = !(d > 0.0);
                    bad = bad || neg;
                    inv[j] = 1.0 / (neg ? 1.0 : d);
#pragma unroll
                    for (int i = j + 1; i < NU; ++i) {
                        double sv = H[(nx + i) * n + nx + j];
#pragma unroll
                        for (int m = 0; m < j; ++m) sv -= U[i][m] * L[j][m];
                        U[i][j] = sv;
                        L[i][j] = sv * inv[j];
                    }
                }
                if (c == 0 && bad) failed = failed ? failed : k + 1;
                double y[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) {  // L y = rhs
                    double sv = c < nx ? -H[(nx + i) * n + c] : -h[nx + i];
#pragma unroll
                    for (int m = 0; m < i; ++m) sv -= L[i][m] * y[m];
                    y[i] = sv;
                }
#pragma unroll
                for (int i = NU - 1; i >= 0; --i) {  // L^T x = D^-1 y
                    double sv = y[i] * inv[i];
#pragma unroll
                    for (int m = i + 1; m < NU; ++m) sv -= L[m][i] * y[m];
                    y[i] = sv;
                    K[i * nk + c] = sv;
                    gains[static_cast<long long>(k) * nu * nk + i * nk + c] = sv;
                }
            });
        } else if constexpr (NU > 12 && NU % UNGAR_RICCATI_BLOCK == 0 && NU <= 4 * UNGAR_RICCATI_BLOCK && NE == 0 && NX >= UNGAR_RICCATI_BLOCKED_MIN_NX) {
            // Input dimension 24 next to a large state (the reference's quadruped with the previous foot positions carried, 25 + 24): BLOCKED Cholesky with NB x NB diagonal blocks that
            // every lane factorises itself, in registers, exactly as above -- 3 NU / NB - 1 phases of register-level work instead of the
            // 2 NU + 1 barrier-separated rank-one phases of the branch below (49 for NU = 24, two thirds of that recursion's time):
            //   A_b  one lane per column of [R_(b, later) | G_b]:  L_bb (registers) from the block's lower triangle,  Y = L_bb^-1 column
            //        (the panel L_(i,b)^T written over R[i][b], i later; the forward-substituted right-hand side over G_b)
            //   B_b  R[i][k] -= Y_i . Y_k (later i >= k),  G[i][:] -= Y_i . Y_G[:]          (Schur complement of the block)
            //   C_b  (blocks in reverse)  one lane per right-hand side:  x_b = L_bb^-T (y_b - sum_(later i) L_(i,b)^T x_i)
            // R = H_uu is read and updated in its LOWER triangle (row i holds R[i][0..i]); G = -[H_ux | h_u] lives in K.
            constexpr int NB = UNGAR_RICCATI_BLOCK, blocks = NU / NB;
            ex.ForEach(nu * nk, [&](int idx) {
                const int i = idx / nk, c = idx - i * nk;
                K[idx] = c < nx ? -H[(nx + i) * n + c] : -h[nx + i];
            });
            auto factorBlock = [&](int b0, double (&L)[NB][NB], double (&inv)[NB]) {  // L L^T = R[b0.., b0..] (lower triangle read); true if not positive definite
                bool bad = false;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double d = H[(nx + b0 + j) * n + nx + b0 + j];
#pragma unroll
                    for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
                    const bool neg = !(d > 0.0);
                    bad = bad || neg;
                    L[j][j] = sqrt(neg ? 1.0 : d);
                    inv[j] = 1.0 / L[j][j];
#pragma unroll
                    for (int i = j + 1; i < NB; ++i) {
                        double sv = H[(nx + b0 + i) * n + nx + b0 + j];
#pragma unroll
                        for (int m = 0; m < j; ++m) sv -= L[i][m] * L[j][m];
                        L[i][j] = sv * inv[j];
                    }
                }
                return bad;
            };
            for (int b = 0; b < blocks; ++b) {
                const int b0 = b * NB, later = nu - b0 - NB;
                ex.ForEach(later + nk, [&](int c) {  // A_b: columns R[b-rows][b0 + NB + c] (c < later), then the right-hand sides
                    double L[NB][NB], inv[NB], y[NB];
                    const bool bad = factorBlock(b0, L, inv);
                    if (c == 0 && bad) failed = failed ? failed : k + 1;
                    double* col = c < later ? H + (nx + b0 + NB + c) * n + nx + b0 : nullptr;  // row (b0 + NB + c) of R, entries of block b: the column by symmetry
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        double sv = col ? col[i] : K[(b0 + i) * nk + (c - later)];
#pragma unroll
                        for (int m = 0; m < i; ++m) sv -= L[i][m] * y[m];
                        y[i] = sv * inv[i];
                    }
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        if (col) col[i] = y[i];
                        else K[(b0 + i) * nk + (c - later)] = y[i];
                    }
                });
                if (later > 0) {
                    const int pairs = later * (later + 1) / 2;
                    ex.ForEach(pairs + later * nk, [&](int idx) {  // B_b
                        if (idx < pairs) {
                            int i = 0, rest = idx;  // (i, k2), k2 <= i, of the later x later lower triangle
                            while (rest > i) {
                                rest -= i + 1;
                                ++i;
                            }
                            const double* yi = H + (nx + b0 + NB + i) * n + nx + b0;
                            const double* yk = H + (nx + b0 + NB + rest) * n + nx + b0;
                            double acc = 0.0;
#pragma unroll
                            for (int m = 0; m < NB; ++m) acc += yi[m] * yk[m];
                            H[(nx + b0 + NB + i) * n + nx + b0 + NB + rest] -= acc;
                        } else {
                            const int e = idx - pairs, i = e / nk, c = e - i * nk;
                            const double* yi = H + (nx + b0 + NB + i) * n + nx + b0;
                            double acc = 0.0;
#pragma unroll
                            for (int m = 0; m < NB; ++m) acc += yi[m] * K[(b0 + m) * nk + c];
                            K[(b0 + NB + i) * nk + c] -= acc;
                        }
                    });
                }
            }
            for (int b = blocks - 1; b >= 0; --b) {
                const int b0 = b * NB, later = nu - b0 - NB;
                ex.ForEach(nk, [&](int c) {  // C_b
                    double L[NB][NB], inv[NB], y[NB];
                    (void)factorBlock(b0, L, inv);
#pragma unroll
                    for (int m = 0; m < NB; ++m) y[m] = K[(b0 + m) * nk + c];
                    for (int i = 0; i < later; ++i) {
                        const double xi = K[(b0 + NB + i) * nk + c];
                        const double* yi = H + (nx + b0 + NB + i) * n + nx + b0;
#pragma unroll
                        for (int m = 0; m < NB; ++m) y[m] -= yi[m] * xi;
                    }
#pragma unroll
                    for (int i = NB - 1; i >= 0; --i) {  // L^T x = y
                        double sv = y[i];
#pragma unroll
                        for (int m = i + 1; m < NB; ++m) sv -= L[m][i] * y[m];
                        sv *= inv[i];
                        y[i] = sv;
                        K[(b0 + i) * nk + c] = sv;
                        gains[static_cast<long long>(k) * nu * nk + (b0 + i) * nk + c] = sv;
                    }
                });
            }
        } else {
            // Larger (or run-time) input dimension: R = H_uu is factorised as L D L^T by RIGHT-LOOKING elimination, applied at the
            // same time to the right-hand sides [K | kff] = -[H_ux | h_u] (forward substitution), then the back substitution is
            // eliminated column by column as well.  Every phase is a flat set of independent rank-one updates over all lanes:
            //   forward, column j:   d_j = R[j][j];   R[i][k] -= R[i][j] R[k][j] / d_j  (j < k <= i);   K[i][:] -= R[i][j] / d_j K[j][:]  (i > j)
            //   scale:               K[i][:] /= d_i
            //   backward, column j:  K[i][:] -= R[j][i] / d_i K[j][:]  (i < j)
            // (2 nu + 1 phases; the left-looking column-by-column Cholesky + one right-hand side per lane it replaces ran nu-long
            // dependent chains of LDS round trips on at most nu lanes: 187 k cycles per knot for the 13 + 24 block of the reference's
            // quadruped OCP.)  The strict lower triangle of R holds the UNSCALED columns L[i][j] d_j afterwards.
            // The reciprocal of pivot j is produced one phase EARLY, by the lane that finishes the diagonal entry (j, j) in phase j - 1
            // (column 0: with the right-hand sides below): a phase then starts from an LDS read instead of every lane's own
            // double-precision division.  A pivot that is not positive is replaced by 1 and reported; every participating lane
            // sees it (they all read d_j), so the first lane -- which writes the status -- does.
            // With equality rows the block is the stage KKT matrix [R D^T; D 0] of order nuE = nu + ne, eliminated in the same order
            // (quasi-definite: positive pivots, then negative ones; exactly-zero rows skipped): rows nu.. of K then hold the multiplier
            // feedback, which enters the cost-to-go below and is not stored.
            ex.ForEach(nuE * nk + 1, [&](int idx) {
                if (idx < nuE * nk) {
                    const int i = idx / nk, c = idx % nk;
                    K[idx] = c < nx ? -rowOf(i)[c] : -rhsOf(i);
                } else {
                    piv[0] = pivotReciprocal(rowOf(0)[nx], 0);
                }
            });
            auto forwardColumn = [&](int j) {
                const int rest = nuE - 1 - j;
                const float restInv = rest > 0 ? 1.0f / static_cast<float>(rest) : 0.0f;
                ex.ForEach(rest * rest + rest * nk + 1, [&](int idx) {
                    if (pivotBad(rowOf(j)[nx + j], j)) failed = failed ? failed : k + 1;
                    const double rd = piv[j];
                    if (idx < rest * rest) {  // trailing block, entries (i, k2) with j < k2 <= i (the square index space is cheaper to decode than the triangle)
                        // quotient by the run-time `rest` through a float reciprocal: exact here (idx + 0.5 is never a multiple of rest, the
                        // operands are far below 2^20), and an integer division costs ~30 instructions per item
                        const int qi = static_cast<int>((static_cast<float>(idx) + 0.5f) * restInv);
                        const int i = j + 1 + qi, k2 = j + 1 + (idx - qi * rest);
                        if (k2 <= i) {
                            double* ri = rowOf(i);
                            const double v = ri[nx + k2] - ri[nx + j] * rowOf(k2)[nx + j] * rd;
                            ri[nx + k2] = v;
                            if (i == j + 1) piv[j + 1] = pivotReciprocal(v, j + 1);  // (k2 == i == j + 1: the next pivot is final)
                        }
                    } else if (idx < rest * rest + rest * nk) {
                        const int e = idx - rest * rest, i = j + 1 + e / nk, c = e % nk;
                        K[i * nk + c] -= rowOf(i)[nx + j] * rd * K[j * nk + c];
                    }
                });
            };
            auto backwardColumn = [&](int j) {
                ex.ForEach(j * nk, [&](int idx) {
                    const int i = idx / nk, c = idx % nk;
                    K[i * nk + c] -= rowOf(j)[nx + i] * piv[i] * K[j * nk + c];
                });
            };
            // (not unrolled: 47 unrolled phase bodies cost the third wavefront per SIMD.  Also measured and dropped for nu = 24: factorising
            // in 24 phases and running both substitutions of a right-hand side in the registers of one lane -- 2 x 276 multiply-adds behind
            // column-wise LDS round trips on 14 lanes, 175 registers: QP step 5.25 -> 6.94 ms; and eliminating an identity along with the
            // right-hand sides, so that L^-1 turns the back substitution into ONE product -- nu + 3 phases instead of 2 nu + 1, but heavier
            // ones and a 23-term chain per entry of the product: 5.16 -> 5.75 ms.)
            for (int j = 0; j < nuE; ++j) forwardColumn(j);
            ex.ForEach(nuE * nk, [&](int idx) { K[idx] *= piv[idx / nk]; });
            for (int j = nuE - 1; j > 0; --j) backwardColumn(j);
            ex.ForEachNoSync(nu * nk, [&](int idx) { gains[static_cast<long long>(k) * nu * nk + idx] = K[idx]; });
        }
        RiccatiMark(ex, 4);  // factorisation of R and gains
        // P <- H_xx + H_ux^T K (symmetrised),  p <- h_x + H_ux^T kff   (into the other buffer, then the buffers swap roles)
        auto costToGoEntry = [&](int i, int j) {  // the symmetrised entry (i, j): the same bits for (j, i)
            double s1 = H[i * n + j], s2 = H[j * n + i];
#pragma unroll 8
            for (int m = 0; m < nu; ++m) {
                s1 += H[(nx + m) * n + i] * K[m * nk + j];
                s2 += H[(nx + m) * n + j] * K[m * nk + i];
            }
            for (int m = nu; m < nuE; ++m) {  // C^T Lambda: the multiplier feedback of the stage equality rows
                const double* row = Eq + (m - nu) * ldq;
                s1 += row[i] * K[m * nk + j];
                s2 += row[j] * K[m * nk + i];
            }
            return 0.5 * (s1 + s2);
        };
        auto costToGoVector = [&](int i) {
            double sv = h[i];
#pragma unroll 8
            for (int m = 0; m < nu; ++m) sv += H[(nx + m) * n + i] * K[m * nk + nx];
            for (int m = nu; m < nuE; ++m) sv += Eq[(m - nu) * ldq + i] * K[m * nk + nx];
            pn[i] = sv;
        };
        if constexpr (RiccatiMatrixCoresFrom<Exec>() > 0 && NX >= RiccatiMatrixCoresFrom<Exec>() && NE == 0 && Exec::kLanes > 64) {
            // on the matrix cores as well (four-wavefront kernels; for the one-wavefront kernels the extra accumulators cost more than the phase saves: measured): T = H_ux^T [K | kff] and its transpose tile by tile, P = H_xx + (T + T^T) / 2, p = h_x + T[:, nx]
            ex.template ProductCostToGo<NX, NU>(H, h, K, Pn, pn);
        } else if constexpr (NX > 0) {
            // compile-time size: one item per PAIR i <= j (each entry computes both orientations anyway), found through the folded
            // rectangle with one division -- half the items of the per-entry loop below
            constexpr int pairs = ((NX + 1) / 2) * (NX + 1);
            ex.ForEach(pairs + nx, [&](int idx) {
                if (idx < pairs) {
                    const int src = RiccatiFoldedSource(nx, idx);
                    if (src < 0) return;
                    const int i = src / nx, j = src - i * nx;
                    const double v = costToGoEntry(i, j);
                    Pn[i * nx + j] = v;
                    Pn[j * nx + i] = v;
                } else {
                    costToGoVector(idx - pairs);
                }
            });
        } else {
            ex.ForEach(nx * nx + nx, [&](int idx) {
                if (idx < nx * nx) Pn[idx] = costToGoEntry(idx / nx, idx % nx);
                else costToGoVector(idx - nx * nx);
            });
        }
        {
            double* swapP = P;
            P = Pn;
            Pn = swapP;
            double* swapp = p;
            p = pn;
            pn = swapp;
        }
        RiccatiMark(ex, 5);  // cost-to-go update
        if constexpr (ahead) {
            if (k > 0) commitKnot();  // AB, H, h, b of knot k are dead from here on
        }
    }
    RiccatiMark(ex, 6);

    // forward pass
    ex.GlobalSync();  // the gains written above are read back below
    ex.ForEach(nx, [&](int i) {
        dx[i] = a.dx0.at(inst, 0, i);
        a.dX.at(inst, 0, i) = dx[i];
    });
    auto forwardKnot = [&](int k, const double* ABk, const double* bkk, const double* gk) {  // gk: [K | kff] of the knot (scratch copy or global)
        if constexpr (NX > 0 && RiccatiSplitDotParts<Exec, NX, NU>() > 1) {
            // Policies with SplitDots (the device kernels): a row's inner product is shared by 2 or 4 adjacent lanes and reduced across them -- the two
            // phases of a knot are dependent multiply-add CHAINS of nx and nx + nu terms on a handful of lanes, and what a phase costs is the length of
            // its chain (8.0 k of 41 k cycles per knot for 37 + 12).
            constexpr int PARTS = RiccatiSplitDotParts<Exec, NX, NU>();
            ex.template SplitDots<PARTS>(
                nu,
                [&](int i, int part) {
                    const double* g = gk + i * nk;
                    double s = 0.0;
#pragma unroll 4
                    for (int m = part; m < nx; m += PARTS) s += g[m] * dx[m];
                    return s;
                },
                [&](int i, double s) {
                    s += gk[i * nk + nx];
                    du[i] = s;
                    a.dU.at(inst, k, i) = s;
                });
            ex.template SplitDots<PARTS>(
                nx,
                [&](int i, int part) {
                    double s = 0.0;
#pragma unroll 4
                    for (int m = part; m < nx; m += PARTS) s += ABk[i * n + m] * dx[m];
#pragma unroll 4
                    for (int m = part; m < nu; m += PARTS) s += ABk[i * n + nx + m] * du[m];
                    return s;
                },
                [&](int i, double s) {
                    s += bkk[i];
                    dxn[i] = s;
                    a.dX.at(inst, k + 1, i) = s;
                });
            double* swapx = dx;
            dx = dxn;
            dxn = swapx;
            return;
        }
        ex.ForEach(nu, [&](int i) {
            const double* g = gk + i * nk;
            double s = g[nx];
#pragma unroll 8
            for (int m = 0; m < nx; ++m) s += g[m] * dx[m];
            du[i] = s;
            a.dU.at(inst, k, i) = s;
        });
        ex.ForEach(nx, [&](int i) {
            double s = bkk[i];
#pragma unroll 8
            for (int m = 0; m < nx; ++m) s += ABk[i * n + m] * dx[m];
#pragma unroll 8
            for (int m = 0; m < nu; ++m) s += ABk[i * n + nx + m] * du[m];
            dxn[i] = s;
            a.dX.at(inst, k + 1, i) = s;
        });
        double* swapx = dx;
        dx = dxn;
        dxn = swapx;
    };
    // Asynchronous copies: the two phases of a knot occupy the first lanes only (nu, nx <= 64), so the OTHER lane groups own the
    // copies -- knot j belongs to owner j % kDmaOwners and lands in buffer j % kDmaOwners of the (now dead) matrix scratch, requested
    // kDmaOwners knots ahead.  (The computing lanes issue no copies: their loads of the gains return in order behind older copies.)
    constexpr bool dmaForward = dma && NX <= 64 && NU <= 64 && Exec::kLanes > 64;  // needs lane groups beyond the computing one
    // (single lane group, below) three buffers in the dead matrix scratch and a wait counter of six bits bound the sizes it serves
    constexpr int kSelfDepth = 3, selfKnotDoubles = NX * (NX + NU) + NX + NU * (NX + 1);
    constexpr int selfKnotCopies = (NX * (NX + NU) + 31) / 32 + (NX + 31) / 32 + (NU * (NX + 1) + 31) / 32;  // copy instructions of one knot
    constexpr bool selfForward = !dmaForward && NX > 0 && NX <= 64 && NU <= 64 && RiccatiExecHasSelfDma<Exec>() && (kSelfDepth - 1) * selfKnotCopies <= 63 &&
                                 kSelfDepth * selfKnotDoubles <= 2 * NX * (NX + NU) + (NX + NU) * (NX + NU) + 2 * NX * NX;
    if constexpr (dmaForward) {
        // per knot: [A|B], b and the gains (read back from global memory otherwise: one exposed round trip per knot)
        constexpr int D = Exec::kDmaOwners, S = NX * (NX + NU) + NX + NU * (NX + 1);
        static_assert(D * S <= 2 * NX * (NX + NU) + (NX + NU) * (NX + NU) + 2 * NX * NX, "forward-pass buffers exceed the matrix scratch");
        auto request = [&](int k) {
            double* buf = scratch + (k % D) * S;
            ex.DmaFetchOne(k, nx * n, [&](int idx) { return &a.jac.at(inst, k, idx); }, buf);
            ex.DmaFetchOne(k, nx, [&](int i) { return &a.b.at(inst, k, i); }, buf + nx * n);
            ex.DmaFetchOne(k, nu * nk, [&](int i) { return gains + static_cast<long long>(k) * nu * nk + i; }, buf + nx * n + nx);
        };
        for (int k = 0; k < D && k < N; ++k) request(k);
        for (int k = 0; k < N; ++k) {
            ex.DmaWaitOne(k);
            ex.Barrier();
            const double* buf = scratch + (k % D) * S;
            forwardKnot(k, buf, buf + nx * n, buf + nx * n + nx);
            if (k + D < N) request(k + D);  // behind the closing barrier of the knot: its buffer is free
        }
    } else if constexpr (selfForward) {
        // One lane group per instance: the same three-deep pipeline, issued and awaited by the computing lanes themselves.  Copies
        // land in order, so "all but the copies of the two younger knots" is a COUNTED wait (the stores of dX / dU issued in between
        // complete in any order and only make the wait conservative); the last knots wait for everything.  The gains come through
        // the scratch memory too: read back from global memory they were one exposed round trip per knot (28 % of the 13 + 4 recursion).
        constexpr int D = kSelfDepth, S = selfKnotDoubles, perKnot = selfKnotCopies;
        auto request = [&](int k) {
            double* buf = scratch + (k % D) * S;
            ex.DmaFetchSelf(nx * n, [&](int idx) { return &a.jac.at(inst, k, idx); }, buf);
            ex.DmaFetchSelf(nx, [&](int i) { return &a.b.at(inst, k, i); }, buf + nx * n);
            ex.DmaFetchSelf(nu * nk, [&](int i) { return gains + static_cast<long long>(k) * nu * nk + i; }, buf + nx * n + nx);
        };
        for (int k = 0; k < D && k < N; ++k) request(k);
        for (int k = 0; k < N; ++k) {
            if (k + D - 1 < N) ex.template DmaWaitSelf<(D - 1) * perKnot>();
            else ex.template DmaWaitSelf<0>();
            ex.Barrier();
            const double* buf = scratch + (k % D) * S;
            forwardKnot(k, buf, buf + nx * n, buf + nx * n + nx);
            if (k + D < N) request(k + D);
        }
    } else {
        // (Staging the gains in registers as well was measured and dropped: 0.88 -> 0.90 ms for the quadrotor QP step.)
        auto fetchForward = [&](int k) {
            ex.Fetch(nx * n, [&](int idx) { return a.jac.at(inst, k, idx); }, sAB);
            ex.Fetch(nx, [&](int i) { return a.b.at(inst, k, i); }, sb);
        };
        auto commitForward = [&] {
            ex.Commit(nx * n, sAB, AB);
            ex.Commit(nx, sb, bk);
            ex.Barrier();
        };
        if constexpr (Exec::kPrefetch) {
            fetchForward(0);
            commitForward();
        }
        for (int k = 0; k < N; ++k) {
            if constexpr (Exec::kPrefetch) {
                if (k + 1 < N) fetchForward(k + 1);
            } else {
                ex.ForEach(nx * n + nx, [&](int idx) {
                    if (idx < nx * n) AB[idx] = a.jac.at(inst, k, idx);
                    else bk[idx - nx * n] = a.b.at(inst, k, idx - nx * n);
                });
            }
            forwardKnot(k, AB, bk, gains + static_cast<long long>(k) * nu * nk);
            if constexpr (Exec::kPrefetch) {
                if (k + 1 < N) commitForward();
            }
        }
    }
    RiccatiMark(ex, 7);  // forward pass
    if (a.status) ex.ForEach(1, [&](int) { a.status[inst] = failed; });
}

}  // namespace ungar_amd::kernels
