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
// One 64-lane workgroup owns one MPC instance: every matrix of the recursion lives in LDS, the lanes share the entries of
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
//   kPrefetch                whether the policy stages a knot's operands in registers (below); without it they are read in place
//   kAhead                   with kPrefetch: stage the NEXT knot's operands while this one is processed (else: this knot's, at its top)
//   Stage<SLOTS>             per-lane registers for a strided global read of up to 64 * SLOTS doubles
//   Fetch(n, f, stage)       issues the loads stage <- f(i); no barrier, nothing waits for the data
//   Commit(n, stage, dst)    dst[i] <- stage (no barrier);  Barrier() synchronises the scratch memory
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
};

/// Doubles of LDS (or host scratch) one instance needs.
UNGAR_HD inline int RiccatiScratchDoubles(int nx, int nu) {
    const int n = nx + nu;
    return nx * n /*AB*/ + n * n /*H*/ + 2 * nx * nx /*P, Pn*/ + nx * n /*PAB*/ + 2 * nx /*p, pn*/ + nx /*t*/ + n /*h*/ + nx /*bk*/ + nu * (nx + 1) /*K|kff*/ +
           nx + nu /*dx, du*/ + nx /*dxn*/ + nu /*Cholesky pivots*/;
}

/// The whole recursion for instance `inst`; `scratch` holds RiccatiScratchDoubles(nx, nu) doubles private to the workgroup.
/// NX / NU > 0 fix the sizes at compile time (the index arithmetic -- a division and a remainder by nx + nu per matrix entry --
/// and the short inner products then cost a fraction of the generic code, which is issue-bound on them); 0 = read them from `a`.
template <class Exec, int NX = 0, int NU = 0>
UNGAR_HD void RiccatiInstance(const RiccatiArgs& a, long long inst, double* scratch, Exec& ex) {
    const int nx = NX > 0 ? NX : a.nx, nu = NU > 0 ? NU : a.nu, n = nx + nu, N = a.N, nk = nx + 1;
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
    double* dx = K + nu * nk;
    double* du = dx + nx;
    double* dxn = du + nu;
    double* piv = dxn + nx;  // diagonal of the Cholesky factor of R
    double* gains = a.gains + inst * static_cast<long long>(N) * nu * nk;
    int failed = 0;

    // terminal cost-to-go
    ex.ForEach(nx * nx, [&](int idx) {
        const int i = idx / nx, j = idx % nx;
        double v = 0.0;
        if (a.hessN.base) v = i <= j ? a.hessN.at(inst, 0, i * nx + j) : a.hessN.at(inst, 0, j * nx + i);
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
    for (int k = N - 1; k >= 0; --k) {
        if constexpr (ahead) {
            if (k > 0) fetchKnot(k - 1);  // in flight while knot k is processed
        } else if constexpr (Exec::kPrefetch) {
            fetchKnot(k);
            commitKnot();
        } else {
            // one loop per operand, ONE barrier: inside a loop every iteration is `LDS[i] = global[i]`, which the compiler unrolls into
            // a batch of loads followed by the stores; a single loop over all four operands with an if-chain serialised them
            // (17 dependent round trips per lane for the 37 + 12 block)
            ex.ForEachNoSync(nx * n, [&](int idx) { AB[idx] = a.jac.at(inst, k, idx); });
            ex.ForEachNoSync(n * n, [&](int idx) { H[idx] = a.hess.at(inst, k, idx); });
            ex.ForEachNoSync(n, [&](int idx) { h[idx] = a.grad.at(inst, k, idx); });
            ex.ForEachNoSync(nx, [&](int idx) { bk[idx] = a.b.at(inst, k, idx); });
            ex.Barrier();
        }
        if constexpr (NX >= UNGAR_RICCATI_TILE_MIN_NX) {
            // Large blocks, sizes fixed at compile time: 2 x 4 register tiles -- eight multiply-adds per six LDS reads instead of per
            // sixteen, and no bounds checks inside the product (a tile on the edge reads past its row / matrix into the neighbouring
            // scratch arrays, which is harmless: only the stores are guarded).  A tile's columns are INTERLEAVED (tc, tc + tilesC, ...):
            // neighbouring lanes then read neighbouring LDS words.  With four contiguous columns per lane the lanes of a read were 32
            // bytes apart and, paired into ds_read2_b64 (32-bank mode), collided four ways: SQ_LDS_BANK_CONFLICT was 4x the LDS issue cycles.
            constexpr int TI = 2, TC = NX >= 24 ? 4 : 2, tilesI = (NX + TI - 1) / TI, tilesC = (NX + NU + TC - 1) / TC;
            ex.ForEach(tilesI * tilesC + nx, [&](int idx) {
                if (idx < tilesI * tilesC) {
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
                    const int i = idx - tilesI * tilesC;
                    double acc = p[i];
#pragma unroll 8
                    for (int m = 0; m < nx; ++m) acc += P[i * nx + m] * bk[m];
                    t[i] = acc;
                }
            });
            // H = W + AB^T PAB: all tiles (interleaved rows and columns straddle the diagonal); entries r <= c are updated and mirrored
            constexpr int tilesR = (NX + NU + TI - 1) / TI;
            ex.ForEach(tilesR * tilesC + n, [&](int idx) {
                if (idx < tilesR * tilesC) {
                    const int r0 = idx / tilesC, c0 = idx % tilesC;
                    double acc[TI][TC] = {};
#pragma unroll 4
                    for (int m = 0; m < nx; ++m) {
                        double lv[TI], qv[TC];
                        for (int a2 = 0; a2 < TI; ++a2) lv[a2] = AB[m * n + r0 + a2 * tilesR];
                        for (int b2 = 0; b2 < TC; ++b2) qv[b2] = PAB[m * n + c0 + b2 * tilesC];
                        for (int a2 = 0; a2 < TI; ++a2)
                            for (int b2 = 0; b2 < TC; ++b2) acc[a2][b2] += lv[a2] * qv[b2];
                    }
                    for (int a2 = 0; a2 < TI; ++a2)
                        for (int b2 = 0; b2 < TC; ++b2) {
                            const int r = r0 + a2 * tilesR, c = c0 + b2 * tilesC;
                            if (r <= c && c < n) {
                                const double e = H[r * n + c] + (r == c ? a.regularization : 0.0) + acc[a2][b2];
                                H[r * n + c] = e;
                                H[c * n + r] = e;
                            }
                        }
                } else {
                    const int c = idx - tilesR * tilesC;
                    double acc = h[c];
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
        if constexpr (NU > 0 && NU <= 12) {
            // Input dimension <= 12 fixed at compile time: every lane factorises R = H_uu = L L^T itself, in registers (NU^3 / 6
            // multiply-adds from NU (NU + 1) / 2 LDS reads), and goes straight on to its right-hand side of
            // [K | kff] = -R^-1 [H_ux | h_u] -- one phase instead of NU + 1 (a barrier and an LDS round trip per Cholesky column).
            ex.ForEach(nk, [&](int c) {
                double L[NU][NU];
                bool bad = false;
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    double d = H[(nx + j) * n + nx + j];
#pragma unroll
                    for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
                    const bool neg = !(d > 0.0);
                    bad = bad || neg;
                    L[j][j] = sqrt(neg ? 1.0 : d);
#pragma unroll
                    for (int i = j + 1; i < NU; ++i) {
                        double sv = H[(nx + i) * n + nx + j];
#pragma unroll
                        for (int m = 0; m < j; ++m) sv -= L[i][m] * L[j][m];
                        L[i][j] = sv / L[j][j];
                    }
                }
                if (c == 0 && bad) failed = failed ? failed : k + 1;
                double y[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) {  // L y = rhs
                    double sv = c < nx ? -H[(nx + i) * n + c] : -h[nx + i];
#pragma unroll
                    for (int m = 0; m < i; ++m) sv -= L[i][m] * y[m];
                    y[i] = sv / L[i][i];
                }
#pragma unroll
                for (int i = NU - 1; i >= 0; --i) {  // L^T x = y
                    double sv = y[i];
#pragma unroll
                    for (int m = i + 1; m < NU; ++m) sv -= L[m][i] * y[m];
                    sv /= L[i][i];
                    y[i] = sv;
                    K[i * nk + c] = sv;
                    gains[static_cast<long long>(k) * nu * nk + i * nk + c] = sv;
                }
            });
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
            ex.ForEach(nu * nk, [&](int idx) {
                const int i = idx / nk, c = idx % nk;
                K[idx] = c < nx ? -H[(nx + i) * n + c] : -h[nx + i];
            });
            // The reciprocal of a pivot is computed once (piv) and multiplied with afterwards.
            auto forwardColumn = [&](int j) {
                const int rest = nu - 1 - j;
                const float restInv = rest > 0 ? 1.0f / static_cast<float>(rest) : 0.0f;
                ex.ForEach(rest * rest + rest * nk + 1, [&](int idx) {
                    const double dj = H[(nx + j) * n + nx + j];
                    const bool bad = !(dj > 0.0);
                    const double rd = 1.0 / (bad ? 1.0 : dj);
                    if (idx < rest * rest) {  // trailing block, entries (i, k2) with j < k2 <= i (the square index space is cheaper to decode than the triangle)
                        // quotient by the run-time `rest` through a float reciprocal: exact here (idx + 0.5 is never a multiple of rest, the
                        // operands are far below 2^20), and an integer division costs ~30 instructions per item
                        const int qi = static_cast<int>((static_cast<float>(idx) + 0.5f) * restInv);
                        const int i = j + 1 + qi, k2 = j + 1 + (idx - qi * rest);
                        if (k2 <= i) H[(nx + i) * n + nx + k2] -= H[(nx + i) * n + nx + j] * H[(nx + k2) * n + nx + j] * rd;
                    } else if (idx < rest * rest + rest * nk) {
                        const int e = idx - rest * rest, i = j + 1 + e / nk, c = e % nk;
                        K[i * nk + c] -= H[(nx + i) * n + nx + j] * rd * K[j * nk + c];
                    } else {
                        piv[j] = rd;
                        if (bad) failed = failed ? failed : k + 1;
                    }
                });
            };
            auto backwardColumn = [&](int j) {
                ex.ForEach(j * nk, [&](int idx) {
                    const int i = idx / nk, c = idx % nk;
                    K[i * nk + c] -= H[(nx + j) * n + nx + i] * piv[i] * K[j * nk + c];
                });
            };
            for (int j = 0; j < nu; ++j) forwardColumn(j);  // (not unrolled: 47 unrolled phase bodies cost the third wavefront per SIMD)
            ex.ForEach(nu * nk, [&](int idx) { K[idx] *= piv[idx / nk]; });
            for (int j = nu - 1; j > 0; --j) backwardColumn(j);
            ex.ForEachNoSync(nu * nk, [&](int idx) { gains[static_cast<long long>(k) * nu * nk + idx] = K[idx]; });
        }
        // P <- H_xx + H_ux^T K (symmetrised),  p <- h_x + H_ux^T kff   (into the other buffer, then the buffers swap roles)
        ex.ForEach(nx * nx + nx, [&](int idx) {
            if (idx < nx * nx) {
                const int i = idx / nx, j = idx % nx;
                double s1 = H[i * n + j], s2 = H[j * n + i];
#pragma unroll 8
                for (int m = 0; m < nu; ++m) {
                    s1 += H[(nx + m) * n + i] * K[m * nk + j];
                    s2 += H[(nx + m) * n + j] * K[m * nk + i];
                }
                Pn[idx] = 0.5 * (s1 + s2);
            } else {
                const int i = idx - nx * nx;
                double sv = h[i];
#pragma unroll 8
                for (int m = 0; m < nu; ++m) sv += H[(nx + m) * n + i] * K[m * nk + nx];
                pn[i] = sv;
            }
        });
        {
            double* swapP = P;
            P = Pn;
            Pn = swapP;
            double* swapp = p;
            p = pn;
            pn = swapp;
        }
        if constexpr (ahead) {
            if (k > 0) commitKnot();  // AB, H, h, b of knot k are dead from here on
        }
    }

    // forward pass
    ex.GlobalSync();  // the gains written above are read back below
    ex.ForEach(nx, [&](int i) {
        dx[i] = a.dx0.at(inst, 0, i);
        a.dX.at(inst, 0, i) = dx[i];
    });
    auto fetchForward = [&](int k) {
        ex.Fetch(nx * n, [&](int idx) { return a.jac.at(inst, k, idx); }, sAB);
        ex.Fetch(nx, [&](int i) { return a.b.at(inst, k, i); }, sb);
    };
    if constexpr (Exec::kPrefetch) {
        fetchForward(0);
        ex.Commit(nx * n, sAB, AB);
        ex.Commit(nx, sb, bk);
        ex.Barrier();
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
        ex.ForEach(nu, [&](int i) {
            const double* g = gains + static_cast<long long>(k) * nu * nk + i * nk;
            double s = g[nx];
#pragma unroll 8
            for (int m = 0; m < nx; ++m) s += g[m] * dx[m];
            du[i] = s;
            a.dU.at(inst, k, i) = s;
        });
        ex.ForEach(nx, [&](int i) {
            double s = bk[i];
#pragma unroll 8
            for (int m = 0; m < nx; ++m) s += AB[i * n + m] * dx[m];
#pragma unroll 8
            for (int m = 0; m < nu; ++m) s += AB[i * n + nx + m] * du[m];
            dxn[i] = s;
            a.dX.at(inst, k + 1, i) = s;
        });
        {
            double* swapx = dx;
            dx = dxn;
            dxn = swapx;
        }
        if constexpr (Exec::kPrefetch) {
            if (k + 1 < N) {
                ex.Commit(nx * n, sAB, AB);
                ex.Commit(nx, sb, bk);
                ex.Barrier();
            }
        }
    }
    if (a.status) ex.ForEach(1, [&](int) { a.status[inst] = failed; });
}

}  // namespace ungar_amd::kernels
