// ungar_amd :: batched SQP kernels for optimal-control structure: the Riccati solve of the
// QP subproblem (ocp_riccati.hpp; replaces the OSQP call of soft_sqp.hpp:143-158 for this structure), the merit terms
// phi = objective + barrier and theta = c |g| of the line search (soft_sqp.hpp:68-87), trial points, and the three-way
// acceptance test of backtracking_line_search.hpp:116-151.  One 64-lane workgroup (= one wavefront) per MPC instance;
// instances are independent, so a launch of thousands of them fills the device.
#include "../runtime/measurement.hpp"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#ifndef UNGAR_RICCATI_MFMA_MIN_NX_ONE_WAVE
#define UNGAR_RICCATI_MFMA_MIN_NX_ONE_WAVE 6
#endif

#include "ocp_sqp.hpp"
#include "ocp_barrier.hpp"

namespace ungar_amd::kernels {

namespace {

constexpr int kBlock = 64;

/// Lanes of the workgroup stride over the index range and meet at a barrier.  SLOTS_AB / SLOTS_W: registers per lane that
/// stage the next knot's [A|B] block and stage Hessian (64 * SLOTS doubles each); 0 = read the operands in place.
/// DMA: operands are copied global -> LDS by the LDS-DMA path (global_load_lds_dword; no registers, nothing waits until DmaWait).
template <int BLOCK, int SLOTS_AB, int SLOTS_W, bool AHEAD = true, bool DMA = false>
struct DeviceExec {
    static constexpr bool kPrefetch = SLOTS_AB > 0;
    static constexpr bool kAhead = AHEAD;
    static constexpr bool kDma = DMA;
    static constexpr int kLanes = BLOCK;
    static constexpr int kWaves = BLOCK / 64, kDmaOwners = kWaves > 1 ? kWaves - 1 : 1;
    static_assert(!DMA || SLOTS_AB == 0, "no register staging next to the asynchronous copies");
    template <int K>
    struct Stage {
        double r[K > 0 ? K : 1];
    };
    using StageAB = Stage<SLOTS_AB>;
    using StageW = Stage<SLOTS_W>;
    using StageV = Stage<1>;

    /// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() also waits for
    /// vmcnt(0), i.e. for every global load in flight -- which would serialise the prefetch of the next knot's operands with the
    /// first phase of the current one (measured: the staged variant was SLOWER than reading in place, 1.03 vs 0.75 ms).
    static __device__ __forceinline__ void LdsBarrier() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    /// Full barrier: global writes of the workgroup (the gains) become visible to its own later reads.
    static __device__ __forceinline__ void GlobalSync() { __syncthreads(); }
    template <class F>
    __device__ __forceinline__ void ForEach(int n, F f) {
        for (int i = static_cast<int>(threadIdx.x); i < n; i += BLOCK) f(i);
        LdsBarrier();
    }
    /// The same loop without the closing barrier: several of them may share one Barrier().
    template <class F>
    __device__ __forceinline__ void ForEachNoSync(int n, F f) {
        for (int i = static_cast<int>(threadIdx.x); i < n; i += BLOCK) f(i);
    }
    template <class F, int K>
    __device__ __forceinline__ void Fetch(int n, F f, Stage<K>& s) {
#pragma unroll
        for (int j = 0; j < K; ++j) {  // fully unrolled: the staging array stays in registers
            const int i = static_cast<int>(threadIdx.x) + j * BLOCK;
            if (i < n) s.r[j] = f(i);
        }
    }
    template <int K>
    __device__ __forceinline__ void Commit(int n, const Stage<K>& s, double* dst) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = static_cast<int>(threadIdx.x) + j * BLOCK;
            if (i < n) dst[i] = s.r[j];
        }
    }
    __device__ __forceinline__ void Barrier() { LdsBarrier(); }
    // ---- inner products shared by PARTS adjacent lanes (forward pass): row r of `rows` by lanes PARTS r .. PARTS r + PARTS - 1, partial sums
    // reduced with DPP quad permutes (register to register), the first lane of the group stores.  Closes with a barrier.
#ifndef UNGAR_RICCATI_NO_SPLIT_DOTS
    static constexpr bool kSplitDots = true;
#else
    static constexpr bool kSplitDots = false;
#endif
    template <int CTRL>
    static __device__ __forceinline__ double QuadPermute(double v) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    }
    template <int PARTS, class Term, class Store>
    __device__ __forceinline__ void SplitDots(int rows, Term term, Store store) {
        static_assert(PARTS == 2 || PARTS == 4);
        const int tid = static_cast<int>(threadIdx.x), r = tid / PARTS, part = tid % PARTS;
        double s = r < rows ? term(r, part) : 0.0;
        s += QuadPermute<0xB1>(s);                            // quad_perm [1, 0, 3, 2]
        if constexpr (PARTS == 4) s += QuadPermute<0x4E>(s);  // quad_perm [2, 3, 0, 1]
        if (r < rows && part == 0) store(r, s);
        LdsBarrier();
    }

    /// dst[i] <- *addr(i), i < n, 32 doubles per instruction: lane l moves dword l & 1 of element 32 c + l / 2 of chunk c (any element
    /// stride, 8-byte alignment); the LDS side of a copy is (wave-uniform base) + 4 l.  Chunks firstChunk, firstChunk + step, ...
    template <class F>
    static __device__ __forceinline__ void DmaIssue(int n, F addr, double* dst, int firstChunk, int step) {
        const int lane = static_cast<int>(threadIdx.x) & 63, chunks = (n + 31) >> 5;
        for (int c = firstChunk; c < chunks; c += step) {
            const int e = (c << 5) + (lane >> 1);
            if (e < n)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(addr(e)) + (lane & 1), (__attribute__((address_space(3))) void*)(dst + (c << 5)), 4, 0, 0);
        }
    }
    static __device__ __forceinline__ int Wave() { return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6); }
    /// true: the copies requested next are issued by the wavefronts other than the first (see ocp_riccati.hpp: deferCopies)
    bool copySpare = false;
#ifndef UNGAR_RICCATI_NO_DEFERRED_COPIES
    __device__ __forceinline__ void SetCopyWaves(bool spareOnly) requires(BLOCK > 64) { copySpare = spareOnly; }
#endif
    template <class F>
    __device__ __forceinline__ void DmaFetch(int n, F addr, double* dst) {
        if (!copySpare) DmaIssue(n, addr, dst, Wave(), kWaves);
        else if (Wave() > 0) DmaIssue(n, addr, dst, Wave() - 1, kWaves - 1);
    }
#ifndef UNGAR_RICCATI_NO_WIDE_COPIES
    /// dst[i] <- src[i], i < n, src contiguous: 128 doubles per instruction (global_load_lds_dwordx4: lane l moves elements 2 l, 2 l + 1 of its chunk;
    /// dst + 128 c must be 16-byte aligned in LDS), an odd last element by the 4-byte form.  The copies cost ISSUE slots of the phase that requests them
    /// (address generation, M0, the instruction: ~10 per copy): 15 -> 4 per wavefront and knot for the 37 x 49 block.
    __device__ __forceinline__ void DmaFetchContiguous(int n, const double* src, double* dst) {
        const int lane = static_cast<int>(threadIdx.x) & 63, pairs = n >> 1, chunks = (pairs + 63) >> 6;
        if (copySpare && Wave() == 0) return;
        const int first = copySpare ? Wave() - 1 : Wave(), step = copySpare ? kWaves - 1 : kWaves;
        for (int c = first; c < chunks; c += step) {
            const int pr = (c << 6) + lane;
            if (pr < pairs) __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(src + 2 * pr), (__attribute__((address_space(3))) void*)(dst + (c << 7)), 16, 0, 0);
        }
        if ((n & 1) && Wave() == kWaves - 1 && lane < 2)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(src + (n - 1)) + lane, (__attribute__((address_space(3))) void*)(dst + (n - 1)), 4, 0, 0);
    }
#endif
    __device__ __forceinline__ void DmaWait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    template <class F>
    __device__ __forceinline__ void DmaFetchOne(int owner, int n, F addr, double* dst) {
        if (Wave() == 1 + owner % kDmaOwners) DmaIssue(n, addr, dst, 0, 1);
    }
    __device__ __forceinline__ void DmaWaitOne(int owner) {
        if (Wave() == 1 + owner % kDmaOwners) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    /// One wavefront per instance: it issues its own copies (they land in issue order) and waits for all but the youngest YOUNGER.
#ifndef UNGAR_RICCATI_NO_SELF_DMA
    static constexpr bool kDmaSelf = BLOCK == 64;
#else
    static constexpr bool kDmaSelf = false;
#endif
    template <class F>
    __device__ __forceinline__ void DmaFetchSelf(int n, F addr, double* dst) {
        DmaIssue(n, addr, dst, 0, 1);
    }
    template <int YOUNGER>
    __device__ __forceinline__ void DmaWaitSelf() {
        static_assert(YOUNGER >= 0 && YOUNGER <= 63);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    }
    // ---- the two large products of a knot on the FP64 matrix cores (four-wavefront kernels) ------------------------------------
    // v_mfma_f64_16x16x4_f64: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[(lane >> 4) + 4 r][lane & 15] in element r.
#ifndef UNGAR_RICCATI_NO_MFMA
    static constexpr bool kMatrixCores = (BLOCK == 256 && DMA) || BLOCK == 64;
#else
    static constexpr bool kMatrixCores = false;
#endif
    // state dimension from which the two large products run on the matrix cores: the one-wavefront kernels share a SIMD with two or three
    // other instances and are bound by VALU / LDS issue there, so even a single padded 16 x 16 tile pays (the matrix pipe runs beside the VALU)
    static constexpr int kMatrixCoresMinNx = BLOCK == 64 ? UNGAR_RICCATI_MFMA_MIN_NX_ONE_WAVE : UNGAR_RICCATI_MFMA_MIN_NX;
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    /// PAB = P [A|B] (NX x n) and t = P b + p.  P is symmetric (bitwise: it is written symmetrised), so A[i][k] = P[k][i] is read along rows.
    /// Wavefront w owns the tile columns w, w + 4, ...; padded rows / columns / k-steps contribute zeros.
    template <int NX, int NU>
    __device__ __forceinline__ void ProductPab(const double* P, const double* AB, const double* bk, const double* p, double* PAB, double* t) {
        constexpr int n = NX + NU, TR = (NX + 15) / 16, TC = (n + 1 + 15) / 16, KS = (NX + 3) / 4;
        const int lane = static_cast<int>(threadIdx.x) & 63, li = lane & 15, lk = lane >> 4;
        for (int tj = Wave(); tj < TC; tj += kWaves) {
            f64x4 acc[TR];
#pragma unroll
            for (int ti = 0; ti < TR; ++ti) acc[ti] = f64x4{0.0, 0.0, 0.0, 0.0};
            const int col = 16 * tj + li;
            // operands are loaded unconditionally from clamped addresses: no branches between the matrix instructions.  A padded ROW of A only
            // feeds the output row of the same index and a padded COLUMN of B only its output column -- neither is stored --, so they need no
            // zeroing (the clamped addresses keep the reads inside initialised arrays: finite numbers); only a padded k-step must contribute
            // nothing, and zeroing ONE operand there (the B word, in the last step alone) does it: 8 selects per step fewer on the critical path.
            const double* bBase = col < n ? AB + col : bk;
            const int bStride = col < n ? n : 1;
            int rowC[TR];
#pragma unroll
            for (int ti = 0; ti < TR; ++ti) rowC[ti] = 16 * ti + li < NX ? 16 * ti + li : NX - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kk = 4 * ks + lk;
                const bool kOk = kk < NX;
                const int kc = kOk ? kk : NX - 1;
                const double braw = bBase[kc * bStride];
                const double bv = kOk ? braw : 0.0;
                double av[TR];
#pragma unroll
                for (int ti = 0; ti < TR; ++ti) av[ti] = P[kc * NX + rowC[ti]];
#pragma unroll
                for (int ti = 0; ti < TR; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], bv, acc[ti], 0, 0, 0);
            }
#pragma unroll
            for (int ti = 0; ti < TR; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + lk + 4 * r;
                    if (row < NX) {
                        if (col < n) PAB[row * n + col] = acc[ti][r];
                        else if (col == n) t[row] = acc[ti][r] + p[row];
                    }
                }
        }
        LdsBarrier();
    }
    /// H = W + [A|B]^T PAB on and above the diagonal (mirrored), h = w + [A|B]^T t.  W: folded upper triangle (RiccatiFoldedIndex) or
    /// null = in place in H; the upper tiles (column-major) are dealt to the wavefronts in contiguous runs.
    template <int NX, int NU, bool FOLDED>
    __device__ __forceinline__ void ProductH(const double* AB, const double* PAB, const double* t, const double* Wfold, const double* wv, double reg, double* H, double* h) {
        constexpr int n = NX + NU, TT = (n + 1 + 15) / 16, tiles = TT * (TT + 1) / 2, CH = (tiles + kWaves - 1) / kWaves, KS = (NX + 3) / 4;
        const int lane = static_cast<int>(threadIdx.x) & 63, li = lane & 15, lk = lane >> 4;
        const int first = Wave() * CH;
        f64x4 acc[CH];
        int ti[CH], tj[CH], rowC[CH], bStride[CH];
        const double* bBase[CH];
        bool live[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
            int idx = first + c, j = 0;  // column-major upper triangle: tile idx -> (i, j), i <= j
            live[c] = idx < tiles;
            if (!live[c]) idx = 0;
            while (idx > j) {
                idx -= j + 1;
                ++j;
            }
            ti[c] = idx;
            tj[c] = j;
            const int row = 16 * idx + li, col = 16 * j + li;
            rowC[c] = row < n ? row : n - 1;
            bBase[c] = col < n ? PAB + col : t;  // (columns beyond n + 1 read t too: their output columns are not stored)
            bStride[c] = col < n ? n : 1;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + lk;
            const bool kOk = kk < NX;
            const int kc = kOk ? kk : NX - 1;
            double av[CH], bv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const double braw = bBase[c][kc * bStride[c]];
                av[c] = AB[kc * n + rowC[c]];  // (padded rows / columns feed outputs that are not stored; a padded k-step is zeroed in B alone: see ProductPab)
                bv[c] = kOk ? braw : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[c], bv[c], acc[c], 0, 0, 0);
        }
#ifdef UNGAR_RICCATI_CLOCKS
        Mark(6);  // (diagnostic builds: the matrix-instruction loop of the H phase under "-", its epilogue and barrier under "H")
#endif
        // epilogue, tile by tile: the four stage-Hessian / gradient words of the lane's entries are loaded first (clamped addresses, no branches),
        // then added and stored under the lane's conditions -- one LDS round trip per tile instead of one per entry
        const double* Wsrc = FOLDED ? Wfold : H;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (!live[c]) continue;  // uniform over the wavefront
            const int col = 16 * tj[c] + li, cc = col < n ? col : n - 1;
            double base[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti[c] + lk + 4 * r, rc = row < n ? row : n - 1;
                const int lo = rc < cc ? rc : cc, hi = rc < cc ? cc : rc;
                const int at = FOLDED ? RiccatiFoldedIndex(n, lo, hi) : lo * n + hi;
                base[r] = col == n ? wv[rc] : Wsrc[at];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti[c] + lk + 4 * r;
                const double e = base[r] + acc[c][r] + ((row == col) ? reg : 0.0);
                if (row < n && col < n && row <= col) {
                    H[row * n + col] = e;
                    H[col * n + row] = e;
                } else if (row < n && col == n) {
                    h[row] = e;
                }
            }
        }
        LdsBarrier();
    }

    /// Cost-to-go update P' = H_xx + H_ux^T K (symmetrised), p' = h_x + H_ux^T kff with K = [K | kff] (NU x (NX + 1)):  for every upper
    /// tile both T = H_ux^T K and its mirror image K^T H_ux are accumulated (the same lane holds T[i][j] and T[j][i]), so that the entry and
    /// its transpose get the same bits.
    template <int NX, int NU>
    __device__ __forceinline__ void ProductCostToGo(const double* H, const double* h, const double* K, double* Pn, double* pn) {
        constexpr int n = NX + NU, nk = NX + 1, TT = (nk + 15) / 16, tiles = TT * (TT + 1) / 2, CH = (tiles + kWaves - 1) / kWaves, KS = (NU + 3) / 4;
        const int lane = static_cast<int>(threadIdx.x) & 63, li = lane & 15, lk = lane >> 4;
        const int first = Wave() * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            int idx = first + c, tj = 0;  // column-major upper triangle: tile idx -> (ti, tj), ti <= tj
            if (idx >= tiles) break;      // uniform over the wavefront
            while (idx > tj) {
                idx -= tj + 1;
                ++tj;
            }
            const int ti = idx, row = 16 * ti + li, col = 16 * tj + li;
            const int rowC = row < NX ? row : NX - 1, colC = col < nk ? col : nk - 1, colX = col < NX ? col : NX - 1;
            f64x4 t1 = f64x4{0.0, 0.0, 0.0, 0.0}, t2 = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int m = 4 * ks + lk;
                const bool mOk = m < NU;
                const int mc = mOk ? m : NU - 1;
                const double* Hrow = H + (NX + mc) * n;  // row m of H_ux
                const double* Krow = K + mc * nk;
                const double a1raw = Hrow[rowC], b1raw = Krow[colC], a2raw = Krow[rowC], b2raw = Hrow[colX];
                const double a1 = a1raw, b1 = mOk ? b1raw : 0.0;  // (padded rows / columns feed outputs that are not stored: see ProductPab)
                const double a2 = a2raw, b2 = mOk ? b2raw : 0.0;
                t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, t1, 0, 0, 0);
                t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, t2, 0, 0, 0);
            }
            double base[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = 16 * ti + lk + 4 * r, rc = rw < NX ? rw : NX - 1;
                base[r] = col == NX ? h[rc] : H[rc * n + colX];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rw = 16 * ti + lk + 4 * r;
                if (rw < NX && col < NX && rw <= col) {
                    const double v = base[r] + 0.5 * (t1[r] + t2[r]);
                    Pn[rw * NX + col] = v;
                    Pn[col * NX + rw] = v;
                } else if (rw < NX && col == NX) {
                    pn[rw] = base[r] + t1[r];
                }
            }
        }
        LdsBarrier();
    }

    // ---- R = H_uu = L D L^T and the gains, BLOCKED, trailing updates on the matrix cores (four-wavefront kernels, NU a multiple of 4) -----------
    // The column-by-column L D L^T of ocp_riccati.hpp is 2 NU + 1 barrier phases, and four workgroups share a CU: the wall-clock of a phase is
    // its critical-path instruction count times ~12 cycles, 40 k cycles per knot for NU = 24.  Same factorisation, 4 x 4 blocks:
    //   P_b  (one phase)  every lane of wavefront 0 factorises the diagonal block itself (registers; redundant work is free, serial
    //        instructions are not); the first lanes, one per ROW from the block down, forward-substitute their four panel entries (unscaled
    //        columns L d, as the column-wise route stores them), the lanes after them, one per right-hand side, the block's rows of K;
    //   T_b  (one phase)  R_22 -= (L_21 d) D^-1 (L_21 d)^T and K_2 -= (L_21 d) D^-1 K_b: one v_mfma_f64_16x16x4_f64 per 16 x 16 tile,
    //        C tile LDS -> accumulator -> LDS (nothing stays in registers across phases);
    //   backwards  B1_b  x_b = D^-1 y_b - L_bb^T-part (one lane per right-hand side; the scaling of the column-wise route folded in),
    //              B2_b  K_above -= (L_b,above d)^T x_b on the matrix cores.
    // 4 NU / 4 - 1 = 23 phases for NU = 24, ~1.5 k critical-path instructions instead of ~3.3 k.  The arithmetic is the L D L^T's (same pivots,
    // IEEE reciprocals; only the association inside a rank-4 update differs), so the accuracy is the column-wise route's (tests).
#ifndef UNGAR_RICCATI_NO_BLOCKED_LDLT
    static constexpr bool kFactorBlocked = BLOCK == 256;
#else
    static constexpr bool kFactorBlocked = false;
#endif
    /// Returns true if a pivot was not positive (replaced by 1).  K (NU x (NX + 1)) and gainsK (global, same layout) receive [K | kff]; piv: NU
    /// doubles (reciprocal pivots); the lower triangle of the H_uu block of H is overwritten by the unscaled columns of L.  Closes with a barrier.
    template <int NX, int NU>
    __device__ __forceinline__ bool FactorGainsBlocked(double* H, const double* h, double* K, double* piv, double* gainsK) {
        constexpr int n = NX + NU, nk = NX + 1, STEPS = NU / 4, TCK = (nk + 15) / 16;
        constexpr int ROWL = (NU + 15) / 16 * 16;  // P_b: rows in lanes 0 .. ROWL - 1, right-hand sides from lane ROWL on
        static_assert(NU % 4 == 0 && NU <= 32 && ROWL + nk <= 64, "two row tiles of trailing rows; rows and right-hand sides share one wavefront");
        const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, li = lane & 15, lk = lane >> 4, w = Wave();
        double* R = H + NX * n + NX;  // R[i][j], j <= i, at R[i * n + j]
        for (int idx = tid; idx < NU * nk; idx += BLOCK) {
            const int i = static_cast<int>((static_cast<float>(idx) + 0.5f) * (1.0f / static_cast<float>(nk))), c = idx - i * nk;
            K[idx] = c < NX ? -H[(NX + i) * n + c] : -h[NX + i];
        }
        LdsBarrier();
        bool bad = false;
        // L D L^T of the 4 x 4 diagonal block at J0 (lower triangle read): reciprocal pivots and unit-lower entries
        struct Block {
            double i0, i1, i2, i3, l10, l20, l21, l30, l31, l32;
        };
        auto factorBlock = [&](int J0) {
            const double* D = R + J0 * n + J0;
            const double d00 = D[0], d10 = D[n], d11 = D[n + 1], d20 = D[2 * n], d21 = D[2 * n + 1], d22 = D[2 * n + 2], d30 = D[3 * n], d31 = D[3 * n + 1], d32 = D[3 * n + 2],
                         d33 = D[3 * n + 3];
            auto reciprocal = [&](double d) {
                const bool neg = !(d > 0.0);
                bad = bad || neg;
#ifdef UNGAR_RICCATI_FAST_RECIPROCAL  // (measurement variant: v_rcp_f64 and two Newton steps instead of the IEEE division)
                const double x = neg ? 1.0 : d;
                double r = __builtin_amdgcn_rcp(x);
                r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
                return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#else
                return 1.0 / (neg ? 1.0 : d);
#endif
            };
            Block B;
            B.i0 = reciprocal(d00);
            B.l10 = d10 * B.i0;
            B.l20 = d20 * B.i0;
            B.l30 = d30 * B.i0;
            B.i1 = reciprocal(d11 - d10 * B.l10);
            const double u21 = d21 - d20 * B.l10, u31 = d31 - d30 * B.l10;
            B.l21 = u21 * B.i1;
            B.l31 = u31 * B.i1;
            B.i2 = reciprocal(d22 - d20 * B.l20 - u21 * B.l21);
            const double u32 = d32 - d30 * B.l20 - u31 * B.l21;
            B.l32 = u32 * B.i2;
            B.i3 = reciprocal(d33 - d30 * B.l30 - u31 * B.l31 - u32 * B.l32);
            return B;
        };
        // one rank-4 update of a 16 x 16 tile of C (leading dimension ldc, rows r0.., columns c0.., valid below rowEnd / colEnd):
        //   C[r][c] += sum_k A(r, k) B(k, c);  A / B clamp their addresses (padded rows / columns feed outputs that are not stored)
        auto updateTile = [&](double* C, int ldc, int r0, int rowEnd, int c0, int colEnd, auto A, auto Bop) {
            f64x4 acc;
            const int col = c0 + li, cc = col < colEnd ? col : colEnd - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + lk + 4 * r, rc = row < rowEnd ? row : rowEnd - 1;
                acc[r] = C[rc * ldc + cc];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A(r0 + li, lk), Bop(lk, c0 + li), acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + lk + 4 * r;
                if (row < rowEnd && col < colEnd) C[row * ldc + col] = acc[r];
            }
        };
#pragma nounroll
        for (int b = 0; b < STEPS; ++b) {
            const int J0 = 4 * b, i0 = J0 + 4, rest = NU - i0;
            if (w == 0) {  // P_b: one wavefront (its loads of the diagonal block precede its stores into it in program order)
                const Block B = factorBlock(J0);
                const int i = J0 + lane;
                if (lane < ROWL && i < NU) {
                    double* ri = R + i * n + J0;
                    const double u0 = ri[0], p1 = ri[1], p2 = ri[2], p3 = ri[3];  // (rows inside the block: entries above the diagonal are never read again)
                    const double u1 = p1 - u0 * B.l10, u2 = p2 - u0 * B.l20 - u1 * B.l21;
                    ri[1] = u1;
                    ri[2] = u2;
                    ri[3] = p3 - u0 * B.l30 - u1 * B.l31 - u2 * B.l32;
                } else if (lane >= ROWL && lane - ROWL < nk) {
                    double* kc = K + J0 * nk + (lane - ROWL);
                    const double y0 = kc[0], y1 = kc[nk] - B.l10 * y0, y2 = kc[2 * nk] - B.l20 * y0 - B.l21 * y1;
                    kc[nk] = y1;
                    kc[2 * nk] = y2;
                    kc[3 * nk] -= B.l30 * y0 + B.l31 * y1 + B.l32 * y2;
                }
                if (lane == 0) {
                    piv[J0] = B.i0;
                    piv[J0 + 1] = B.i1;
                    piv[J0 + 2] = B.i2;
                    piv[J0 + 3] = B.i3;
                }
            }
            LdsBarrier();
            if (rest > 0) {  // T_b
                const int TR = (rest + 15) >> 4, nR = TR * (TR + 1) / 2, NT = nR + TR * TCK;
                auto A = [&](int row, int k) {  // -(L d)[row][J0 + k] / d_(J0 + k)   (a padded row feeds an output row that is not stored)
                    const int rc = row < NU ? row : NU - 1;
                    return -(R[rc * n + J0 + k] * piv[J0 + k]);
                };
                for (int t = w; t < NT; t += kWaves) {
                    if (t < nR) {
                        const int ti = t == 0 ? 0 : 1, tj = t == 2 ? 1 : 0;
                        updateTile(R, n, i0 + 16 * ti, NU, i0 + 16 * tj, NU, A, [&](int k, int col) {
                            const int cc = col < NU ? col : NU - 1;
                            return R[cc * n + J0 + k];
                        });
                    } else {
                        const int q = t - nR, ti = q / TCK, tc = q - ti * TCK;
                        updateTile(K, nk, i0 + 16 * ti, NU, 16 * tc, nk, A, [&](int k, int col) {
                            const int cc = col < nk ? col : nk - 1;
                            return K[(J0 + k) * nk + cc];
                        });
                    }
                }
                LdsBarrier();
            }
        }
#pragma nounroll
        for (int b = STEPS - 1; b >= 0; --b) {
            const int J0 = 4 * b;
            if (w == 0 && lane < nk) {  // B1_b
                const double* D = R + J0 * n + J0;
                const double u10 = D[n], u20 = D[2 * n], u21 = D[2 * n + 1], u30 = D[3 * n], u31 = D[3 * n + 1], u32 = D[3 * n + 2];
                double* kc = K + J0 * nk + lane;
                const double x3 = kc[3 * nk] * piv[J0 + 3];
                const double x2 = (kc[2 * nk] - u32 * x3) * piv[J0 + 2];
                const double x1 = (kc[nk] - u21 * x2 - u31 * x3) * piv[J0 + 1];
                const double x0 = (kc[0] - u10 * x1 - u20 * x2 - u30 * x3) * piv[J0];
                kc[0] = x0;
                kc[nk] = x1;
                kc[2 * nk] = x2;
                kc[3 * nk] = x3;
                double* gc = gainsK + J0 * nk + lane;
                gc[0] = x0;
                gc[nk] = x1;
                gc[2 * nk] = x2;
                gc[3 * nk] = x3;
            }
            LdsBarrier();
            if (J0 > 0) {  // B2_b
                const int TR = (J0 + 15) >> 4, NT = TR * TCK;
                for (int t = w; t < NT; t += kWaves) {
                    const int ti = t / TCK, tc = t - ti * TCK;
                    updateTile(
                        K, nk, 16 * ti, J0, 16 * tc, nk,
                        [&](int row, int k) {  // -(L d)[J0 + k][row]
                            const int rc = row < J0 ? row : J0 - 1;
                            return -R[(J0 + k) * n + rc];
                        },
                        [&](int k, int col) {
                            const int cc = col < nk ? col : nk - 1;
                            return K[(J0 + k) * nk + cc];
                        });
                }
                LdsBarrier();
            }
        }
        return bad;
    }


#ifdef UNGAR_RICCATI_CLOCKS
    /// Diagnostic build: cycles of the first workgroup's first lane between consecutive marks, summed per mark id
    /// (read back with ungar_amd_debug_riccati_clocks; tools/bench_riccati_phases.py).
    unsigned long long last = 0;
    __device__ __forceinline__ void Mark(int id);
#endif
};

#ifdef UNGAR_RICCATI_CLOCKS
__device__ unsigned long long gRiccatiClocks[8];
template <int BLOCK, int SLOTS_AB, int SLOTS_W, bool AHEAD, bool DMA>
__device__ __forceinline__ void DeviceExec<BLOCK, SLOTS_AB, SLOTS_W, AHEAD, DMA>::Mark(int id) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (id > 0) gRiccatiClocks[id] += now - last;
        last = now;
    }
}
#endif

/// Wavefronts per SIMD the register allocation has to leave room for: what the LDS footprint of the instantiation admits anyway
/// (37 + 12: two workgroups of four wavefronts per CU; 13 + 4 on one wavefront: four).  Without the bound the allocator takes a
/// few registers more than the next occupancy step allows (131 of 128, 256 + AGPRs of 256) for no measurable gain per wavefront.
template <int BLOCK, int NX, int NU>
#ifdef UNGAR_RICCATI_NO_OCCUPANCY_BOUND
inline constexpr int kRiccatiWavesPerSimd = 1;
#else
inline constexpr int kRiccatiWavesPerSimd = ((NX == 37 && NU == 12) || (NX == 25 && NU == 24 && BLOCK == 256)) ? 2 : (NX == 13 && NU == 4 && BLOCK == 64) ? 4 : 1;
#endif

template <int BLOCK, int SLOTS_AB, int SLOTS_W, int NX = 0, int NU = 0, bool AHEAD = true, bool DMA = false, int NE = 0>
__global__ __launch_bounds__(BLOCK, (kRiccatiWavesPerSimd<BLOCK, NX, NU>)) void RiccatiKernel(const RiccatiArgs a) {
    extern __shared__ double scratch[];
    const long long inst = blockIdx.x;
    if (inst >= a.batch) return;
    DeviceExec<BLOCK, SLOTS_AB, SLOTS_W, AHEAD, DMA> ex;
    RiccatiInstance<DeviceExec<BLOCK, SLOTS_AB, SLOTS_W, AHEAD, DMA>, NX, NU, NE>(a, inst, scratch, ex);
}

/// One workgroup per (instance, knot): the inequality Jacobian of the knot and the barrier derivatives are staged in LDS once,
/// then the lanes share the (nx+nu)^2 entries of the block.  (A first version walked the knots of an instance in sequence and
/// re-read the Jacobian from global memory for every entry: 0.73 ms per 4096 x 30 quadrotor knots -- as long as the Riccati
/// solve; this one is bound by writing the blocks.)
__global__ __launch_bounds__(256) void StageQpKernel(const StageQpArgs a) {
    extern __shared__ double lds[];  // [nh][n] inequality Jacobian, d1[nh], d2[nh], then the dense cost Hessian block [n][n]
    const long long node = blockIdx.x;
    const long long b = node / a.N;
    const int k = static_cast<int>(node - b * a.N);
    if (b >= a.batch) return;
    const int lane = static_cast<int>(threadIdx.x), n = a.nx + a.nu, lanes = static_cast<int>(blockDim.x);  // 64 lanes for small stages, 256 from n = 32 on
    double* jh = lds;
    double* d1 = lds + a.nh * n;
    double* d2 = d1 + a.nh;
    double* hc = d2 + a.nh;
    const bool ineq = a.h.base != nullptr;
    for (int idx = lane; idx < n * n; idx += lanes) hc[idx] = 0.0;
    __syncthreads();
    for (int e = lane; e < a.hesNnz; e += lanes) hc[a.hesRow[e] * n + a.hesCol[e]] = a.costHes.at(b, k, e);  // distinct (row, col) per entry
    if (k == 0)
        for (int i = lane; i < a.nx; i += lanes) a.dx0.at(b, 0, i) = a.xm.at(b, 0, i) - a.X.at(b, 0, i);
    for (int i = lane; i < a.nx; i += lanes) a.b.at(b, k, i) = a.f.at(b, k, i) - a.X.at(b, k + 1, i);
    if (ineq) {
        for (int idx = lane; idx < a.nh * n; idx += lanes) jh[idx] = a.hJac.at(b, k, idx);
        for (int j = lane; j < a.nh; j += lanes) {
            const double z = -a.h.at(b, k, j);
            d1[j] = BarrierD1(a.barrier, z);
            d2[j] = BarrierD2(a.barrier, z);
        }
    }
    __syncthreads();
    for (int idx = lane; idx < n * n; idx += lanes) {  // upper triangle only: nobody reads the rest (ungar_ocp_riccati_solve: "upper triangle read"),
        const int r = idx / n, c = idx % n;            // and writing it was half of this kernel's HBM traffic
        if (r > c) continue;
        double acc = hc[idx];
        if (ineq)
            for (int j = 0; j < a.nh; ++j) acc += d2[j] * jh[j * n + r] * jh[j * n + c];
        a.hess.at(b, k, idx) = acc;
    }
    for (int c = lane; c < n; c += lanes) {
        double acc = a.costGrad.at(b, k, c);
        if (ineq)
            for (int j = 0; j < a.nh; ++j) acc -= d1[j] * jh[j * n + c];  // d/dz b(-h) = -b'(-h) dh/dz
        a.grad.at(b, k, c) = acc;
    }
}

__global__ __launch_bounds__(kBlock) void MeritKernel(const MeritArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    const int lane = static_cast<int>(threadIdx.x);
    double g2 = 0.0, phi = 0.0, slope = 0.0;
    const long long bm = a.xmPeriod > 0 ? b % a.xmPeriod : b;  // stacked trial points share the measured state of their instance
    for (int i = lane; i < a.nx; i += kBlock) {
        const double r = a.X.at(b, 0, i) - a.xm.at(bm, 0, i);
        g2 += r * r;
    }
    for (int idx = lane; idx < a.N * a.nx; idx += kBlock) {
        const int k = idx / a.nx, i = idx % a.nx;
        const double r = a.X.at(b, k + 1, i) - a.f.at(b, k, i);
        g2 += r * r;
    }
    if (a.cost.base)
        for (int k = lane; k < a.N; k += kBlock) phi += a.cost.at(b, k, 0);
    if (a.costN.base && lane == 0) phi += a.costN.at(b, 0, 0);
    if (a.h.base)
        for (int idx = lane; idx < a.N * a.nh; idx += kBlock) phi += Barrier(a.barrier, -a.h.at(b, idx / a.nh, idx % a.nh));
    const bool wantSlope = a.grad.base && a.dX.base && a.slope;
    if (wantSlope) {
        const int n = a.nx + a.nu;
        for (int idx = lane; idx < a.N * n; idx += kBlock) {
            const int k = idx / n, c = idx % n;
            slope += a.grad.at(b, k, c) * (c < a.nx ? a.dX.at(b, k, c) : a.dU.at(b, k, c - a.nx));
        }
        if (a.gradN.base)
            for (int i = lane; i < a.nx; i += kBlock) slope += a.gradN.at(b, 0, i) * a.dX.at(b, a.N, i);
    }
    g2 = WaveSum(g2);
    phi = WaveSum(phi);
    slope = WaveSum(slope);
    if (lane == 0) {
        a.theta[b] = a.violationMultiplier * sqrt(g2);
        a.phi[b] = phi;
        if (wantSlope) a.slope[b] = slope;
    }
}

__global__ __launch_bounds__(kBlock) void TrialKernel(const TrialArgs a) {
    const long long out = blockIdx.x;  // stacked: candidate out / batch of instance out % batch
    const long long b = a.candidates > 0 ? out % a.batch : out;
    if (out >= (a.candidates > 0 ? a.candidates * a.batch : a.batch)) return;
    const double alpha = a.candidates > 0 ? a.alphas[out / a.batch] : a.alpha;
    for (int idx = static_cast<int>(threadIdx.x); idx < (a.N + 1) * a.nx; idx += kBlock) {
        const int k = idx / a.nx, i = idx % a.nx;
        a.Xt.at(out, k, i) = a.X.at(b, k, i) + alpha * a.dX.at(b, k, i);
    }
    for (int idx = static_cast<int>(threadIdx.x); idx < a.N * a.nu; idx += kBlock) {
        const int k = idx / a.nu, i = idx % a.nu;
        a.Ut.at(out, k, i) = a.U.at(b, k, i) + alpha * a.dU.at(b, k, i);
    }
}

__global__ __launch_bounds__(kBlock) void SelectKernel(const SelectArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    const double theta = a.theta0[b], phi = a.phi0[b], slope = a.slope[b];
    int chosen = -1;  // uniform over the workgroup: every lane runs the same scalar search
    const bool solved = !a.status || a.status[b] == 0;
    for (int c = 0; solved && c < a.candidates && chosen < 0; ++c)
        if (StepAcceptable(theta, phi, slope, a.thetaT[c * a.batch + b], a.phiT[c * a.batch + b], a.alphas[c], a.thetaMin, a.thetaMax, a.eta, a.gammaPhi, a.gammaTheta)) chosen = c;
    if (chosen < 0) {
        if (threadIdx.x == 0) a.accepted[b] = 0.0;
        return;
    }
    const long long from = chosen * a.batch + b;
    for (int idx = static_cast<int>(threadIdx.x); idx < (a.N + 1) * a.nx; idx += kBlock) a.X.at(b, idx / a.nx, idx % a.nx) = a.Xt.at(from, idx / a.nx, idx % a.nx);
    for (int idx = static_cast<int>(threadIdx.x); idx < a.N * a.nu; idx += kBlock) a.U.at(b, idx / a.nu, idx % a.nu) = a.Ut.at(from, idx / a.nu, idx % a.nu);
    if (threadIdx.x == 0) a.accepted[b] = a.alphas[chosen];
}

__global__ __launch_bounds__(kBlock) void AcceptKernel(const AcceptArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    if (a.accepted[b] != 0.0) return;  // uniform over the workgroup
    if (a.status && a.status[b] != 0) return;
    const double theta = a.theta0[b], phi = a.phi0[b], thetaNext = a.thetaT[b], phiNext = a.phiT[b], slope = a.slope[b];
    if (!StepAcceptable(theta, phi, slope, thetaNext, phiNext, a.alpha, a.thetaMin, a.thetaMax, a.eta, a.gammaPhi, a.gammaTheta)) return;
    for (int idx = static_cast<int>(threadIdx.x); idx < (a.N + 1) * a.nx; idx += kBlock) a.X.at(b, idx / a.nx, idx % a.nx) = a.Xt.at(b, idx / a.nx, idx % a.nx);
    for (int idx = static_cast<int>(threadIdx.x); idx < a.N * a.nu; idx += kBlock) a.U.at(b, idx / a.nu, idx % a.nu) = a.Ut.at(b, idx / a.nu, idx % a.nu);
    __syncthreads();
    if (threadIdx.x == 0) a.accepted[b] = a.alpha;
}

}  // namespace
}  // namespace ungar_amd::kernels

using namespace ungar_amd::kernels;

namespace {
template <int BLOCK, int SLOTS_AB, int SLOTS_W, int NX = 0, int NU = 0, bool AHEAD = true, bool DMA = false, int NE = 0>
int LaunchRiccati(const RiccatiArgs* a, std::size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) {  // above the default dynamic-LDS limit the kernel has to opt in (160 KiB per CU on gfx950)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(RiccatiKernel<BLOCK, SLOTS_AB, SLOTS_W, NX, NU, AHEAD, DMA, NE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL((RiccatiKernel<BLOCK, SLOTS_AB, SLOTS_W, NX, NU, AHEAD, DMA, NE>), dim3(static_cast<unsigned>(a->batch)), dim3(BLOCK), lds, stream, *a);
    return static_cast<int>(hipGetLastError());
}
}  // namespace

extern "C" int ungar_amd_launch_riccati_wave(const RiccatiArgs* a, void* stream);  // ocp_riccati_wave.hip: one wavefront per instance, matrices in registers

extern "C" int ungar_amd_launch_riccati(const RiccatiArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    {
        // Default route for the sizes it is instantiated for (the large blocks: 37 + 12, 25 + 24, 13 + 24, no equality rows inside the recursion): the
        // register-resident one-wavefront kernels.  UNGAR_AMD_RICCATI_VARIANT (any value, e.g. "fixed") keeps the LDS-resident kernels below: the A/B switch.
        const bool ldsResident = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_RICCATI_VARIANT") != nullptr;  // (read per call: a test switches routes inside one process)
        if (!ldsResident) {
            const int e = ungar_amd_launch_riccati_wave(a, stream);
            if (e >= 0) return e;
        }
    }
    const std::size_t lds = static_cast<std::size_t>(RiccatiScratchDoubles(a->nx, a->nu, a->ne)) * sizeof(double);
    if (lds > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int n = a->nx + a->nu, ab = a->nx * n, w = n * n;
    // The reference's OCPs with their carried quantities in the stage state (ocp_shooting.hpp): quadrotor [u_prev; x] 17 + 4,
    // RC car 8 + 2 (one wavefront, next knot's operands staged in registers), quadruped [feet_prev; x] 25 + 24 with its 16
    // foot-contact rows (four wavefronts, operands by LDS-DMA).
    if (a->ne == 0 && a->nx == 17 && a->nu == 4) return LaunchRiccati<64, 6, 7, 17, 4>(a, lds, s);
    if (a->ne == 0 && a->nx == 8 && a->nu == 2) return LaunchRiccati<64, 2, 2, 8, 2>(a, lds, s);
    if (a->ne == 16 && a->nx == 25 && a->nu == 24) return LaunchRiccati<256, 0, 0, 25, 24, true, true, 16>(a, lds, s);
    if (a->ne == 0 && a->nx == 25 && a->nu == 24) return LaunchRiccati<256, 0, 0, 25, 24, true, true>(a, lds, s);  // ... with the rows eliminated before the recursion
    if (a->ne > 0) return n >= 32 ? LaunchRiccati<256, 0, 0>(a, lds, s) : LaunchRiccati<64, 0, 0>(a, lds, s);  // run-time sizes
    // Variants of the same recursion (tools/bench_sqp.py): lanes per instance (one or four wavefronts
    // share the entries of every product) and whether the next knot's operands are staged in registers.
    //   "64" / "128" / "256"   one / two / four wavefronts per instance, operands read in place;   "128p" / "256p"   staged
#if !UNGAR_AMD_MEASUREMENT_BUILD
    // shipped library: the default route only (the variants below exist in the measurement build, tools/bench_sqp.py)
    if (a->nx == 13 && a->nu == 4) return LaunchRiccati<64, 4, 5, 13, 4>(a, lds, s);
    if (a->nx == 6 && a->nu == 2) return LaunchRiccati<64, 1, 1, 6, 2>(a, lds, s);
    if (a->nx == 13 && a->nu == 24) return LaunchRiccati<256, 0, 0, 13, 24, true, true>(a, lds, s);
    if (a->nx == 37 && a->nu == 12) return LaunchRiccati<256, 0, 0, 37, 12, true, true>(a, lds, s);
    (void)ab, (void)w;
    return LaunchRiccati<64, 0, 0>(a, lds, s);
#else
    static const std::string variant = [] {
        const char* e = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_RICCATI_VARIANT");
        return std::string(e ? e : "fixed");
    }();
    // sizes of the reference's three OCPs and of the full-body quadruped, fixed at compile time (default route)
    if (variant == "fixedp") {  // compile-time sizes AND the next knot's operands staged in registers while this knot is processed
        if (a->nx == 13 && a->nu == 4) return LaunchRiccati<64, 4, 5, 13, 4>(a, lds, s);
        if (a->nx == 6 && a->nu == 2) return LaunchRiccati<64, 1, 1, 6, 2>(a, lds, s);
        if (a->nx == 13 && a->nu == 24) return LaunchRiccati<256, 2, 6, 13, 24>(a, lds, s);
        if (a->nx == 37 && a->nu == 12) return LaunchRiccati<256, 8, 10, 37, 12>(a, lds, s);
    }
    if (variant == "fixed" || variant == "fixeds" || variant == "fixed1" || variant == "fixedp" || variant == "fixedn" || variant == "fixedq" || variant == "64") {
        // small blocks: the next knot's operands are staged in registers while this knot is processed (quadrotor QP step 1.03 -> 0.92 ms;
        // for the 37 + 12 block the staging registers cost more than the hidden latency: 8.9 -> 9.7 ms)
        if (variant == "fixedq" && a->nx == 13 && a->nu == 4) return LaunchRiccati<64, 0, 0, 13, 4, true, true>(a, lds, s);  // one wavefront, backward operands by LDS-DMA too: 0.877 vs 0.852 ms (register staging wins here)
        if (variant == "fixedq" && a->nx == 6 && a->nu == 2) return LaunchRiccati<64, 0, 0, 6, 2, true, true>(a, lds, s);
        if ((variant == "fixed" || variant == "fixedn") && a->nx == 13 && a->nu == 4) return LaunchRiccati<64, 4, 5, 13, 4>(a, lds, s);
        if ((variant == "fixed" || variant == "fixedn") && a->nx == 6 && a->nu == 2) return LaunchRiccati<64, 1, 1, 6, 2>(a, lds, s);
        if (variant == "fixed1" && a->nx == 13 && a->nu == 4) return LaunchRiccati<64, 0, 0, 13, 4>(a, lds, s);
        if (variant == "fixed1" && a->nx == 6 && a->nu == 2) return LaunchRiccati<64, 0, 0, 6, 2>(a, lds, s);
        // the two large blocks fit only 2-4 instances per CU by their LDS (74 / 30 KB): four wavefronts per instance keep the SIMDs busy
        // (full-body quadruped, 4096 instances x N = 20: QP step 20.7 ms with one wavefront per instance, 14.0 generic x 4 wavefronts)
        // ... and get their operands by LDS-DMA one knot ahead, the forward pass three knots ahead (no registers held: the staging
        // registers of "fixeds" cost the second workgroup per CU).  QP step, 4096 instances: 6.74 -> 5.81 ms (37 + 12), 5.47 -> 5.31 ms (13 + 24);
        // "fixedn" = the same kernels reading their operands in place at the top of every knot
        if (variant == "fixed" && a->nx == 13 && a->nu == 24) return LaunchRiccati<256, 0, 0, 13, 24, true, true>(a, lds, s);
        if (variant == "fixed" && a->nx == 37 && a->nu == 12) return LaunchRiccati<256, 0, 0, 37, 12, true, true>(a, lds, s);
        if (variant == "fixedn" && a->nx == 13 && a->nu == 24) return LaunchRiccati<256, 0, 0, 13, 24>(a, lds, s);
        if (variant == "fixedn" && a->nx == 37 && a->nu == 12) return LaunchRiccati<256, 0, 0, 37, 12>(a, lds, s);
        // "fixeds": each knot's operands staged in registers at its top (17 loads per lane back to back instead of one latency each): the
        // recursion itself gets 1.5x faster per instance, but the 40 staging registers push the kernel past 256 -> one workgroup per CU: slower overall
        if (variant == "fixeds" && a->nx == 37 && a->nu == 12) return LaunchRiccati<256, 8, 10, 37, 12, false>(a, lds, s);
        if (variant == "fixed1" && a->nx == 13 && a->nu == 24) return LaunchRiccati<64, 0, 0, 13, 24>(a, lds, s);
        if (variant == "fixed1" && a->nx == 37 && a->nu == 12) return LaunchRiccati<64, 0, 0, 37, 12>(a, lds, s);
        return LaunchRiccati<64, 0, 0>(a, lds, s);
    }
    if (variant == "128") return LaunchRiccati<128, 0, 0>(a, lds, s);
    if (variant == "128p" && ab <= 768 && w <= 768) return LaunchRiccati<128, 6, 6>(a, lds, s);
    if (variant == "128p" && ab <= 1792 && w <= 1792) return LaunchRiccati<128, 14, 14>(a, lds, s);
    if (variant == "128p") return LaunchRiccati<128, 0, 0>(a, lds, s);
    if (variant == "256p" && ab <= 768 && w <= 768) return LaunchRiccati<256, 3, 3>(a, lds, s);    // quadrotor, rc_car (n <= 27)
    if (variant == "256p" && ab <= 1792 && w <= 1792) return LaunchRiccati<256, 7, 7>(a, lds, s);  // single-rigid-body quadruped (n <= 42)
    return LaunchRiccati<256, 0, 0>(a, lds, s);
#endif
}

extern "C" int ungar_amd_launch_ocp_stage_qp(const StageQpArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    if (a->nh > 64 || a->hesNnz > 160) return static_cast<int>(hipErrorInvalidValue);
    const std::size_t n = static_cast<std::size_t>(a->nx + a->nu);
    const std::size_t lds = (static_cast<std::size_t>(a->nh) * n + 2 * static_cast<std::size_t>(a->nh) + n * n) * sizeof(double);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(StageQpKernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(StageQpKernel, dim3(static_cast<unsigned>(a->batch * a->N)), dim3(n >= 32 ? 256 : kBlock), lds, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_merit(const MeritArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    hipLaunchKernelGGL(MeritKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_trial(const TrialArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    const long long groups = a->candidates > 0 ? a->candidates * a->batch : a->batch;
    hipLaunchKernelGGL(TrialKernel, dim3(static_cast<unsigned>(groups)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_select(const SelectArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    hipLaunchKernelGGL(SelectKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_accept(const AcceptArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    hipLaunchKernelGGL(AcceptKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

#ifdef UNGAR_RICCATI_CLOCKS
/// Diagnostic builds only (tools/make_riccati_clocks.sh): copies and clears the phase clocks of DeviceExec::Mark.
extern "C" int ungar_amd_debug_riccati_clocks(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(ungar_amd::kernels::gRiccatiClocks), 8 * sizeof(unsigned long long));
    if (e != hipSuccess) return static_cast<int>(e);
    const unsigned long long zero[8] = {};
    return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(ungar_amd::kernels::gRiccatiClocks), zero, sizeof zero));
}
#endif
