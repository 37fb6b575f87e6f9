// ungar_amd :: batched SQP kernels for shooting problems with carried quantities and stage equality rows (ocp_shooting.hpp):
// QP data from the stage functions' sparse outputs, merit terms, stacked trial rows, and the per-instance line search + iteration
// bookkeeping of SoftSQPOptimizer::Optimize (reference include/ungar/optimization/soft_sqp.hpp:62-112, 143-158, 247-264;
// backtracking_line_search.hpp:116-151).  The QP itself is solved by the Riccati kernel (ocp_riccati.hip) on what is assembled here.
#include "../runtime/measurement.hpp"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>


#include "../../../include/ungar_amd.h"
#include "../runtime/kernel_jit.hpp"
#include "ocp_shooting_wave_kernel.hpp"

namespace ungar_amd::kernels {
namespace {

/// One workgroup per node (instance, knot <= N).  Sparse values are scattered into dense LDS images first (a dense block is written
/// to global memory exactly once, coalesced; zero-filling and scattering in global memory would race), then the barrier terms are added
/// and -- on request -- the stage equality rows eliminated.  Dense loops map (wavefront -> row, lane -> column): no index divisions.
// Four wavefronts per SIMD: with W packed, four quadruped nodes (34 KB of LDS each) fit a CU, and the register budget has to admit them too (128 per lane: the
// k-steps of the substitution then go in groups of two).  UNGAR_ASSEMBLE_WAVES_PER_EU / UNGAR_ASSEMBLE_K_GROUP are measurement knobs (tools/make_shooting_variants.sh).
#ifndef UNGAR_ASSEMBLE_WAVES_PER_EU
#define UNGAR_ASSEMBLE_WAVES_PER_EU 4
#endif
#ifndef UNGAR_ASSEMBLE_K_GROUP
#define UNGAR_ASSEMBLE_K_GROUP 2
#endif
__global__ __launch_bounds__(256, UNGAR_ASSEMBLE_WAVES_PER_EU) void ShootingAssembleKernel(const ShootingAssembleArgs a) {
    extern __shared__ double lds[];
    const ShootingDims& d = a.d;
    const long long node = blockIdx.x;
    const long long b = node / (d.N + 1);
    const int k = static_cast<int>(node - b * (d.N + 1));
    if (b >= d.batch) return;
    const int lane = static_cast<int>(threadIdx.x), lanes = static_cast<int>(blockDim.x);
    const int wave = __builtin_amdgcn_readfirstlane(lane >> 6), wl = lane & 63, waves = lanes >> 6;  // (the wavefront's index in a scalar register: tile loops and the single-wavefront jobs branch without execution masks)
    const int nz = d.nz(), nd = d.nd(), nc = d.nc, nx = d.nx;
    const bool stage = k < d.N;  // knot N: terminal cost only
    const float ndInv = 1.0f / static_cast<float>(nd);  // row of a flat index through a float reciprocal: exact for nd <= 256 (idx + 0.5 is never a multiple of nd)
    // W lives as its PACKED upper triangle (row r holds columns r .. nd - 1): 9.8 instead of 19.2 KB for the quadruped's 49 x 49, the difference between three and
    // four nodes per CU for this latency-bound kernel; no mirrored copy to maintain
    const int nW = nd * (nd + 1) / 2;
    auto tri = [nd](int r, int c) { return ((r * (2 * nd + 1 - r)) >> 1) + (c - r); };  // r <= c
    auto sym = [&tri](int r, int c) { return r <= c ? tri(r, c) : tri(c, r); };
    double* Wd = lds;                 // nd (nd + 1) / 2
    double* gd = Wd + nW;             // nd
    double* d1 = gd + nd;             // nh
    double* d2 = d1 + a.nh;           // nh
    double* ABd = d2 + a.nh;          // nz x nd
    const int ld = nd + 1;            // equality tableau [C | D | e]: row stride
    double* Ed = ABd + nz * nd;       // ne x (nd + 1)
    double* bd = Ed + a.ne * ld;      // nz
    double* prow = bd + nz;           // nd + 1: w + W s of the substitution
    double* vbuf = prow + nd + 1;     // 6 + ne x (nd + 1): pivot-search keys, the second tableau of the elimination, then W_BB G
    // ne: bit pattern of the largest |entry| of every ORIGINAL equality row (Jacobian entries and residual), collected while the row is scattered.  A row that the
    // elimination reduces to rounding noise of its original entries (a redundant / linearly dependent row: ~1e-16 of its scale, never exactly zero) is recognised
    // against this scale -- relative to its own reduced entries the noise would pass as a pivot and 1 / pivot would blow up W, [A|B] and w.
    unsigned long long* rowScale = reinterpret_cast<unsigned long long*>(vbuf + 6 + a.ne * ld);
    int* pivCol = reinterpret_cast<int*>(vbuf + 6 + a.ne * ld + a.ne);  // ne pivot inputs
    int* used = pivCol + a.ne;        // nu flags
    int* list = used + d.nu;          // max(nd, ne) indices of a support
    int* listSize = list + (nd > a.ne ? nd : a.ne);
#ifdef UNGAR_SHOOTING_CLOCKS  // diagnostic build (tools/make_shooting_clocks.sh): cycles of the sections of one stage node, printed by its first lane
    __shared__ unsigned long long jobCycles[9];  // the single-wavefront jobs: [wavefront], [4] = barrier terms of wavefront 1
    if (threadIdx.x < 9) jobCycles[threadIdx.x] = 0ull;
    unsigned long long marks[12];
    int markCount = 0;
#define UNGAR_SHOOTING_MARK() marks[markCount++] = __builtin_amdgcn_s_memtime()
#else
#define UNGAR_SHOOTING_MARK() ((void)0)
#endif
    UNGAR_SHOOTING_MARK();
    const int total = nW + nd + 2 * a.nh + nz * nd + a.ne * ld;
    const long long nodeOff = b * (d.N + 1) + k;
    const int elim = a.eliminate & 3;  // (bit 2: measurement switch -- the generic sections below for every node)
    const long long stageOff = b * d.N + k;
    // A stage node with few equality rows and a tableau no wider than a wavefront (the quadruped: 16 x 50) runs the next three sections as
    // INDEPENDENT jobs of single wavefronts, with no workgroup barrier until all are done: the Gauss-Jordan elimination in the registers of
    // wavefront 0 (16 barrier-separated LDS sweeps before: 35-51 k of a node's ~115 k cycles), the barrier terms + regularisation + mirror in
    // wavefront 1 (in-order LDS traffic of one wavefront replaces a workgroup barrier per inequality row), the defect b in the others.
    const bool specialised = stage && elim && a.ne > 0 && a.ne <= kRegisterRows && ld <= 64 && a.ph.nnz <= 64 && waves >= 4 && !(a.eliminate & 4);
    const int role = wave;  // (rotating the jobs over the wavefronts with the node index -- other SIMDs for co-resident nodes -- measured: no change)
    // The first entry of every sparse output per lane (all of them, for patterns of up to `lanes` non-zeros) is requested BEFORE the images are zeroed: six
    // scatter loops one after the other were six exposed round trips to memory (~10 k cycles per node).
    struct First {
        int target;  // < 0: none
        double value;
    };
    auto first = [&](const StagePattern& pattern, const double* values, bool wanted, auto targetOf) {
        First f{-1, 0.0};
        if (wanted && values && lane < pattern.nnz) {
            f.target = targetOf(pattern.rows ? pattern.rows[lane] : 0, pattern.cols[lane]);
            f.value = values[nodeOff * pattern.nnz + lane];
        }
        return f;
    };
    auto hessianTarget = [&](int r, int c) { return r <= c ? tri(r, c) : -1; };  // (Function::Hessian's pattern is upper triangular; anything below the diagonal is ignored, as the mirror did)
    auto gradientTarget = [&](int, int c) { return c; };
    auto dynamicsTarget = [&](int r, int c) { return (nc + r) * nd + nc + c; };
    auto carryTarget = [&](int r, int c) { return r * nd + nc + c; };
    auto equalityTarget = [&](int r, int c) { return r * ld + c; };
    const First fH = first(a.pH, a.lH, true, hessianTarget), fg = first(a.pg, a.lg, true, gradientTarget), ff = first(a.pf, a.fJ, stage, dynamicsTarget),
                fc = first(a.pc, a.cJ, stage && !d.carryInputs, carryTarget), fe = first(a.pe, a.eJ, stage, equalityTarget);
    const double hFirst = stage && lane < a.nh ? a.h[nodeOff * a.nh + lane] : 0.0;
    // (what the single-wavefront jobs of a specialised node start from is requested here as well: the inequality Jacobian entry of each lane of wavefront 1 and
    // the equality residuals -- a first memory round trip inside a job is on that job's critical path)
    const bool ownsInequalityEntry = specialised && a.nh > 0 && role == 1 && wl < a.ph.nnz;
    const int inequalityRow = ownsInequalityEntry ? a.ph.rows[wl] : -1, inequalityColumn = ownsInequalityEntry ? a.ph.cols[wl] : 0;
    const double inequalityValue = ownsInequalityEntry ? a.hJ[nodeOff * a.ph.nnz + wl] : 0.0;
    const double residualFirst = specialised && a.e && lane < a.ne ? a.e[nodeOff * a.ne + lane] : 0.0;
    for (int i = lane; i < total; i += lanes) lds[i] = 0.0;
    for (int i = lane; i < a.ne; i += lanes) rowScale[i] = 0ull;
    __syncthreads();
    auto offerScale = [&](int row, double value) {  // (non-negative doubles order like their bit patterns)
        const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(fabs(value)));
        if (bits) atomicMax(&rowScale[row], bits);
    };
    if (fH.target >= 0) Wd[fH.target] = fH.value;
    for (int e = lane + lanes; e < a.pH.nnz; e += lanes)
        if (a.pH.rows[e] <= a.pH.cols[e]) Wd[tri(a.pH.rows[e], a.pH.cols[e])] = a.lH[nodeOff * a.pH.nnz + e];
    if (fg.target >= 0) gd[fg.target] = fg.value;
    for (int e = lane + lanes; e < a.pg.nnz; e += lanes) gd[a.pg.cols[e]] = a.lg[nodeOff * a.pg.nnz + e];
    if (stage) {
        for (int j = lane; j < a.nh; j += lanes) {
            const double z = -(j == lane ? hFirst : a.h[nodeOff * a.nh + j]);
            d1[j] = BarrierD1(a.barrier, z);
            d2[j] = BarrierD2(a.barrier, z);
        }
        // [A|B]: function column j over [x|u] is column nc + j of [c|x|u]
        if (ff.target >= 0) ABd[ff.target] = ff.value;
        for (int e = lane + lanes; e < a.pf.nnz; e += lanes) ABd[(nc + a.pf.rows[e]) * nd + nc + a.pf.cols[e]] = a.fJ[nodeOff * a.pf.nnz + e];
        if (d.carryInputs) {
            for (int r = lane; r < nc; r += lanes) ABd[r * nd + nz + r] = 1.0;
        } else {
            if (fc.target >= 0) ABd[fc.target] = fc.value;
            for (int e = lane + lanes; e < a.pc.nnz; e += lanes) ABd[a.pc.rows[e] * nd + nc + a.pc.cols[e]] = a.cJ[nodeOff * a.pc.nnz + e];
        }
        if (fe.target >= 0) {
            Ed[fe.target] = fe.value;
            if (elim) offerScale(a.pe.rows[lane], fe.value);
        }
        if (specialised && lane < a.ne) {
            Ed[lane * ld + nd] = residualFirst;  // (the generic sections fill this column later)
            offerScale(lane, residualFirst);
        }
        for (int e = lane + lanes; e < a.pe.nnz; e += lanes) {
            const double v = a.eJ[nodeOff * a.pe.nnz + e];
            Ed[a.pe.rows[e] * ld + a.pe.cols[e]] = v;
            if (elim) offerScale(a.pe.rows[e], v);
        }
    }
    __syncthreads();
    UNGAR_SHOOTING_MARK();  // 1: zeroed images, scattered stage outputs
    if (specialised) {
#ifdef UNGAR_SHOOTING_CLOCKS
        const unsigned long long jobStart = __builtin_amdgcn_s_memtime();
#endif
        // the barrier terms touch W[c1][c2] with c1, c2 >= the smallest column of the inequality Jacobian's pattern: the rows above it are regularised and
        // mirrored by the wavefronts that have nothing else to do, the rest by wavefront 1 once its terms are in
#ifdef UNGAR_ASSEMBLE_MIRROR_ONE_WAVE
        const int firstBarrierColumn = 0;
#else
        const int firstBarrierColumn = role == 0 ? 0 : (a.nh > 0 && a.ph.nnz > 0 ? WaveMinInt(wl < a.ph.nnz ? a.ph.cols[wl] : nd) : nd);
#endif
        auto regularise = [&](int firstRow, int lastRow, int lane0, int step) {  // the reference's 1e-6 I over its decision variables (soft_sqp.hpp:149-151)
            for (int r = firstRow + lane0; r < lastRow; r += step)
                if (r >= nc) Wd[tri(r, r)] += a.regularization;
        };
        if (role == 0) {
            // ---- Gauss-Jordan on [C | D | e], lane = column, the rows in registers: the same pivot rule and the same arithmetic as the generic
            // section below (largest unused input coefficient of the row, compared on the bit pattern with the low byte replaced by 255 - input),
            // entries exchanged with v_readlane and the two maxima of a step reduced with DPP moves.
            double t[kRegisterRows];
#pragma unroll
            for (int r = 0; r < kRegisterRows; ++r) {
                double v = 0.0;
                if (r < a.ne && wl < ld) v = Ed[r * ld + wl];
                t[r] = v;
            }
            // rows that are identically zero now (inactive contacts: half of them on average) stay zero under every update and take no pivot: found once,
            // their steps are skipped (a step is ~200 dependent instructions of this one wavefront)
            unsigned nonZeroRows = 0u;
#pragma unroll
            for (int r = 0; r < kRegisterRows; ++r) nonZeroRows |= (__double_as_longlong(t[r]) << 1) != 0ll ? 1u << r : 0u;
            nonZeroRows = WaveOr(nonZeroRows);
#ifdef UNGAR_SHOOTING_CLOCKS
            if (wl == 0) jobCycles[8] = __popc(nonZeroRows);
#endif
            const double scaleOfMyRow = wl < a.ne ? __longlong_as_double(static_cast<long long>(rowScale[wl])) : 0.0;  // lane r: original scale of row r
            unsigned long long taken = 0ull;  // inputs that are pivots already (wave-uniform)
            unsigned pivotRows = 0u;          // rows that took a pivot (wave-uniform)
            const int myInput = wl - nz;
            const bool inputLane = wl >= nz && wl < nd;
#pragma unroll
            for (int i = 0; i < kRegisterRows; ++i) {
                if (i < a.ne && !((nonZeroRows >> i) & 1u)) {
                    if (wl == 0) pivCol[i] = -1;
                } else if (i < a.ne) {  // (uniform)
                    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(fabs(t[i])));
                    const bool candidate = bits != 0ull && inputLane && !((taken >> (myInput & 63)) & 1ull);
                    unsigned long long key = candidate ? (bits & ~0xFFull) | static_cast<unsigned long long>(255 - myInput) : 0ull, rowBits = bits;
                    WaveMaxPair(key, rowBits);
                    const double best = __longlong_as_double(static_cast<long long>(key & ~0xFFull)), rowMax = __longlong_as_double(static_cast<long long>(rowBits));
                    int j = key ? 255 - static_cast<int>(key & 0xFFull) : -1;
                    if (rowMax <= 1e-12 * ReadLane(scaleOfMyRow, i)) {
                        j = -1;  // what is left of the row is rounding noise of its original entries: redundant, no pivot (an inconsistent residual is not noise: below)
                    } else {
                        if (j >= 0 && best <= 1e-12 * rowMax) j = -2;
                        if (j == -1 && rowMax > 0.0) j = -2;
                    }
                    if (wl == 0) pivCol[i] = j;
                    if (j >= 0) {
                        taken |= 1ull << j;
                        pivotRows |= 1u << i;
                        const int J = nz + j;
                        const double rpiv = 1.0 / ReadLane(t[i], J);
                        const double p = wl == J ? 1.0 : t[i] * rpiv;
#pragma unroll
                        for (int r = 0; r < kRegisterRows; ++r) {
                            if (r == i) continue;
                            const double m = ReadLane(t[r], J);
                            t[r] = t[r] - m * p;  // (lane J: m - m * 1 = +0 exactly, the generic section's literal zero)
                        }
                        t[i] = p;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < kRegisterRows; ++r)
                if (r < a.ne && wl < ld) Ed[r * ld + wl] = t[r];
            if (wl < d.nu) used[wl] = static_cast<int>((taken >> wl) & 1ull);
            // the ascending list of pivot rows the substitution walks, while the flags are at hand (the generic path builds it behind a barrier of its own)
            if (wl < a.ne && ((pivotRows >> wl) & 1u)) list[__popc(pivotRows & ((1u << wl) - 1u))] = wl;
            if (wl == 0) *listSize = __popc(pivotRows);
        } else if (role == 1) {
            if (a.nh > 0) {
                const int nnz = a.ph.nnz, mine = wl < nnz ? wl : -1;
                const int myRow = inequalityRow, myCol = inequalityColumn;
                const double myValue = inequalityValue;
                // entries of the same row from this one on (columns ascend within a row: upper triangle; a row's entries are consecutive): the distance to the
                // row's last entry, from one ballot instead of a loop of dependent loads
                const int nextRow = __shfl_down(myRow, 1);
                const unsigned long long rowEnds = __ballot(mine >= 0 && (mine == nnz - 1 || nextRow != myRow));
                const int partners = mine >= 0 ? __ffsll(static_cast<unsigned long long>(rowEnds >> mine)) : 0;
                // one lane per PAIR (e1, e2 >= e1) of a row where the pairs fit a wavefront (the quadruped: 28 entries in 12 rows, 46 pairs): a row is then ONE
                // read-modify-write per lane instead of a chain of `partners` of them.  Pair p of the concatenation over e1 (same targets, same order per target).
                int pairFirst = -1, pairOffset = 0, pairs = 0;
#ifndef UNGAR_ASSEMBLE_ENTRY_LANES
                for (int e1 = 0; e1 < nnz; ++e1) {
                    const int count = __builtin_amdgcn_readlane(partners, e1);
                    if (wl >= pairs && wl < pairs + count) {
                        pairFirst = e1;
                        pairOffset = wl - pairs;
                    }
                    pairs += count;
                }
#else
                pairs = 65;
#endif
                if (pairs <= 64) {
                    const bool havePair = pairFirst >= 0;
                    // (a pair's entries come from the lanes that hold them)
                    const int first = havePair ? pairFirst : 0, second = havePair ? pairFirst + pairOffset : 0;
                    const int pairRowAny = __shfl(myRow, first), c1 = __shfl(myCol, first), c2 = __shfl(myCol, second);
                    const double v1 = __shfl(myValue, first), v2 = __shfl(myValue, second);
                    const int pairRow = havePair ? pairRowAny : -1, target = havePair ? tri(c1, c2) : 0;
                    // row by row, in order (rows share targets): consecutive LDS instructions of ONE wavefront execute in order.  The two reads of a row are issued
                    // together, then the two writes -- one LDS round trip per row; the factors of a lane's row are fetched before the loop.
                    const double d1Mine = myRow >= 0 ? d1[myRow] : 0.0, d2Mine = havePair ? d2[pairRow] : 0.0;
                    for (int j = 0; j < a.nh; ++j) {
                        const bool g = myRow == j, w = pairRow == j;
                        double g0 = 0.0, w0 = 0.0;
                        if (g) g0 = gd[myCol];
                        if (w) w0 = Wd[target];
                        if (g) gd[myCol] = __builtin_fma(-d1Mine, myValue, g0);      // (explicit: left as products, the loop-invariant factors are multiplied out
                        if (w) Wd[target] = __builtin_fma(d2Mine * v1, v2, w0);  // before the loop and rounded once more than the generic section's fused forms)
                        asm volatile("" ::: "memory");  // (order for the compiler only: the hardware executes the LDS instructions of a wavefront in order, so the
                        __builtin_amdgcn_wave_barrier();  // next row's reads see these writes without waiting for them to complete)
                    }
                } else {
                    for (int j = 0; j < a.nh; ++j) {
                        if (myRow == j) {
                            gd[myCol] -= d1[j] * myValue;
                            for (int q = 0; q < partners; ++q) Wd[tri(myCol, a.ph.cols[mine + q])] += d2[j] * myValue * a.hJ[nodeOff * nnz + mine + q];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
#ifdef UNGAR_SHOOTING_CLOCKS
            if (wl == 0) jobCycles[4] = __builtin_amdgcn_s_memtime() - jobStart;
#endif
            regularise(firstBarrierColumn, nd, wl, 64);
        } else {
            const double* next = RowOf(a.rows, d, b, k + 1);
            const int helper = (role - 2) * 64 + wl;  // 0 .. 127 over the two wavefronts without a job of their own
            for (int i = helper; i < nz; i += 128) bd[i] = i < nc ? 0.0 : a.f[nodeOff * nx + (i - nc)] - next[i];
            regularise(0, firstBarrierColumn, helper, 128);
        }
#ifdef UNGAR_SHOOTING_CLOCKS
        if (wl == 0) jobCycles[role] = __builtin_amdgcn_s_memtime() - jobStart;
#endif
        __syncthreads();
        UNGAR_SHOOTING_MARK();  // 2: the jobs
        UNGAR_SHOOTING_MARK();  // 3: regularisation, mirror
    } else {
        // barrier terms  W += J_h^T diag(b''(-h)) J_h,  w -= J_h^T b'(-h)  straight from the SPARSE inequality Jacobian (28 of 588 entries for the
        // quadruped's rows: a dense product over every entry of W read 44 k LDS words per node).  Row by row, in order: inside a row the pairs of its
        // entries hit distinct targets, so plain read-modify-writes suffice and the sums are accumulated in the same order on every run.
        if (stage && a.nh > 0) {
            // the lane's entries (one for up to `lanes` non-zeros, the common case) are fetched once; the row loop below touches LDS only
            const int mine = lane < a.ph.nnz ? lane : -1;
            const int myRow = mine >= 0 ? a.ph.rows[mine] : -1, myCol = mine >= 0 ? a.ph.cols[mine] : 0;
            const double myValue = mine >= 0 ? a.hJ[nodeOff * a.ph.nnz + mine] : 0.0;
            int partners = 0;  // entries of the same row from this one on (columns ascend within a row: upper triangle)
            if (mine >= 0)
                for (int e2 = mine; e2 < a.ph.nnz && a.ph.rows[e2] == myRow; ++e2) ++partners;
            for (int j = 0; j < a.nh; ++j) {
                if (myRow == j) {
                    gd[myCol] -= d1[j] * myValue;  // d/dz b(-h) = -b'(-h) dh/dz
                    for (int t = 0; t < partners; ++t) Wd[tri(myCol, a.ph.cols[mine + t])] += d2[j] * myValue * a.hJ[nodeOff * a.ph.nnz + mine + t];
                }
                for (int e1 = lane + lanes; e1 < a.ph.nnz; e1 += lanes) {  // (patterns with more non-zeros than lanes)
                    if (a.ph.rows[e1] != j) continue;
                    const int c1 = a.ph.cols[e1];
                    const double v1 = a.hJ[nodeOff * a.ph.nnz + e1];
                    gd[c1] -= d1[j] * v1;
                    for (int e2 = e1; e2 < a.ph.nnz && a.ph.rows[e2] == j; ++e2) Wd[tri(c1, a.ph.cols[e2])] += d2[j] * v1 * a.hJ[nodeOff * a.ph.nnz + e2];
                }
                __syncthreads();
            }
        }
        UNGAR_SHOOTING_MARK();  // 2: barrier terms
        // regularisation: the reference's 1e-6 I over its decision variables (soft_sqp.hpp:149-151)
        for (int r = lane; r < nd; r += lanes)
            if (r >= nc && (stage || r < nz)) Wd[tri(r, r)] += a.regularization;
        if (stage) {
            const double* next = RowOf(a.rows, d, b, k + 1);
            for (int i = lane; i < nz; i += lanes) bd[i] = i < nc ? 0.0 : a.f[nodeOff * nx + (i - nc)] - next[i];
            for (int j = lane; j < a.ne; j += lanes) {
                const double residual = a.e ? a.e[nodeOff * a.ne + j] : 0.0;
                Ed[j * ld + nd] = residual;
                if (elim) offerScale(j, residual);
            }
        }
        __syncthreads();
        UNGAR_SHOOTING_MARK();  // 3: regularisation, mirror, b
    }
    if (stage && elim && a.ne > 0) {
        /// list <- {c < n : pred(c)} in ascending order, by the first wavefront (no closing barrier)
        auto buildList = [&](int n, auto pred) {
            if (wave == 0) {
                int size = 0;
                for (int c0 = 0; c0 < n; c0 += 64) {
                    const int c = c0 + wl;
                    const bool in = c < n && pred(c);
                    const unsigned long long mask = __ballot(in);
                    if (in) list[size + __popcll(mask & ((1ull << wl) - 1ull))] = c;
                    size += __popcll(mask);
                }
                if (wl == 0) *listSize = size;
            }
        };
        if (!specialised) {
            // ---- Gauss-Jordan on the tableau [C | D | e]: one pivot input per active row.  The pivot of a row is its largest unused input
            // coefficient, found with ONE LDS atomic per lane: for non-negative doubles the bit pattern orders like the value, so
            // max(bits(|v|) with the low byte replaced by 255 - input) is the largest coefficient up to 2^-44 relative, ties to the lowest input.
            // Step i reads one tableau and writes the other (no read-after-write inside a step) and, while writing row i + 1, already
            // collects that row's keys: ONE barrier per step.  Three key sets rotate so that a set is cleared a full step before its reuse.
            unsigned long long* keys = reinterpret_cast<unsigned long long*>(vbuf);  // set s: [2 s] pivot key, [2 s + 1] bits of the row's largest |entry|
            double* T[2] = {Ed, vbuf + 6};
            const float ldInv = 1.0f / static_cast<float>(ld);
            auto offer = [&](int set, int c, double value, int excluded) {  // entry c of the row whose keys are collected into `set`
                const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(fabs(value)));
                if (!bits) return;
                atomicMax(&keys[2 * set + 1], bits);
                if (c >= nz && c < nd && c - nz != excluded && !used[c - nz]) atomicMax(&keys[2 * set], (bits & ~0xFFull) | static_cast<unsigned long long>(255 - (c - nz)));
            };
            for (int j = lane; j < d.nu; j += lanes) used[j] = 0;
            if (lane < 6) keys[lane] = 0ull;
            __syncthreads();
            for (int c = lane; c < ld; c += lanes) offer(0, c, Ed[c], -1);
            __syncthreads();
            int cur = 0;
            for (int i = 0; i < a.ne; ++i) {
                const double* src = T[cur];
                const int set = i % 3, nextSet = (i + 1) % 3;
                const unsigned long long key = keys[2 * set];
                const double best = __longlong_as_double(static_cast<long long>(key & ~0xFFull)), rowMax = __longlong_as_double(static_cast<long long>(keys[2 * set + 1]));
                // no usable input coefficient: an identically-zero (or redundant) row takes no pivot; anything else cannot be met by this knot's inputs
                int j = key ? 255 - static_cast<int>(key & 0xFFull) : -1;
                if (rowMax <= 1e-12 * __longlong_as_double(static_cast<long long>(rowScale[i]))) {
                    j = -1;  // rounding noise of the row's original entries: redundant (same rule as the single-wavefront job above)
                } else {
                    if (j >= 0 && best <= 1e-12 * rowMax) j = -2;
                    if (j == -1 && rowMax > 0.0) j = -2;
                }
                if (lane == 0) {
                    pivCol[i] = j;
                    if (j >= 0) used[j] = 1;  // (readers of this step exclude j themselves; later steps see the flag behind the barrier)
                    keys[2 * ((i + 2) % 3)] = 0ull;
                    keys[2 * ((i + 2) % 3) + 1] = 0ull;
                }
                if (j >= 0) {
                    double* dst = T[cur ^ 1];
                    const int J = nz + j;
                    const double rpiv = 1.0 / src[i * ld + J];
                    // (wavefront -> rows, lane -> column: the scaled pivot-row entry of a column is computed once per lane, a row's multiplier is one broadcast
                    // read, and no index is decoded -- a step is 4 short iterations for 16 rows; the flat index space cost ~35 instructions per item)
                    for (int c = wl; c < ld; c += 64) {
                        const double p = c == J ? 1.0 : src[i * ld + c] * rpiv;  // (the pivot exactly 1, its column exactly 0 elsewhere)
                        for (int r = wave; r < a.ne; r += waves) {
                            const double v = r == i ? p : (c == J ? 0.0 : src[r * ld + c] - src[r * ld + J] * p);
                            dst[r * ld + c] = v;
                            if (r == i + 1) offer(nextSet, c, v, j);
                        }
                    }
                    cur ^= 1;
                } else if (i + 1 < a.ne) {
                    for (int c = lane; c < ld; c += lanes) offer(nextSet, c, src[(i + 1) * ld + c], -1);
                }
                __syncthreads();
            }
            if (cur) {  // the reduced rows back where the substitution and the output expect them
                for (int idx = lane; idx < a.ne * ld; idx += lanes) Ed[idx] = T[1][idx];
                __syncthreads();
            }
        }
        UNGAR_SHOOTING_MARK();  // 4: Gauss-Jordan
        double* V = vbuf + 6;  // pivots x nd (the second tableau is dead)
        // ---- substitute u_j = -(G_i . [z; u] + g0_i) for ALL pivot rows at once (the reduced rows have zeros in each other's pivot columns):
        //   W'[a][c] = W[a][c] - sum_i (W[a][J_i] G_i[c] + G_i[a] W[J_i][c]) + sum_i G_i[a] V_i[c],   V_i[c] = sum_i' W[J_i][J_i'] G_i'[c]
        // for a, c outside the pivot set (in place: only pivot rows / columns are read besides the entry itself); likewise w, [A|B], b.
        // On the FP64 matrix cores, over ALL ne rows t (a row without a pivot contributes nothing: its operands are masked): every entry of the flat version
        // was a dependent chain of 16 x 3 multiply-adds behind index lookups -- 33-58 k of a node's ~120 k cycles; a 16 x 16 tile is 2 x ceil(ne / 4) matrix
        // instructions.  v_mfma_f64_16x16x4_f64: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[(lane >> 4) + 4 r][lane & 15] in element r.
        if (!specialised) {
            buildList(a.ne, [&](int r) { return pivCol[r] >= 0; });
            __syncthreads();
        }
        const int pivots = elim == 2 ? 0 : *listSize;  // (2: measurement only -- rows reduced, substitution skipped)
        using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
        const int li = wl & 15, lk = wl >> 4, KS = pivots > 0 ? (a.ne + 3) >> 2 : 0, TD = (nd + 15) >> 4, TE = (a.ne + 15) >> 4, TZ = (nz + 15) >> 4;
        auto pivotColumn = [&](int t) { return t < a.ne ? pivCol[t] : -1; };  // input the row t was solved for, or < 0
        // the k-steps of this lane's operand rows, looked up once (the same for every tile): up to 64 equality rows (the C ABI's bound)
        constexpr int kMaxSteps = 16, kGroup = UNGAR_ASSEMBLE_K_GROUP;
        int stepPivot[kMaxSteps], stepRow[kMaxSteps];
#pragma unroll
        for (int ks = 0; ks < kMaxSteps; ++ks) {
            const int t = 4 * ks + lk;
            stepPivot[ks] = ks < KS ? pivotColumn(t) : -1;
            stepRow[ks] = t < a.ne ? t : a.ne - 1;
        }
        // V = W_JJ G  (ne x nd)
        for (int tile = wave; tile < TE * TD; tile += waves) {
            const int ti = tile / TD, tj = tile - ti * TD;
            const int rowA = 16 * ti + li, colB = 16 * tj + li, cB = colB < nd ? colB : nd - 1;
            const int jA = pivotColumn(rowA);
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            // (k-steps in groups of four: the operands of a group are requested together and the matrix instructions follow -- one LDS round trip per group
            // instead of one per step; a step beyond KS inside a group has no pivot and multiplies by zero.  A `break` would keep the loop rolled and the
            // lookup tables in scratch.)
#pragma unroll
            for (int g = 0; g < kMaxSteps / kGroup; ++g) {
                if (g * kGroup < KS) {  // (uniform)
                    double av[kGroup], bv[kGroup];
#pragma unroll
                    for (int q = 0; q < kGroup; ++q) {
                        const int jt = stepPivot[g * kGroup + q], tc = stepRow[g * kGroup + q];
                        av[q] = Wd[sym(nz + (jA < 0 ? 0 : jA), nz + (jt < 0 ? 0 : jt))];
                        bv[q] = Ed[tc * ld + cB];
                    }
#pragma unroll
                    for (int q = 0; q < kGroup; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((jA >= 0 && stepPivot[g * kGroup + q] >= 0) ? av[q] : 0.0, bv[q], acc, 0, 0, 0);
                }
            }
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lk + 4 * r;
                if (row < a.ne && colB < nd) V[row * nd + colB] = acc[r];
            }
        }
        // the three vector updates: an entry's sum over the pivot rows is shared by the four lanes of a quad (DPP reduction) -- as one lane's loop it was a
        // chain of `pivots` multiply-adds, each behind three dependent index lookups
        // (the pivot rows a lane sums over -- list index t = part, part + 4, ... -- and their pivot inputs are looked up ONCE: inside the sums they were two
        // dependent LDS reads in front of every product; with them in registers a lane's products are independent and their operands travel together)
        constexpr int kHeldTerms = 4;
        int heldRow[kHeldTerms], heldColumn[kHeldTerms];
#pragma unroll
        for (int m = 0; m < kHeldTerms; ++m) {
            const int t = (lane & 3) + 4 * m;
            heldRow[m] = t < pivots ? list[t] : -1;
            heldColumn[m] = heldRow[m] >= 0 ? pivCol[heldRow[m]] : 0;
        }
        auto quadSum = [&](int count, auto term, auto store) {  // item i < count: store(i, sum_t term(i, row_t, input_t)); lanes 4 i .. 4 i + 3 share the sum
            for (int base = 0; base < count; base += lanes >> 2) {
                const int i = base + (lane >> 2), part = lane & 3;
                double sv = 0.0;
                if (i < count) {
#pragma unroll
                    for (int m = 0; m < kHeldTerms; ++m)
                        if (heldRow[m] >= 0) sv += term(i, heldRow[m], heldColumn[m]);
                    for (int t = part + 4 * kHeldTerms; t < pivots; t += 4) sv += term(i, list[t], pivCol[list[t]]);
                }
                sv += QuadPermute<0xB1>(sv);  // quad_perm [1, 0, 3, 2]
                sv += QuadPermute<0x4E>(sv);  // quad_perm [2, 3, 0, 1]
                if (i < count && part == 0) store(i, sv);
            }
        };
        quadSum(
            nd, [&](int c, int row, int input) { return Wd[sym(c, nz + input)] * Ed[row * ld + nd]; }, [&](int c, double sv) { prow[c] = gd[c] - sv; });  // w + W s,  s = -sum_i e_(J_i) g0_i
        quadSum(
            nz, [&](int r, int row, int input) { return ABd[r * nd + nz + input] * Ed[row * ld + nd]; }, [&](int r, double sv) { bd[r] -= sv; });
        __syncthreads();
        UNGAR_SHOOTING_MARK();  // (substitution: list, V = W_JJ G, w + W s, b)
        auto isPivot = [&](int c) { return c >= nz && used[c - nz] != 0; };
        // W' on and above the diagonal (mirrored) and [A|B]': in place -- the operands are pivot rows / columns, V and G, which nobody writes here
        const int upper = TD * (TD + 1) / 2;
        for (int tile = wave; tile < upper + TZ * TD; tile += waves) {
            const bool hessian = tile < upper;
            int ti = 0, tj = 0;
            if (hessian) {  // column-major upper triangle of tiles
                int idx = tile;
                while (idx > tj) {
                    idx -= tj + 1;
                    ++tj;
                }
                ti = idx;
            } else {
                ti = (tile - upper) / TD;
                tj = tile - upper - ti * TD;
            }
#ifdef UNGAR_SHOOTING_CLOCKS
            const unsigned long long tile0 = __builtin_amdgcn_s_memtime();
#endif
            const int rows = hessian ? nd : nz;
            const int rowA = 16 * ti + li, rA = rowA < rows ? rowA : rows - 1, rG = rowA < nd ? rowA : nd - 1;
            const int colB = 16 * tj + li, cB = colB < nd ? colB : nd - 1;
            f64x4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};  // (two accumulators: the two products of a k-step do not wait for each other)
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lk + 4 * r;
                const int rc = row < rows ? row : rows - 1;
                acc[r] = hessian ? Wd[sym(rc, cB)] : ABd[rc * nd + cB];  // (below the diagonal of a diagonal tile: the transposed entry, computed and not stored)
            }
#pragma unroll
            for (int g = 0; g < kMaxSteps / kGroup; ++g) {
                if (g * kGroup < KS) {  // (uniform; groups of four k-steps as above)
                    double a1[kGroup], b1[kGroup], a2[kGroup], b2[kGroup], b3[kGroup];
#pragma unroll
                    for (int q = 0; q < kGroup; ++q) {
                        const int jt = stepPivot[g * kGroup + q], tc = stepRow[g * kGroup + q], J = nz + (jt < 0 ? 0 : jt);
                        a1[q] = hessian ? Wd[sym(rA, J)] : ABd[rA * nd + J];
                        b1[q] = Ed[tc * ld + cB];
                        if (hessian) {
                            a2[q] = Ed[tc * ld + rG];
                            b2[q] = V[tc * nd + cB];
                            b3[q] = Wd[sym(J, cB)];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < kGroup; ++q) {
                        const bool pivot = stepPivot[g * kGroup + q] >= 0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pivot ? -a1[q] : 0.0, b1[q], acc, 0, 0, 0);  // - W[a][J_t] G_t[c]   /   - [A|B][r][J_t] G_t[c]
                        if (hessian) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pivot ? a2[q] : 0.0, b2[q] - b3[q], acc2, 0, 0, 0);  // + G_t[a] (V_t[c] - W[J_t][c])
                    }
                }
            }
#ifdef UNGAR_SHOOTING_CLOCKS
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long tile1 = __builtin_amdgcn_s_memtime();
#endif
            // (the pivot flags of the four rows are read before the first store: a store to W may alias them for the compiler, which then waits for every flag in turn)
            bool rowIsPivot[4];
            for (int r = 0; r < 4; ++r) rowIsPivot[r] = hessian && 16 * ti + lk + 4 * r < nd && isPivot(16 * ti + lk + 4 * r);
            if (colB < nd && !isPivot(colB)) {
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + lk + 4 * r;
                    acc[r] += acc2[r];
                    if (hessian) {
                        if (row <= colB && !rowIsPivot[r]) {
                            Wd[tri(row, colB)] = acc[r];
                        }
                    } else if (row < nz) {
                        ABd[row * nd + colB] = acc[r];
                    }
                }
            }
#ifdef UNGAR_SHOOTING_CLOCKS
            if (lane == 0) {
                const unsigned long long tile2 = __builtin_amdgcn_s_memtime();
                jobCycles[5] += tile1 - tile0;
                jobCycles[6] += tile2 - tile1;
                jobCycles[7] += 1;
            }
#endif
        }
        quadSum(
            nd, [&](int c, int row, int input) { return Ed[row * ld + c] * prow[nz + input]; }, [&](int c, double sv) { gd[c] = isPivot(c) ? 0.0 : prow[c] - sv; });
        __syncthreads();  // every read of a pivot row / column is done: they become the dummies' identity rows
        UNGAR_SHOOTING_MARK();  // (substitution: W', [A|B]', w')
        for (int t = wave; t < pivots; t += waves) {
            const int J = nz + pivCol[list[t]];
            for (int c = wl; c < nd; c += 64) {
                Wd[sym(J, c)] = c == J ? 1.0 : 0.0;
            }
            for (int r = wl; r < nz; r += 64) ABd[r * nd + J] = 0.0;
        }
        __syncthreads();
        for (int i = lane; i < a.ne; i += lanes) {
            a.pivots[stageOff * a.ne + i] = pivCol[i];
            a.er[stageOff * a.ne + i] = Ed[i * ld + nd];
        }
    }
    UNGAR_SHOOTING_MARK();  // 5: substitution (with elimination; else 3 again)
    double* W = a.W + nodeOff * nd * nd;
    for (int idx = lane; idx < nd * nd; idx += lanes) {
        const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv);
        if (r <= idx - r * nd) W[idx] = Wd[tri(r, idx - r * nd)];
    }
    double* w = a.w + nodeOff * nd;
    for (int c = lane; c < nd; c += lanes) w[c] = gd[c];
    if (stage) {
        double* AB = a.AB + stageOff * nz * nd;
        for (int idx = lane; idx < nz * nd; idx += lanes) AB[idx] = ABd[idx];
        double* bo = a.b + stageOff * nz;
        for (int i = lane; i < nz; i += lanes) bo[i] = bd[i];
        if (a.ne > 0) {
            double* E = a.E + stageOff * a.ne * nd;
            for (int idx = lane; idx < a.ne * nd; idx += lanes) {
                const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv);
                E[idx] = Ed[r * ld + (idx - r * nd)];
            }
        }
        if (k == 0) {
            const double* row0 = RowOf(a.rows, d, b, 0);
            for (int i = lane; i < nz; i += lanes) a.dz0[b * nz + i] = i < nc ? 0.0 : a.xm[b * nx + (i - nc)] - row0[i];
        }
    }
#ifdef UNGAR_SHOOTING_CLOCKS
    __syncthreads();
    UNGAR_SHOOTING_MARK();  // last: results written
    if (lane == 0 && (node == 1 || node == d.batch * (d.N + 1) / 2 + 1)) {
        unsigned long long dt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int m = 1; m < markCount && m <= 10; ++m) dt[m - 1] = marks[m] - marks[m - 1];
        // (one call per node: the lines of two nodes do not interleave)
        printf("[assemble clocks] node %lld (%u non-zero equality rows): %llu %llu %llu %llu %llu %llu %llu %llu %llu   jobs: elimination %llu, barrier terms %llu (wavefront 1 in all %llu), defect %llu %llu; wavefront 0: %llu tiles, operands + products %llu, epilogue %llu\n",
               node, static_cast<unsigned>(jobCycles[8]), dt[0], dt[1], dt[2], dt[3], dt[4], dt[5], dt[6], dt[7], dt[8], jobCycles[0], jobCycles[4], jobCycles[1], jobCycles[2], jobCycles[3], jobCycles[7], jobCycles[5], jobCycles[6]);
    }
#endif
}


/// ONE WAVEFRONT per node for stage problems WITHOUT equality rows whose differentiated row fits the lanes (nd + 1 <= 64: the quadrotor's 21, the RC car's 12) --
/// run-time sizes, the same ingredients as the kernel above: every sparse value requested as one batch of range-checked MUBUF loads before anything is computed
/// from them, dense LDS images (the packed upper triangle of W_e = [W w; w^T 0] and [A|B]_e = [A|B  b]) filled by scatter, the barrier terms as read-modify-writes
/// of one wavefront (LDS instructions of a wavefront execute in order: no barrier), each dense block written to memory once, coalesced.  The workgroup kernel's
/// generic sections did the same with a workgroup barrier after every inequality row and loads from global memory inside the row loop: 0.46 ms per
/// 4096 x 31 quadrotor nodes against 0.18 ms of compulsory traffic.
__global__ __launch_bounds__(64) void ShootingAssembleSmallKernel(const ShootingAssembleArgs a) {
    extern __shared__ double lds[];
    const ShootingDims& d = a.d;
    const long long node = blockIdx.x;
    const long long b = node / (d.N + 1);
    const int k = static_cast<int>(node - b * (d.N + 1));
    const int lane = static_cast<int>(threadIdx.x);
    const int nc = d.nc, nx = d.nx, nz = d.nz(), nd = d.nd(), nh1 = nd + 1;  // (column nd: the linear terms)
    const bool stage = k < d.N;
    const int nW = nh1 * (nh1 + 1) / 2, nAB = nz * nh1;
    double* R = lds;                    // packed upper triangle of W_e
    double* ABi = R + ((nW + 1) & ~1);  // nz x (nd + 1)
    double* d1 = ABi + ((nAB + 1) & ~1);
    double* d2 = d1 + a.nh;
    int* pairTable = reinterpret_cast<int*>(d2 + a.nh);  // 64
    auto tri = [nh1](int r, int c) { return ((r * (2 * nh1 + 1 - r)) >> 1) + (c - r); };  // r <= c < nd + 1
    auto fence = [] { asm volatile("" ::: "memory"); };
    const long long nodeOff = node, stageOff = b * d.N + k;
    constexpr int kSlotsH = 4, kSlotsF = 4, kSlotsC = 2;
    struct Raw {
        int r, c;
        double value;
        bool valid;
    };
    auto resourceOver = [](const void* base, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base ? bytes : 0, 0x00020000); };
    auto loadInt = [](__amdgpu_buffer_rsrc_t rs, int byteOffset) { return __builtin_amdgcn_raw_buffer_load_b32(rs, byteOffset, 0, 0); };
    auto loadDouble = [](__amdgpu_buffer_rsrc_t rs, int byteOffset) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, byteOffset, 0, 0)); };
    auto request = [&](const StagePattern& pattern, const double* values, bool wanted, int slot) {
        const int nnz = wanted && values ? pattern.nnz : 0, e = lane + 64 * slot;
        Raw f;
        f.r = loadInt(resourceOver(pattern.rows, nnz * 4), e * 4);
        f.c = loadInt(resourceOver(pattern.cols, nnz * 4), e * 4);
        f.value = loadDouble(resourceOver(values ? values + nodeOff * pattern.nnz : nullptr, nnz * 8), e * 8);
        f.valid = e < nnz;
        return f;
    };
    Raw rH[kSlotsH], rF[kSlotsF], rC[kSlotsC];
#pragma unroll
    for (int s = 0; s < kSlotsH; ++s) rH[s] = request(a.pH, a.lH, true, s);
    const Raw rG = request(a.pg, a.lg, true, 0);
    const double hMine = loadDouble(resourceOver(a.h ? a.h + nodeOff * a.nh : nullptr, stage ? a.nh * 8 : 0), lane * 8);
    const Raw rI = request(a.ph, a.hJ, stage && a.nh > 0, 0);
#pragma unroll
    for (int s = 0; s < kSlotsF; ++s) rF[s] = request(a.pf, a.fJ, stage, s);
#pragma unroll
    for (int s = 0; s < kSlotsC; ++s) rC[s] = request(a.pc, a.cJ, stage && !d.carryInputs, s);
    const double fMine = loadDouble(resourceOver(a.f + nodeOff * nx, stage ? nx * 8 : 0), (lane - nc) * 8);
    const double nextMine = loadDouble(resourceOver(RowOf(a.rows, d, b, stage ? k + 1 : k), stage ? nz * 8 : 0), lane * 8);
    const double xmMine = loadDouble(resourceOver(a.xm + b * nx, k == 0 ? nx * 8 : 0), (lane - nc) * 8);
    const double row0Mine = loadDouble(resourceOver(RowOf(a.rows, d, b, 0), k == 0 ? nz * 8 : 0), lane * 8);
    // ---- images
    const int zeroed = ((nW + 1) & ~1) + ((nAB + 1) & ~1) + 2 * a.nh;
    for (int i = lane; i < zeroed; i += 64) lds[i] = 0.0;
    fence();
#pragma unroll
    for (int s = 0; s < kSlotsH; ++s)
        if (rH[s].valid && rH[s].r <= rH[s].c) R[tri(rH[s].r, rH[s].c)] = rH[s].value;  // (Function::Hessian's pattern is upper triangular; anything below the diagonal is ignored)
    if (rG.valid) R[tri(rG.c, nd)] = rG.value;
    if (stage) {
#pragma unroll
        for (int s = 0; s < kSlotsF; ++s)
            if (rF[s].valid) ABi[(nc + rF[s].r) * nh1 + nc + rF[s].c] = rF[s].value;
        if (d.carryInputs) {
            if (lane < nc) ABi[lane * nh1 + nz + lane] = 1.0;
        } else {
#pragma unroll
            for (int s = 0; s < kSlotsC; ++s)
                if (rC[s].valid) ABi[rC[s].r * nh1 + nc + rC[s].c] = rC[s].value;
        }
        if (lane >= nc && lane < nz) ABi[lane * nh1 + nd] = fMine - nextMine;
        if (lane < a.nh) {
            d1[lane] = BarrierD1(a.barrier, -hMine);
            d2[lane] = BarrierD2(a.barrier, -hMine);
        }
    }
    fence();
    // ---- barrier terms  W += J_h^T diag(b''(-h)) J_h,  w -= J_h^T b'(-h)
    if (stage && a.nh > 0) WaveBarrierTerms(R, tri, nd, a.nh, d1, d2, pairTable, lane, a.ph.nnz, rI.valid ? rI.r : -1, rI.valid ? rI.c : 0, rI.value, a.ph.cols, a.hJ + nodeOff * a.ph.nnz);
    if (lane >= nc && lane < (stage ? nd : nz)) R[tri(lane, lane)] += a.regularization;  // the reference's 1e-6 I over its decision variables (soft_sqp.hpp:149-151)
    fence();
    // ---- results
    const float ndInv = 1.0f / static_cast<float>(nd);
    double* W = a.W + nodeOff * nd * nd;
    for (int idx = lane; idx < nd * nd; idx += 64) {
        const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv), c = idx - r * nd;
        if (r <= c) W[idx] = R[tri(r, c)];
    }
    if (lane < nd) a.w[nodeOff * nd + lane] = R[tri(lane, nd)];
    if (stage) {
        double* AB = a.AB + stageOff * nz * nd;
        for (int idx = lane; idx < nz * nd; idx += 64) {
            const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv);
            AB[idx] = ABi[r * nh1 + (idx - r * nd)];
        }
        if (lane < nz) a.b[stageOff * nz + lane] = ABi[lane * nh1 + nd];
        if (k == 0 && lane < nz) a.dz0[b * nz + lane] = lane >= nc ? xmMine - row0Mine : 0.0;
    }
}

/// SEVERAL NODES per wavefront for the narrow stage problems without equality rows: G = 16 lanes per node (nd + 1 <= G: the RC car's 11 columns),
/// 64 / G nodes per wavefront, each group with its own LDS images.  The kernel above is bound by the number of instructions it issues per node
/// (~2.5 k SIMD cycles per node whatever the size: 127 us per 4096 x 31 RC-car nodes whose results are 30 us of traffic) and most of its lanes carry nothing
/// when a row has 11 columns; here one instruction works on four nodes (127 -> 52 us).  Same arithmetic in the same order as the kernel above (bitwise the same results):
/// sparse values requested first, scatter into the packed image of W_e and the image of [A|B]_e, barrier terms row by row (one lane per entry of the inequality
/// Jacobian, its partners one after the other), regularisation, every dense block written once.
template <int G, int SLOTS>  // SLOTS: G-entry batches of a sparse pattern the lanes hold in registers (2, 4, 8)
__global__ __launch_bounds__(64) void ShootingAssemblePackedKernel(const ShootingAssembleArgs a) {
    extern __shared__ double lds[];
    constexpr int kNodes = 64 / G, kSlots = SLOTS, kCarrySlots = SLOTS / 2;
    const ShootingDims& d = a.d;
    const int lane = static_cast<int>(threadIdx.x), gl = lane & (G - 1), sub = lane / G;
    const long long nodes = static_cast<long long>(d.batch) * (d.N + 1), nodeRaw = static_cast<long long>(blockIdx.x) * kNodes + sub;
    const bool live = nodeRaw < nodes;  // (a group beyond the last node works on the last node once more and stores nothing)
    const long long node = live ? nodeRaw : nodes - 1;
    const long long b = node / (d.N + 1);
    const int k = static_cast<int>(node - b * (d.N + 1));
    const int nc = d.nc, nx = d.nx, nz = d.nz(), nd = d.nd(), nh1 = nd + 1, nh = a.nh;  // (column nd: the linear terms)
    const bool stage = k < d.N;
    const int nW = nh1 * (nh1 + 1) / 2, nAB = nz * nh1;
    const int perNode = ((nW + 1) & ~1) + ((nAB + 1) & ~1) + 2 * nh;
    double* R = lds + sub * perNode;    // packed upper triangle of W_e
    double* ABi = R + ((nW + 1) & ~1);  // nz x (nd + 1)
    double* d1 = ABi + ((nAB + 1) & ~1);
    double* d2 = d1 + nh;
    auto tri = [nh1](int r, int c) { return ((r * (2 * nh1 + 1 - r)) >> 1) + (c - r); };  // r <= c < nd + 1
    auto fence = [] { asm volatile("" ::: "memory"); };
    const long long stageOff = b * d.N + k;
    struct Raw {
        int r, c;  // r < 0: nothing
        double value;
    };
    auto request = [&](const StagePattern& pattern, const double* values, bool wanted, int slot) {
        Raw f{-1, 0, 0.0};
        const int e = gl + G * slot;
        if (wanted && values && e < pattern.nnz) {
            f.r = pattern.rows[e];
            f.c = pattern.cols[e];
            f.value = values[node * pattern.nnz + e];
        }
        return f;
    };
    Raw rH[kSlots], rF[kSlots], rC[kCarrySlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) rH[s] = request(a.pH, a.lH, true, s);
    const Raw rG = request(a.pg, a.lg, true, 0);
    const Raw rI = request(a.ph, a.hJ, stage && nh > 0, 0);
    const double hMine = stage && a.h && gl < nh ? a.h[node * nh + gl] : 0.0;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) rF[s] = request(a.pf, a.fJ, stage, s);
#pragma unroll
    for (int s = 0; s < kCarrySlots; ++s) rC[s] = request(a.pc, a.cJ, stage && !d.carryInputs, s);
    const bool stateLane = gl >= nc && gl < nz;
    const double fMine = stage && stateLane ? a.f[node * nx + (gl - nc)] : 0.0;
    const double nextMine = stage && gl < nz ? RowOf(a.rows, d, b, k + 1)[gl] : 0.0;
    const double xmMine = k == 0 && stateLane ? a.xm[b * nx + (gl - nc)] : 0.0;
    const double row0Mine = k == 0 && gl < nz ? RowOf(a.rows, d, b, 0)[gl] : 0.0;
    // ---- images
    for (int i = gl; i < perNode; i += G) R[i] = 0.0;
    fence();
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
        if (rH[s].r >= 0 && rH[s].r <= rH[s].c) R[tri(rH[s].r, rH[s].c)] = rH[s].value;  // (Function::Hessian's pattern is upper triangular; anything below the diagonal is ignored)
    if (rG.r >= 0) R[tri(rG.c, nd)] = rG.value;
    if (stage) {
#pragma unroll
        for (int s = 0; s < kSlots; ++s)
            if (rF[s].r >= 0) ABi[(nc + rF[s].r) * nh1 + nc + rF[s].c] = rF[s].value;
        if (d.carryInputs) {
            if (gl < nc) ABi[gl * nh1 + nz + gl] = 1.0;
        } else {
#pragma unroll
            for (int s = 0; s < kCarrySlots; ++s)
                if (rC[s].r >= 0) ABi[rC[s].r * nh1 + nc + rC[s].c] = rC[s].value;
        }
        if (stateLane) ABi[gl * nh1 + nd] = fMine - nextMine;
        if (gl < nh) {
            d1[gl] = BarrierD1(a.barrier, -hMine);
            d2[gl] = BarrierD2(a.barrier, -hMine);
        }
    }
    fence();
    // ---- barrier terms  W += J_h^T diag(b''(-h)) J_h,  w -= J_h^T b'(-h):  lane e of a group holds entry e of its node's inequality Jacobian (the entries of a row
    // are consecutive, columns ascending) and owns the pairs (e, e + q) of its row; row by row, in order (rows share targets; LDS instructions of a wavefront execute in order)
    if (nh > 0) {
        const int myRow = rI.r, nnz = a.ph.nnz;
        const int nextRow = __shfl_down(myRow, 1);
        const unsigned long long rowEnds = __ballot(myRow >= 0 && (gl == nnz - 1 || nextRow != myRow));
        const int partners = myRow >= 0 ? __ffsll(static_cast<unsigned long long>(rowEnds >> lane)) : 0;  // entries of the same row from this one on (never beyond the group: its last entry ends a row)
        const double d1Mine = myRow >= 0 ? d1[myRow] : 0.0, d2Mine = myRow >= 0 ? d2[myRow] : 0.0;
        for (int j = 0; j < nh; ++j) {
            if (myRow == j) {
                const int gTarget = tri(rI.c, nd);
                R[gTarget] = __builtin_fma(-d1Mine, rI.value, R[gTarget]);
            }
            for (int q = 0; __ballot(myRow == j && q < partners) != 0ull; ++q) {
                const int c2 = __shfl(rI.c, lane + q);
                const double v2 = __shfl(rI.value, lane + q);
                if (myRow == j && q < partners) {
                    const int target = tri(rI.c, c2);
                    R[target] = __builtin_fma(d2Mine * rI.value, v2, R[target]);
                }
            }
            fence();
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (gl >= nc && gl < (stage ? nd : nz)) R[tri(gl, gl)] += a.regularization;  // the reference's 1e-6 I over its decision variables (soft_sqp.hpp:149-151)
    fence();
    // ---- results
    if (!live) return;
    const float ndInv = 1.0f / static_cast<float>(nd);
    double* W = a.W + node * nd * nd;
    for (int idx = gl; idx < nd * nd; idx += G) {
        const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv), c = idx - r * nd;
        if (r <= c) W[idx] = R[tri(r, c)];
    }
    if (gl < nd) a.w[node * nd + gl] = R[tri(gl, nd)];
    if (stage) {
        double* AB = a.AB + stageOff * nz * nd;
        for (int idx = gl; idx < nz * nd; idx += G) {
            const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv);
            AB[idx] = ABi[r * nh1 + (idx - r * nd)];
        }
        if (gl < nz) a.b[stageOff * nz + gl] = ABi[gl * nh1 + nd];
        if (k == 0 && gl < nz) a.dz0[b * nz + gl] = gl >= nc ? xmMine - row0Mine : 0.0;
    }
}

/// FOUR lanes per (stage node, reduced row): lane q of a quad sums the columns q, q + 4, ... of its row, the quad adds up with two DPP moves.  (One lane per row
/// read its 49 coefficients one after the other, 392 bytes from its neighbour's: 64 separate 8-byte segments per load instruction, 0.29 ms per 4096 x 30 x 16
/// quadruped rows; four adjacent lanes read 32 consecutive bytes.)
__global__ __launch_bounds__(256) void ShootingRecoverKernel(const ShootingRecoverArgs a) {
    const ShootingDims& d = a.d;
    const long long lane = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x, idx = lane >> 2;
    const int q = static_cast<int>(lane & 3);
    const bool mine = idx < d.batch * d.N * a.ne;
    const long long row0 = mine ? idx : 0;  // (lanes beyond the last row keep the quad's DPP moves company and store nothing)
    const long long node = row0 / a.ne;
    const int nz = d.nz(), nd = d.nd();
    const long long b = node / d.N;
    const int k = static_cast<int>(node - b * d.N);
    const int j = mine ? a.pivots[row0] : -1;
    if (j == -2 && q == 0 && a.status) a.status[b] = -(k + 1);  // (several lanes may write: any of them is a true report)
    double acc = 0.0;
    if (j >= 0) {
        const double* row = a.E + row0 * nd;
        const double* dz = a.dZ + (b * (d.N + 1) + k) * nz;
        const double* du = a.dU + node * d.nu;
        if (q == 0) acc = a.er[row0];
        for (int c = q; c < nd; c += 4) {
            const double coefficient = row[c];
            if (c < nz) acc += coefficient * dz[c];
            else if (c - nz != j && coefficient != 0.0) acc += coefficient * du[c - nz];  // the other pivot inputs have exact zeros here: never read while they are written
        }
    }
    acc += QuadPermute<0xB1>(acc);  // quad_perm [1, 0, 3, 2]
    acc += QuadPermute<0x4E>(acc);  // quad_perm [2, 3, 0, 1]
    if (j >= 0 && q == 0) a.dU[node * d.nu + j] = -acc;
}

__global__ __launch_bounds__(kBlock) void ShootingMeritKernel(const ShootingMeritArgs a) {
    const ShootingDims& d = a.d;
    const long long s = blockIdx.x;
    if (s >= d.batch) return;
    const long long b = a.period > 0 ? (a.instances ? a.instances[s % a.period] : s % a.period) : s;
    const int lane = static_cast<int>(threadIdx.x), nz = d.nz(), nd = d.nd(), nc = d.nc, nx = d.nx, N = d.N;
    double g2 = 0.0, obj = 0.0, bar = 0.0, slope = 0.0;
    auto state = [&](int k, int i) {  // x_k[i] of (stacked) instance s
        return a.rowsStride > 0 ? a.rows[(nc + i) * a.rowsStride + s * (N + 1) + k] : RowOf(a.rows, d, s, k)[nc + i];
    };
    for (int i = lane; i < nx; i += kBlock) {
        const double r = state(0, i) - a.xm[b * nx + i];
        g2 += r * r;
    }
    for (int idx = lane; idx < N * nx; idx += kBlock) {
        const int k = idx / nx, i = idx - k * nx;
        const double r = state(k + 1, i) - a.f[(s * (N + 1) + k) * (a.valueStride > 0 ? a.valueStride : nx) + i];
        g2 += r * r;
    }
    if (a.e)
        for (int idx = lane; idx < N * a.ne; idx += kBlock) {
            const int k = idx / a.ne, j = idx - k * a.ne;
            const double r = a.e[(s * (N + 1) + k) * (a.valueStride > 0 ? a.valueStride : a.ne) + j];
            g2 += r * r;
        }
    for (int k = lane; k <= N; k += kBlock) obj += a.l[(s * (N + 1) + k) * (a.valueStride > 0 ? a.valueStride : 1)];
    if (a.h)
        for (int idx = lane; idx < N * a.nh; idx += kBlock) {
            const int k = idx / a.nh, j = idx - k * a.nh;
            bar += Barrier(a.barrier, -a.h[(s * (N + 1) + k) * (a.valueStride > 0 ? a.valueStride : a.nh) + j]);
        }
    const bool wantSlope = a.lg && a.dZ && a.slope;
    if (wantSlope)
        for (int idx = lane; idx < (N + 1) * a.pg.nnz; idx += kBlock) {
            const int k = idx / a.pg.nnz, e = idx - k * a.pg.nnz, c = a.pg.cols[e];
            double step = 0.0;
            if (c < nz) step = a.dZ[(b * (N + 1) + k) * nz + c];
            else if (k < N) step = a.dU[(b * N + k) * d.nu + (c - nz)];
            slope += a.lg[(s * (N + 1) + k) * a.pg.nnz + e] * step;
        }
    g2 = WaveSum(g2);
    obj = WaveSum(obj);
    bar = WaveSum(bar);
    slope = WaveSum(slope);
    if (lane == 0) {
        a.theta[s] = a.violationMultiplier * sqrt(g2);
        a.phi[s] = obj + bar;
        if (a.objective) a.objective[s] = obj;
        if (wantSlope) a.slope[s] = slope;
    }
}

__global__ __launch_bounds__(256) void ShootingTrialKernel(const ShootingTrialArgs a) {
    const ShootingDims& d = a.d;
    const long long node = blockIdx.x;  // stacked node: (candidate c, instance b, knot k)
    const long long s = node / (d.N + 1);
    const int k = static_cast<int>(node - s * (d.N + 1));
    const long long stacked = a.listed > 0 ? a.listed : d.batch;
    if (s >= a.candidates * stacked) return;
    const long long b = a.listed > 0 ? a.instances[s % stacked] : s % stacked;
    const double alpha = a.alphas[s / stacked];
    const int nz = d.nz(), nd = d.nd(), nv = d.nv(), nc = d.nc, N = d.N;
    const double* row = RowOf(a.rows, d, b, k);
    double* out = a.trial + node * nv;
    for (int j = static_cast<int>(threadIdx.x); j < nv; j += static_cast<int>(blockDim.x)) {
        double v = row[j];
        if (alpha == 0.0) {  // (a step of length 0 is the row itself, whatever the direction holds)
            if (j < nc && d.carryInputs && k > 0) v = RowOf(a.rows, d, b, k - 1)[nz + j];
        } else if (j < nc && d.carryInputs) {
            if (k > 0) {  // the trial input of the previous knot, the same bits the trial row k - 1 holds
                const double* prev = RowOf(a.rows, d, b, k - 1);
                v = fma(alpha, a.dU[(b * N + (k - 1)) * d.nu + j], prev[nz + j]);
            }
        } else if (j < nz) {
            v = fma(alpha, a.dZ[(b * (N + 1) + k) * nz + j], v);
        } else if (j < nd && k < N) {
            v = fma(alpha, a.dU[(b * N + k) * d.nu + (j - nz)], v);
        }
        out[j] = v;
    }
}

/// carry_inputs problems: the carried slots of row k + 1 <- the inputs of row k (k < N), in place; one lane per (instance, knot, input).  Row 0's carried slots
/// are the caller's.  (What BatchedSoftSQPOptimizer::RefreshCarried needs after the rows were written from outside: a zero-length step through the trial
/// kernel did the same but cleared the last search direction and went through the trial buffer.)
__global__ __launch_bounds__(256) void ShootingRefreshCarriedInputsKernel(const ShootingDims d, double* rows) {
    const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= d.batch * d.N * d.nu) return;
    const long long node = idx / d.nu;
    const int i = static_cast<int>(idx - node * d.nu);
    const long long b = node / d.N;
    const int k = static_cast<int>(node - b * d.N);
    double* row = rows + (b * (d.N + 1) + k) * static_cast<long long>(d.nv());
    row[d.nv() + i] = row[d.nz() + i];  // (reads inputs, writes carried slots of the next row: disjoint elements, no ordering needed)
}

/// The same trial rows UNIT-FASTEST: a workgroup takes 64 consecutive nodes (instance, knot) of the (listed) instances and writes their elements for ALL the
/// candidates of the call -- row and direction are read once, up to 16 elements at a time, into two LDS tiles (the (node, element) pairs of a chunk dealt to the
/// lanes element-fastest: a chunk of any width keeps every lane busy and reads each row segment once, contiguously), then every candidate's values leave as
/// fma(alpha_c, direction, value), 64 consecutive nodes per element: coalesced.  The 64 nodes are decomposed into (instance, knot) once per workgroup.
/// (History: one workgroup per 64 STACKED nodes (candidate, instance, knot) read row and direction again for every candidate and lived for ~2.5 loads per lane
/// behind a prologue of 64-bit divisions: 95 us for the 142 MB of 14 candidates x 4096 RC-car instances, 1.5 TB/s.  Before that: 32 lanes per node whatever the
/// width left two thirds of the lanes idle on a 10-element window; one lane per (element, node) with the element outermost over the whole launch re-read every
/// row line once per element from L2.)
__global__ __launch_bounds__(256) void ShootingTrialUnitFastestKernel(const ShootingTrialArgs a) {
    constexpr int kLd = 73, kChunk = 16;  // (9 mod 32: the pairs a wavefront writes in one instruction fall into different banks for the widths that occur)
    __shared__ double value[kChunk * kLd], direction[kChunk * kLd];
    __shared__ long long nodeInstance[64];
    __shared__ int nodeKnot[64];
    const ShootingDims& d = a.d;
    const long long stacked = a.listed > 0 ? a.listed : d.batch;
    const long long nodes = stacked * (d.N + 1);  // nodes of ONE candidate; candidate c of node i at stacked node c * nodes + i
    const long long node0 = static_cast<long long>(blockIdx.x) * 64;
    const int t = static_cast<int>(threadIdx.x), nz = d.nz(), nd = d.nd(), nv = d.nv(), nc = d.nc, N = d.N;
    if (t < 64 && node0 + t < nodes) {
        const long long node = node0 + t, i = node / (N + 1);
        nodeKnot[t] = static_cast<int>(node - i * (N + 1));
        nodeInstance[t] = a.listed > 0 ? a.instances[i] : i;
    }
    __syncthreads();
    const int first = a.first, end = a.elements > 0 ? a.first + a.elements : nv;  // window of row elements this launch writes
    const int live = nodes - node0 < 64 ? static_cast<int>(nodes - node0) : 64;
    const int chunks = (end - first + kChunk - 1) / kChunk, perChunk = chunks > 0 ? (end - first + chunks - 1) / chunks : kChunk;  // (49 elements: 13 + 12 + 12 + 12, not 16 + 16 + 16 + 1)
    for (int j0 = first; j0 < end; j0 += perChunk) {
        const int width = end - j0 < perChunk ? end - j0 : perChunk;
        // (64 nodes x at most 16 elements over 256 lanes: four trips, all their requests in front of the first use -- a trip per round trip to memory made a
        // pass take 10 us with the whole device doing the same)
        double v[4], step[4];
        bool has[4];
        const int pairs = live * width;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = t + 256 * u < pairs ? t + 256 * u : pairs - 1;
            const int nl = f / width, jj = f - nl * width, j = j0 + jj, k = nodeKnot[nl];
            const long long b = nodeInstance[nl];
            const bool carried = j < nc && d.carryInputs;
            const double* from = carried && k > 0 ? RowOf(a.rows, d, b, k - 1) + nz + j : RowOf(a.rows, d, b, k) + j;
            const double* along = nullptr;  // (parameters, the last knot's input slots, the carried slots of knot 0: copied)
            if (carried) {
                if (k > 0) along = a.dU + (b * N + (k - 1)) * d.nu + j;  // the trial input of the previous knot, the same bits the trial row k - 1 holds
            } else if (j < nz) {
                along = a.dZ + (b * (N + 1) + k) * nz + j;
            } else if (j < nd && k < N) {
                along = a.dU + (b * N + k) * d.nu + (j - nz);
            }
            v[u] = *from;
            step[u] = *(along ? along : from);  // (one request whatever the element is)
            has[u] = along != nullptr;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = t + 256 * u;
            if (f < pairs) {
                const int nl = f / width, jj = f - nl * width;
                value[jj * kLd + nl] = v[u];
                direction[jj * kLd + nl] = has[u] ? step[u] : 0.0;
            }
        }
        __syncthreads();
        // this workgroup's candidates: blockIdx.y, blockIdx.y + gridDim.y, ... (a launch over few nodes -- the listed instances of a later stage -- spreads its
        // candidates over workgroups instead: one workgroup writing 12 candidates one after the other was 27 us of nothing but its own latency)
        double tv[4], ts[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = t + 256 * u < 64 * width ? t + 256 * u : 0;
            tv[u] = value[(f >> 6) * kLd + (f & 63)];
            ts[u] = direction[(f >> 6) * kLd + (f & 63)];
        }
        for (int c = static_cast<int>(blockIdx.y); c < a.candidates; c += static_cast<int>(gridDim.y)) {
            const double alpha = a.alphas[c];
            double* out = a.trial + c * nodes + node0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = t + 256 * u, row = f >> 6, nl = f & 63;
                // (a step of length 0 is the row itself, whatever the direction holds -- the unit-fastest image of the current rows)
                // (no direction: tile entry 0, the value as it is -- its sign of zero included)
                if (f < 64 * width && nl < live) out[(j0 + row - first) * a.trialStride + nl] = alpha == 0.0 || ts[u] == 0.0 ? tv[u] : fma(alpha, ts[u], tv[u]);
            }
        }
        __syncthreads();
    }
}

/// One wavefront per (listed) instance, kSelectWaves of them in a workgroup.  The instances a stage leaves unresolved are appended to the next stage's list at
/// positions handed out by ONE counter: the workgroup adds its count once and deals the positions out itself -- one atomic per instance on one address took 35 of
/// the kernel's 47 us when nine tenths of 4096 RC-car instances went on to the second stage (profiles/r06z_batched_rc_car_kernel_stats.csv, kernel trace).
constexpr int kSelectWaves = 8;
__global__ __launch_bounds__(kBlock * kSelectWaves) void ShootingSelectKernel(const ShootingSelectArgs a) {
    __shared__ int pending[kSelectWaves];
    __shared__ int firstPosition;
    const ShootingDims& d = a.d;
    const int wave = static_cast<int>(threadIdx.x) / kBlock, lane = static_cast<int>(threadIdx.x) % kBlock;
    const long long slot = static_cast<long long>(blockIdx.x) * kSelectWaves + wave, stacked = a.listed > 0 ? a.listed : d.batch;  // stacked point (c, slot) at c * stacked + slot
    const bool collects = !a.last && a.unresolved;  // (uniform: this stage lists the instances it leaves unresolved)
    long long b = 0;
    int chosen = -1;
    bool unresolved = false;
    if (slot < stacked) {  // (uniform over the wavefront)
        b = a.listed > 0 ? a.instances[slot] : slot;
        if (a.active && a.active[b] == 0) {
            if (lane == 0 && a.first) a.accepted[b] = 0.0;
        } else if (a.first || a.accepted[b] == 0.0) {  // (otherwise: took its step in an earlier stage of this search)
            const double theta = a.theta0[b], phi = a.phi0[b], slope = a.slope[b];
            const bool solved = !a.status || a.status[b] == 0;
            for (int c = 0; solved && c < a.candidates && chosen < 0; ++c)
                if (StepAcceptable(theta, phi, slope, a.thetaT[c * stacked + slot], a.phiT[c * stacked + slot], a.alphas[c], a.thetaMin, a.thetaMax, a.eta, a.gammaPhi, a.gammaTheta)) chosen = c;
            if (chosen < 0) {
                if (lane == 0) a.accepted[b] = 0.0;
                if (a.last || !solved) {
                    if (lane == 0 && a.active) a.active[b] = 0;  // the reference's `break` on a rejected step (soft_sqp.hpp:88-90)
                } else {
                    unresolved = true;
                }
            }
        }
    }
    if (collects) {  // (any order: the instances are independent)
        if (lane == 0) pending[wave] = unresolved ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int count = 0;
            for (int w = 0; w < kSelectWaves; ++w) count += pending[w];
            firstPosition = count ? atomicAdd(a.unresolved, count) : 0;
        }
        __syncthreads();
        if (unresolved && lane == 0 && a.nextInstances) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += pending[w];
            a.nextInstances[firstPosition + before] = static_cast<int>(b);
        }
    }
    if (chosen < 0) return;
    const long long from = chosen * stacked + slot;
    const int nd = d.nd(), nv = d.nv();
    // the chosen trial row back into the rows, four trips' loads in front of their stores (rows and trial may alias as far as the compiler knows: written
    // plainly, every trip waits for its own round trip -- 24 in a row for the quadruped's 31 x 49 variables)
    const int total = (d.N + 1) * nd;
    for (int base = lane; base < total; base += 4 * kBlock) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * kBlock < total ? base + u * kBlock : total - 1, k = idx / nd, j = idx - k * nd;
            v[u] = a.trialStride > 0 ? a.trial[j * a.trialStride + from * (d.N + 1) + k] : a.trial[(from * (d.N + 1) + k) * nv + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * kBlock, k = idx / nd, j = idx - k * nd;
            if (idx < total) a.rows[(b * (d.N + 1) + k) * nv + j] = v[u];
        }
    }
    if (lane == 0) {
        a.accepted[b] = a.alphas[chosen];
        const double difference = a.objectiveT[from] - a.objective0[b];
        if (a.active && difference < 0.0 && fabs(difference) < 1e-6) a.active[b] = 0;  // convergence criterion (soft_sqp.hpp:92-99)
    }
}

}  // namespace
}  // namespace ungar_amd::kernels

using namespace ungar_amd::kernels;

namespace {
/// The one-wavefront kernel for this shape, if the node fits its bounds (0: launched; -1: not applicable, the caller takes the workgroup kernel).
template <int NZ, int NU, int NE>
int LaunchAssembleWave(const ShootingAssembleArgs* a, void* stream) {
    const ShootingDims& d = a->d;
    if (d.nz() != NZ || d.nu != NU || a->ne != NE || a->eliminate != 1) return -1;
    if (a->nh > 64 || (a->nh > 0 && a->ph.nnz > 64) || a->pH.nnz > 256 || a->pg.nnz > 64 || a->pf.nnz > 256 || a->pe.nnz > 256 || (!d.carryInputs && a->pc.nnz > 128)) return -1;
    const std::size_t lds = ShootingAssembleWaveLds(NZ, NU, NE, a->nh);
    const char* clocks = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ASSEMBLE_CLOCKS");
    if (clocks && clocks[0] == '1') hipLaunchKernelGGL((ShootingAssembleWaveKernel<NZ, NU, NE, true>), dim3(static_cast<unsigned>(d.batch * (d.N + 1))), dim3(64), lds, static_cast<hipStream_t>(stream), *a);
    else hipLaunchKernelGGL((ShootingAssembleWaveKernel<NZ, NU, NE, false>), dim3(static_cast<unsigned>(d.batch * (d.N + 1))), dim3(64), lds, static_cast<hipStream_t>(stream), *a);
    return 0;
}
}  // namespace

/// The same kernel for any other shape it fits, instantiated by the kernel factory on first use (runtime/kernel_jit.cpp; 0: launched, -1: not applicable or no
/// compiler on the machine -- reported once, the workgroup kernel takes over).
static const ungar_amd::runtime::JitKernel* FactoryAssembleWave(int nz, int nu, int ne) {
    static std::mutex mutex;  // (asked once per shape and process: the factory reads the kernel sources to key the entry)
    static std::map<std::tuple<int, int, int, int>, const ungar_amd::runtime::JitKernel*> known;  // per (shape, device): a code object is loaded into one device's context
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    std::lock_guard<std::mutex> guard(mutex);
    if (auto it = known.find({nz, nu, ne, device}); it != known.end()) return it->second;
    ungar_amd::runtime::KernelRequest rq;
    rq.name = "shooting_assemble_wave_" + std::to_string(nz) + "_" + std::to_string(nu) + "_" + std::to_string(ne);
    rq.kernel = "ungar_shooting_assemble_wave";
    rq.source = "#include \"kernels/ocp_shooting_wave_kernel.hpp\"\n"
                "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(%W%, %W%))) void ungar_shooting_assemble_wave(const ungar_amd::kernels::ShootingAssembleArgs a) {\n"
                "    ungar_amd::kernels::ShootingAssembleWaveBody<" + std::to_string(nz) + ", " + std::to_string(nu) + ", " + std::to_string(ne) + ">(a);\n}\n";
    rq.occupancies = {1, 2, 3, 4};
    const ungar_amd::runtime::JitKernel* k = ungar_amd::runtime::GetKernel(rq);
    if (!k) {
        static bool reported = false;
        if (!reported) std::fprintf(stderr, "[ungar_amd] one-wavefront assembly kernel %d + %d, %d rows not available (%s): the workgroup kernel takes over\n", nz, nu, ne, ungar_last_error());
        reported = true;
    }
    if (k) known[{nz, nu, ne, device}] = k;  // (a failed build is asked again next time)
    return k;
}

static bool AssembleWaveApplies(const ShootingAssembleArgs* a) {
    const ShootingDims& d = a->d;
    if (a->eliminate != 1 || !ShootingAssembleWaveFits(d.nz(), d.nu, a->ne)) return false;
    return !(a->nh > 64 || (a->nh > 0 && a->ph.nnz > 64) || a->pH.nnz > 256 || a->pg.nnz > 64 || a->pf.nnz > 256 || a->pe.nnz > 256 || (!d.carryInputs && a->pc.nnz > 128));
}

static int LaunchAssembleWaveFactory(const ShootingAssembleArgs* a, void* stream) {
    if (!AssembleWaveApplies(a)) return -1;
    const ShootingDims& d = a->d;
    const ungar_amd::runtime::JitKernel* k = FactoryAssembleWave(d.nz(), d.nu, a->ne);
    if (!k || !k->function) return -1;
    ShootingAssembleArgs args = *a;
    void* params[] = {&args};
    const hipError_t e = hipModuleLaunchKernel(k->function, static_cast<unsigned>(d.batch * (d.N + 1)), 1, 1, 64, 1, 1,
                                               static_cast<unsigned>(ShootingAssembleWaveLds(d.nz(), d.nu, a->ne, a->nh)), static_cast<hipStream_t>(stream), params, nullptr);
    return static_cast<int>(e);  // a launch error is the caller's to report, not a reason to take the slow route silently
}

/// Which assembly kernel a stage problem of this shape takes: 0 workgroup kernel, 1 one-wavefront kernel compiled into the library, 2 one-wavefront kernel from the
/// kernel factory (built now if `prepare`), 3 the run-time-size one-wavefront kernel of the problems without equality rows.
extern "C" int ungar_amd_shooting_assemble_route(int nz, int nu, int ne, int nh, int prepare) {
    if (ne == 0) return nz + nu + 1 <= 64 && nh <= 64 ? 3 : 0;
    if (!ShootingAssembleWaveFits(nz, nu, ne) || nh > 64) return 0;
    if (nz == 25 && nu == 24 && ne == 16) return 1;
    if (prepare && !FactoryAssembleWave(nz, nu, ne)) return 0;
    return 2;
}

/// Several nodes per wavefront for the narrow ones of those problems (0: launched; -1: not applicable).
static int LaunchAssemblePacked(const ShootingAssembleArgs* a, void* stream) {
    const ShootingDims& d = a->d;
    const int nd = d.nd(), nz = d.nz(), nh1 = nd + 1;
    // (rows of up to 16 columns only.  With 32 lanes per node the quadrotor's 26 columns were measured too: 297 against 273 us per 4096 x 31 nodes -- its dense blocks are
    // 0.9 GB per launch, the loops that write them cost the same number of instructions per node either way and cover 256 instead of 512 contiguous bytes per instruction.)
    constexpr int G = 16;
    if (a->ne != 0 || nh1 > G) return -1;
    if (a->nh > G || (a->nh > 0 && a->ph.nnz > G) || a->pH.nnz > 8 * G || a->pg.nnz > G || a->pf.nnz > 8 * G || (!d.carryInputs && a->pc.nnz > 4 * G)) return -1;
    const int widest = std::max(std::max(a->pH.nnz, a->pf.nnz), d.carryInputs ? 0 : 2 * a->pc.nnz);
    const long long nodes = static_cast<long long>(d.batch) * (d.N + 1);
    const int perWave = 64 / G;
    const std::size_t lds = static_cast<std::size_t>(perWave) * ((((nh1 * (nh1 + 1) / 2) + 1) & ~1) + ((nz * nh1 + 1) & ~1) + 2 * static_cast<std::size_t>(a->nh)) * sizeof(double);
    const dim3 grid(static_cast<unsigned>((nodes + perWave - 1) / perWave));
#define UNGAR_PACKED(SS) hipLaunchKernelGGL((ShootingAssemblePackedKernel<G, SS>), grid, dim3(64), lds, static_cast<hipStream_t>(stream), *a)
    if (widest <= 2 * G) UNGAR_PACKED(2);
    else if (widest <= 4 * G) UNGAR_PACKED(4);
    else UNGAR_PACKED(8);
#undef UNGAR_PACKED
    return 0;
}

/// The one-wavefront kernel for stage problems without equality rows (0: launched; -1: not applicable).
static int LaunchAssembleSmall(const ShootingAssembleArgs* a, void* stream) {
    const ShootingDims& d = a->d;
    const int nd = d.nd(), nz = d.nz(), nh1 = nd + 1;
    if (a->ne != 0 || nh1 > 64 || a->nh > 64 || (a->nh > 0 && a->ph.nnz > 64) || a->pH.nnz > 256 || a->pg.nnz > 64 || a->pf.nnz > 256 || (!d.carryInputs && a->pc.nnz > 128)) return -1;
    const std::size_t lds = ((((nh1 * (nh1 + 1) / 2) + 1) & ~1) + ((nz * nh1 + 1) & ~1) + 2 * static_cast<std::size_t>(a->nh)) * sizeof(double) + 64 * sizeof(int);
    hipLaunchKernelGGL(ShootingAssembleSmallKernel, dim3(static_cast<unsigned>(d.batch * (d.N + 1))), dim3(64), lds, static_cast<hipStream_t>(stream), *a);
    return 0;
}

extern "C" int ungar_amd_launch_shooting_assemble(const ShootingAssembleArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    {
        const char* variant = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ASSEMBLE_VARIANT");  // "workgroup": the kernel below for every shape (measurement, A/B tests); read per call
        if (!(variant && variant[0] == 'w')) {
            if (LaunchAssembleWave<25, 24, 16>(a, stream) == 0) return static_cast<int>(hipGetLastError());
            const bool onePerWave = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ASSEMBLE_ONE_NODE_PER_WAVEFRONT") != nullptr;  // A/B switch (read per call): the kernel with one node per wavefront for the narrow problems too
            if (!onePerWave && LaunchAssemblePacked(a, stream) == 0) return static_cast<int>(hipGetLastError());
            if (LaunchAssembleSmall(a, stream) == 0) return static_cast<int>(hipGetLastError());
            if (const int rc = LaunchAssembleWaveFactory(a, stream); rc == 0) return static_cast<int>(hipGetLastError());
            else if (rc > 0) return rc;  // a HIP error of the launch itself (-1: no factory kernel for this shape: the workgroup kernel below)
        }
    }
    const std::size_t nd = static_cast<std::size_t>(a->d.nd()), nz = static_cast<std::size_t>(a->d.nz());
    const std::size_t ne = static_cast<std::size_t>(a->ne), nu = static_cast<std::size_t>(a->d.nu);
    const std::size_t lds = (nd * (nd + 1) / 2 + nd + 2 * static_cast<std::size_t>(a->nh) + nz * nd + 2 * ne * (nd + 1) + nz + (nd + 1) + 6 + ne /*row scales*/) * sizeof(double) +
                            (ne + nu + (nd > ne ? nd : ne) + 2) * sizeof(int) + 16;
    if (lds > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ShootingAssembleKernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    static const int forcedLanes = [] {
        const char* e = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ASSEMBLE_LANES");  // measurement knob: 64 / 128 / 256 lanes per node
        return e ? atoi(e) : 0;
    }();
    const int lanesPerNode = forcedLanes == 64 || forcedLanes == 128 || forcedLanes == 256 ? forcedLanes : (nd >= 32 ? 256 : kBlock);
    hipLaunchKernelGGL(ShootingAssembleKernel, dim3(static_cast<unsigned>(a->d.batch * (a->d.N + 1))), dim3(lanesPerNode), lds, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_recover(const ShootingRecoverArgs* a, void* stream) {
    const long long items = a->d.batch * a->d.N * a->ne;
    if (items <= 0) return 0;
    hipLaunchKernelGGL(ShootingRecoverKernel, dim3(static_cast<unsigned>((4 * items + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_refresh_carried_inputs(const ShootingDims* d, double* rows, void* stream) {
    const long long items = d->batch * d->N * d->nu;
    if (items <= 0) return 0;
    hipLaunchKernelGGL(ShootingRefreshCarriedInputsKernel, dim3(static_cast<unsigned>((items + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), *d, rows);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_merit(const ShootingMeritArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    hipLaunchKernelGGL(ShootingMeritKernel, dim3(static_cast<unsigned>(a->d.batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_trial(const ShootingTrialArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    const long long nodes = static_cast<long long>(a->candidates) * (a->listed > 0 ? a->listed : a->d.batch) * (a->d.N + 1);
    if (a->trialStride > 0) {
        // as many candidates per workgroup as possible -- they share the reads of row and direction -- once there are four workgroups per CU (measured on 1984
        // groups of nodes: two candidates in two workgroups each 77 us, in one 52 us; three groups of candidates 40 us, one 35 us)
        const long long groupsOfNodes = (nodes / a->candidates + 63) / 64, wanted = (1024 + groupsOfNodes - 1) / groupsOfNodes;
        const unsigned candidateGroups = static_cast<unsigned>(wanted < a->candidates ? wanted : a->candidates);
        hipLaunchKernelGGL(ShootingTrialUnitFastestKernel, dim3(static_cast<unsigned>(groupsOfNodes), candidateGroups), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    }
    else hipLaunchKernelGGL(ShootingTrialKernel, dim3(static_cast<unsigned>(nodes)), dim3(a->d.nv() > 64 ? 128 : kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_select(const ShootingSelectArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    hipLaunchKernelGGL(ShootingSelectKernel, dim3(static_cast<unsigned>(((a->listed > 0 ? a->listed : a->d.batch) + kSelectWaves - 1) / kSelectWaves)), dim3(kBlock * kSelectWaves), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}
