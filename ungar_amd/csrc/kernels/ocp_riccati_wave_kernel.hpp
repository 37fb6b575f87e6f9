// ungar_amd :: the Riccati recursion of ocp_riccati.hpp with ONE WAVEFRONT PER INSTANCE and every matrix of the recursion in REGISTERS, in the operand /
// accumulator layouts of v_mfma_f64_16x16x4_f64 (replaces the OSQP call of soft_sqp.hpp:143-158 for shooting structure).
//
// Why.  The LDS-resident kernels (ocp_riccati.hip) run one workgroup of four wavefronts per instance: 66-78 KB of LDS per instance keep two workgroups on
// a CU, every phase ends in a workgroup barrier and costs its critical-path instruction count -- 0.10-0.15 of the roofline for the
// 37 + 12, 25 + 24 and 13 + 24 blocks.  Here nothing is shared between wavefronts, so there is no barrier at all, and the products feed each other WITHOUT
// any data movement because of how the matrix instruction lays its operands out over the lanes:
//     A[i][k] in lane 16 k + i,   B[k][j] in lane 16 k + j,   D[(lane >> 4) + 4 r][lane & 15] in accumulator element r.
//   * element r of an accumulator tile is rows 4 r .. 4 r + 3 of the tile in exactly the B layout: an accumulator is the B operand of the next product,
//     k-step by k-step (P [A|B] -> [A|B]^T (P [A|B]));
//   * the A and the B layout coincide, so the registers that hold [A|B] as B operand of P [A|B] are the A operand ([A|B]^T) of the second product;
//   * P is symmetric: its accumulator tiles are the A operand of P [A|B] (A[i][k] = P[k][i]).
// The affine parts ride along in HOMOGENEOUS coordinates -- x_e = [x; 1], P_e = [P p; p^T *], [A|B]_e = [A b B; 0 1 0] -- so that
//     H_e = W_e + [A|B]_e^T P_e [A|B]_e = [H_xx h_x H_xu; h_x^T * h_u^T; H_ux h_u R]      (one chain of matrix instructions, no separate vector updates),
//     [K | kff] = -R^-1 [H_ux | h_u],   P_e' = H_e,xx + H_e,xu [K | kff]                    (ocp_riccati.hpp: the same recursion).
// R is factorised by symmetric elimination without pivoting on the tableau [R | H_ux h_u], lane = column, rows in registers, pivot columns broadcast with
// v_readlane: the multipliers and pivots of the right-looking L D L^T of ocp_riccati.hpp (a pivot that is not positive is replaced by 1 and reported).
// LDS (a few KB per wavefront, private to it) is used only to change layouts: accumulator tiles -> tableau columns, gains -> B operand, tile transposes.
//
// This header holds the kernel TEMPLATE: the library instantiates it for the stage sizes of the reference's own problems (ocp_riccati_wave.hip), and the kernel
// factory (runtime/kernel_jit.cpp) instantiates it at run time for whatever sizes a problem declares, in a translation unit of three lines that includes this file.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "ocp_riccati.hpp"

namespace ungar_amd::kernels {
namespace {


using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;

__device__ __forceinline__ double ReadLaneF64(double v, int sourceLane) {  // sourceLane wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), sourceLane), hi = __builtin_amdgcn_readlane(__double2hiint(v), sourceLane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double QuadPermuteF64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

using v2i = __attribute__((__vector_size__(2 * sizeof(int)))) int;
constexpr int kOutOfRange = static_cast<int>(0x80000000u);  // a lane offset no buffer below reaches (also with a scalar / instruction offset added): the load returns 0

/// Operands are read with MUBUF loads: (wave-uniform resource) + (32-bit lane offset) + (scalar offset).  A lane that has nothing to read passes
/// kOutOfRange and gets 0 from the range check -- padded rows / columns cost neither a branch nor a select on the loaded value (a select between a load
/// and a constant is turned into a branch around the load by the compiler, and every such load then waits for its own round trip: measured, 27-54 k
/// cycles per knot), and one lane register serves every load of a tile pattern, the tile's position being the scalar offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t BufferOver(const double* base, int doubles) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, doubles * 8, 0x00020000);
}
using v4i = __attribute__((__vector_size__(4 * sizeof(int)))) int;
__device__ __forceinline__ v4i BufferLoad16(__amdgpu_buffer_rsrc_t r, int laneOffsetBytes, int scalarOffsetBytes) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, laneOffsetBytes, scalarOffsetBytes, 0);
}
__device__ __forceinline__ double BufferLoad(__amdgpu_buffer_rsrc_t r, int laneOffsetBytes, int scalarOffsetBytes) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, laneOffsetBytes, scalarOffsetBytes, 0));
}

/// Prefetch: every lane names one address; the cache line behind it is brought into L2 / the memory-side cache by an LDS-DMA load of four bytes into a
/// junk area of LDS (`ldsJunk`: LDS byte address of 256 bytes nobody reads).  No destination register, nothing for the compiler to wait for, no count
/// of ours either: the data is never used -- only the cache line matters.  (The operands of a knot are 34 KB per wavefront and all wavefronts reach the
/// same point of the recursion together: requested where they are needed, the whole device waits for HBM -- 70 MB per knot of the 37 + 12 problem, 17 us --
/// and computes afterwards; touched one knot ahead, HBM works during the products.)
__device__ __forceinline__ void TouchLine(const void* lanePointer, unsigned ldsJunk) {
#if defined(UNGAR_RICCATI_NO_TOUCH)  // (A/B of the prefetch through the kernel factory: UNGAR_AMD_JIT_FLAGS="-O3 -DUNGAR_RICCATI_NO_TOUCH")
    (void)lanePointer, (void)ldsJunk;
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lanePointer), "s"(ldsJunk) : "memory");
}

/// LDS traffic of one wavefront needs no barrier, only program order: the hardware executes the LDS instructions of a wavefront in order, the fence keeps
/// the compiler from moving accesses across.
__device__ __forceinline__ void WaveLdsFence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/// 16-byte pieces (two doubles, counted from the start of the block) of a row-major n x n block that hold at least one entry on or right of the diagonal, row by
/// row (a piece shared by two rows counts twice: harmless) -- what the recursion reads of a stage Hessian, of which only the upper triangle is meaningful.
constexpr int UpperTrianglePieces(int n) {
    int count = 0;
    for (int r = 0; r < n; ++r) count += ((r * n + n - 1) >> 1) - ((r * n + r) >> 1) + 1;
    return count;
}

template <int NX, int NU>
struct WaveSizes {
    static constexpr int n = NX + NU, nk = NX + 1;
    static constexpr int XE = NX + 1;                     // [x; 1]
    static constexpr int XT = (XE + 15) / 16, UT = (NU + 15) / 16, NT = XT + UT;
    static constexpr int KX = (XE + 3) / 4, KU = (NU + 3) / 4;  // k-steps over [x; 1] and over u
    static constexpr int TW = NU + XE;                    // tableau [R | H_ux h_u]: one lane per column
    static constexpr int TLD = TW | 1;                    // (odd row stride)
    static constexpr int KLD = XE | 1;                    // gains in LDS: NU rows of [K | kff]
    static constexpr int kAbDoubles = (NX * n + NX + 1) / 2 * 2;  // [A|B] of the knot as it lies in memory, then b (LDS-DMA: 16 bytes per lane)
    static constexpr int kWDoubles = (n * n + n + 1) / 2 * 2;     // stage Hessian of the knot as it lies in memory (upper triangle meaningful), then the stage gradient
    // The tableau passes from the accumulator tiles to the lanes (one column each) through LDS.  Where the state is one tile wide and the inputs are two
    // (13 + 24: 23.0 KB of LDS per wavefront = six per CU, three rounds of 4096 instances with the last one two-thirds empty), it goes in two halves of
    // its rows through a buffer of half the size: 19.3 KB, eight wavefronts per CU, two rounds.
    static constexpr int kTableauPasses = (NU > 16 && XT == 1) ? 2 : 1;
    static constexpr int kTableauRows = (NU + kTableauPasses - 1) / kTableauPasses;
    static constexpr int kMax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int kTableauDoubles = kMax3(kTableauRows * TLD, NU * KLD, 16 * 17);  // tableau (a pass of it), then gains, then tile transposes
    static constexpr int kLdsDoubles = kAbDoubles + kWDoubles + kTableauDoubles + XE + NU + 8;
    static_assert(TW <= 64, "the tableau of the factorisation needs one lane per column");
};

/// kLdsDoubles of WaveSizes<nx, nu> for sizes known at run time (the launcher of an instantiation made by the kernel factory).
constexpr int RiccatiWaveLdsDoubles(int nx, int nu) {
    const int n = nx + nu, XE = nx + 1, XT = (XE + 15) / 16, TW = nu + XE, TLD = TW | 1, KLD = XE | 1;
    const int ab = (nx * n + nx + 1) / 2 * 2, w = (n * n + n + 1) / 2 * 2;
    const int passes = (nu > 16 && XT == 1) ? 2 : 1, rows = (nu + passes - 1) / passes;
    const int t0 = rows * TLD, t1 = nu * KLD, t2 = 16 * 17;
    const int tableau = t0 > t1 ? (t0 > t2 ? t0 : t2) : (t1 > t2 ? t1 : t2);
    return ab + w + tableau + XE + nu + 8;
}
static_assert(RiccatiWaveLdsDoubles(37, 12) == WaveSizes<37, 12>::kLdsDoubles && RiccatiWaveLdsDoubles(13, 24) == WaveSizes<13, 24>::kLdsDoubles &&
                  RiccatiWaveLdsDoubles(17, 4) == WaveSizes<17, 4>::kLdsDoubles && RiccatiWaveLdsDoubles(10, 3) == WaveSizes<10, 3>::kLdsDoubles,
              "RiccatiWaveLdsDoubles mirrors WaveSizes");

/// The recursion of one instance by one wavefront.  CLOCKS: diagnostic instantiation (measurement build, UNGAR_AMD_RICCATI_WAVE_CLOCKS=1) -- the first
/// wavefront prints the cycles it spent in every section.
template <int NX, int NU, bool CLOCKS = false>
__device__ __forceinline__ void RiccatiWaveBody(const RiccatiArgs& a) {
    using S = WaveSizes<NX, NU>;
    constexpr int n = S::n, nk = S::nk, XE = S::XE, XT = S::XT, UT = S::UT, NT = S::NT, KX = S::KX, KU = S::KU, TW = S::TW, TLD = S::TLD, KLD = S::KLD;
    extern __shared__ double lds[];
    const long long inst = blockIdx.x;
    if (inst >= a.batch) return;
    const int lane = static_cast<int>(threadIdx.x), lj = lane & 15, lk = lane >> 4;
    const int N = a.N;
    double* ABl = lds;                 // NX x n: [A|B] of the knot, row-major as in memory (16-byte aligned: the LDS-DMA writes 16 bytes per lane)
    double* bl = ABl + NX * n;         // NX: b of the knot
    double* Wl = lds + S::kAbDoubles;  // n x n: stage Hessian of the knot (upper triangle), row-major as in memory
    double* gl = Wl + n * n;           // n: stage gradient
    double* T = Wl + S::kWDoubles;     // NU x TLD: tableau of the factorisation
    double* Kb = T;                    // NU x KLD: gains (the tableau is in registers by then)
    double* tr = T;                    // 16 x 17: tile transposes (the gains are in registers by then)
    double* dxv = T + S::kTableauDoubles;  // XE: [dx; 1]
    double* duv = dxv + XE;            // NU
    double* gains = a.gains + inst * static_cast<long long>(N) * NU * nk;
    const double reg = a.regularization;
    constexpr int oneTile = NX / 16, oneCol = NX % 16;  // where the homogeneous 1 sits among the x columns

    // ---- operands of a knot: global -> registers -> LDS, 16 bytes per lane and instruction, one knot ahead.  They are consumed (copied into the operand /
    // accumulator registers) at the top of their knot; the NEXT knot's are then fetched in NT shares, one per column tile of the products: a share is
    // requested at the top of its tile and written to LDS at the top of the following one, its round trip to memory in the shadow of the tile's matrix
    // instructions.  (All wavefronts reach the same point of the recursion together and ask for 34 KB each: requested where they are needed, the device waits
    // for HBM -- 17 us per knot of the 37 + 12 problem -- and computes afterwards.  The LDS-DMA path needs no staging registers, but its instructions take
    // 200+ cycles each to issue here, 37 of them per knot: measured, 8 k cycles per knot even between matrix instructions.)
    // Of the stage Hessian only the upper triangle is fetched (the rest of the block is never read: half the bytes of the largest operand): the 16-byte pieces
    // that hold an entry on or right of the diagonal, dealt to the lanes in order; a lane's piece offsets are the same for every knot and computed once.
    constexpr int abBytes = NX * n * 8, wBytes = n * n * 8, abPieces = (abBytes + 1023) / 1024, wPieces = (UpperTrianglePieces(n) + 63) / 64, pieces = abPieces + wPieces;
    int wOff[wPieces];
#pragma unroll
    for (int i = 0; i < wPieces; ++i) wOff[i] = kOutOfRange;
    {
        int begin = 0;
        for (int r = 0; r < n; ++r) {
            const int first = (r * n + r) >> 1, count = ((r * n + n - 1) >> 1) - first + 1;
#pragma unroll
            for (int i = 0; i < wPieces; ++i) {
                const int q = 64 * i + lane - begin;
                if (q >= 0 && q < count) wOff[i] = (first + q) * 16;
            }
            begin += count;
        }
    }
    // (measured and dropped: all shares requested during the first two column tiles and stored after the last one -- the staging registers of a whole knot,
    // 136 for 37 + 12, spill: 1.48 -> 2.4 ms)
    constexpr int kRequestTiles = NT, perTile = (pieces + kRequestTiles - 1) / kRequestTiles;
    v4i stage[perTile];  // (one share: a tile stores the previous one before it requests its own)
    double stageB = 0.0, stageG = 0.0;
    auto requestShare = [&](int k, int tile) {
        const __amdgpu_buffer_rsrc_t rsJ = BufferOver(&a.jac.at(inst, k, 0), NX * n), rsW = BufferOver(&a.hess.at(inst, k, 0), n * n);
#pragma unroll
        for (int i = 0; i < perTile; ++i) {
            const int p = i * kRequestTiles + tile;
            if (p < abPieces) stage[i] = BufferLoad16(rsJ, lane * 16, p * 1024);  // (beyond the block: zeros, not stored)
            else if (p < pieces) stage[i] = BufferLoad16(rsW, wOff[p - abPieces], 0);
        }
        if (tile == 0) {
            stageB = BufferLoad(BufferOver(&a.b.at(inst, k, 0), NX), lane * 8, 0);
            stageG = BufferLoad(BufferOver(&a.grad.at(inst, k, 0), n), lane * 8, 0);
        }
    };
    auto storeShare = [&](int tile) {
#pragma unroll
        for (int i = 0; i < perTile; ++i) {
            const int p = i * kRequestTiles + tile;
            if (p >= pieces) continue;
            const bool first = p < abPieces;
            const int off = first ? p * 1024 + lane * 16 : wOff[p - abPieces], bytes = first ? abBytes : wBytes;  // (a lane without a piece: kOutOfRange < 0, nothing stored)
            double* dst = (first ? ABl : Wl) + (off >= 0 ? off : 0) / 8;
            if (off >= 0 && off + 16 <= bytes) *reinterpret_cast<v4i*>(dst) = stage[i];
            else if (off >= 0 && off + 8 <= bytes) *dst = __hiloint2double(stage[i][1], stage[i][0]);  // an odd number of doubles: the last one
        }
        if (tile == 0) {
            if (lane < NX) bl[lane] = stageB;
            if (lane < n) gl[lane] = stageG;
        }
    };
    // entry (4 ks + lk, column lj of tile tj) of [A b B; 0 1 0]: the B operand of P_e [A|B]_e and, the two layouts being the same, the A operand ([A|B]_e^T)
    // of the second product.  Padded rows / columns are masked in the integer domain (a select between a loaded value and a constant would become a branch).
    auto masked = [](double v, bool keepIt) { return __longlong_as_double(__double_as_longlong(v) & (keepIt ? -1ll : 0ll)); };
    auto abOperand = [&](int ks, int tj) -> double {
        const int row = 4 * ks + lk;
        const bool rowInterior = 4 * ks + 3 < NX;  // (compile time)
        const int col0 = tj < XT ? 16 * tj : 16 * (tj - XT), limit = tj < XT ? NX : NU, colBase = tj < XT ? col0 : NX + col0;
        const bool colInterior = col0 + 15 < limit;
        if (4 * ks > NX) return 0.0;  // (compile time: rows behind the homogeneous 1)
        const bool rowOk = rowInterior || row < NX, colOk = colInterior || col0 + lj < limit;
        double v = 0.0;
        if (4 * ks < NX && col0 < limit) {
            const double raw = ABl[(rowOk ? row : 0) * n + colBase + (colOk ? lj : 0)];
            v = rowInterior && colInterior ? raw : masked(raw, rowOk && colOk);
        }
        if (tj == oneTile) {
            if (4 * ks < NX) v += masked(bl[rowOk ? row : 0], rowOk && lj == oneCol);
            if (4 * ks <= NX && NX < 4 * ks + 4) v += lk == NX % 4 && lj == oneCol ? 1.0 : 0.0;
        }
        return v;
    };

    // ---- cost-to-go in homogeneous form P_e = [P p; p^T *], all XT x XT accumulator tiles: P[tr][tc][r] = P_e[16 tr + 4 r + lk][16 tc + lj].  It is the A
    // operand of P_e [A|B]_e READ AS ITS OWN TRANSPOSE (A[i][k] = P_e[k][i]), so it must be symmetric to the last bit: what the recursion does to an
    // antisymmetric part is P_a' = -A^T P_a A -- undamped by the feedback, it grows like |A|^(2 k) from a rounding error (measured: 1e-3 after 30 knots).
    f64x4 P[XT][XT];
    {
        const int ldN = a.hessNld > 0 ? a.hessNld : NX;
#pragma unroll
        for (int ti = 0; ti < XT; ++ti)
#pragma unroll
            for (int tj = 0; tj < XT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + lk, col = 16 * tj + lj;
                    const int lo = row < col ? row : col, hi = row < col ? col : row;
                    double v = 0.0;
                    if (hi < NX) {
                        if (a.hessN.base) v = a.hessN.at(inst, 0, lo * ldN + hi);
                        if (lo == hi) v += reg;
                    } else if (hi == NX && lo < NX) {
                        if (a.gradN.base) v = a.gradN.at(inst, 0, lo);
                    }
                    P[ti][tj][r] = v;
                }
    }

    // ---- accumulators of H_e: xx on and above the tile diagonal, ux whole, uu on and below the tile diagonal
    f64x4 Hxx[XT][XT];  // [ti][tj], ti <= tj used
    f64x4 Hux[UT][XT];
    f64x4 Huu[UT][UT];  // [tu][tv], tv <= tu used
    // lane parts of the offsets (the same for every tile): entry (lk, lj) and entry (lj, lk) of the row-major n x n block, and the entry of a DIAGONAL
    // tile of a symmetric block of which the upper triangle is stored: (4 r + lk, lj) or its mirror image
    const int vb = lk * n + lj, vt = lj * n + lk;
    int vd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vd[r] = 4 * r + lk <= lj ? (4 * r + lk) * n + lj : lj * n + 4 * r + lk;
    auto fromLds = [&](const double* block, int index, bool ok) { return masked(block[ok ? index : 0], ok); };
    double ab[KX][NT];
    auto takeOperands = [&]() {  // LDS -> registers: [A b B; 0 1 0] in operand layout, H_e <- W_e (stage Hessian, gradient in the row / column of the 1, regularisation)
#pragma unroll
        for (int ks = 0; ks < KX; ++ks)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) ab[ks][tj] = abOperand(ks, tj);
#pragma unroll
        for (int ti = 0; ti < XT; ++ti)
#pragma unroll
            for (int tj = ti; tj < XT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowL = 4 * r + lk;
                    double v;
                    if (ti < tj) {  // rows all inside x (16 (XT - 1) <= NX), columns up to NX - 1
                        const bool colOk = 16 * tj + 15 < NX || 16 * tj + lj < NX;
                        const double raw = Wl[(16 * ti + 4 * r) * n + 16 * tj + (colOk ? vb : lk * n)];
                        v = 16 * tj + 15 < NX ? raw : masked(raw, colOk);
                        if (tj == oneTile) v += fromLds(gl, 16 * ti + rowL, lj == oneCol);  // W_e[x][1] = w[x]
                    } else {
                        const int hi = rowL > lj ? rowL : lj, lo = rowL > lj ? lj : rowL;
                        const bool ok = 16 * ti + 15 < NX || 16 * ti + hi < NX;
                        const double raw = Wl[16 * ti * (n + 1) + (ok ? vd[r] : 0)];
                        v = (16 * ti + 15 < NX ? raw : masked(raw, ok)) + (ok && rowL == lj ? reg : 0.0);
                        if (ti == oneTile) v += fromLds(gl, 16 * ti + lo, hi == oneCol && lo < oneCol);  // row and column of the 1
                    }
                    Hxx[ti][tj][r] = v;
                }
#pragma unroll
        for (int tu = 0; tu < UT; ++tu)
#pragma unroll
            for (int tx = 0; tx < XT; ++tx)
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // H_ux[u][x] = W[x][NX + u]
                    const bool uOk = 16 * tu + 4 * r + 3 < NU || 16 * tu + 4 * r + lk < NU, colOk = 16 * tx + 15 < NX || 16 * tx + lj < NX;
                    double v = 0.0;
                    if (16 * tu + 4 * r < NU) {  // (compile time)
                        v = fromLds(Wl + 16 * tx * n + NX + 16 * tu + 4 * r, vt, uOk && colOk);
                        if (tx == oneTile) v += fromLds(gl + NX + 16 * tu + 4 * r, lk, uOk && lj == oneCol);
                    }
                    Hux[tu][tx][r] = v;
                }
#pragma unroll
        for (int tu = 0; tu < UT; ++tu)
#pragma unroll
            for (int tv = 0; tv <= tu; ++tv)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowL = 4 * r + lk;
                    double v = 0.0;
                    if (16 * tu + 4 * r < NU) {  // (compile time; padded input rows / columns stay zero and are never read)
                        if (tv < tu) {           // R[u][v] = W[NX + v][NX + u], every column v inside u's range
                            const bool uOk = 16 * tu + 4 * r + 3 < NU || 16 * tu + rowL < NU;
                            v = fromLds(Wl + (NX + 16 * tv) * n + NX + 16 * tu + 4 * r, vt, uOk);
                        } else {
                            const int hi = rowL > lj ? rowL : lj;
                            const bool ok = 16 * tu + 15 < NU || 16 * tu + hi < NU;
                            v = fromLds(Wl + (NX + 16 * tu) * (n + 1), vd[r], ok) + (ok && rowL == lj ? reg : 0.0);
                        }
                    }
                    Huu[tu][tv][r] = v;
                }
    };

    unsigned long long clocks[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    auto mark = [&](int id) {
        if constexpr (CLOCKS) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (id >= 0) clocks[id] += now - last;
            last = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int failed = 0;
#pragma unroll
    for (int tile = 0; tile < kRequestTiles; ++tile) {
        requestShare(N - 1, tile);
        storeShare(tile);
    }
    mark(-1);
    for (int k = N - 1; k >= 0; --k) {
        WaveLdsFence();
        mark(6);  // waiting for the knot's operands
        takeOperands();
        WaveLdsFence();  // the buffers are free again: the next knot's operands are requested during the products below and have the rest of the knot to land in
        mark(1);  // operands LDS -> registers
        // ---- H_e += [A|B]_e^T (P_e [A|B]_e), one column tile of P_e [A|B]_e at a time (its XT accumulators are the B operand of that column's H tiles)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            if (k > 0) {
                __builtin_amdgcn_sched_barrier(0);  // (the previous tile's matrix instructions stay in front of the stores that wait for its share)
                if (tj > 0) storeShare(tj - 1);
                requestShare(k - 1, tj);
                __builtin_amdgcn_sched_barrier(0);
            }
            f64x4 pab[XT];
#pragma unroll
            for (int ti = 0; ti < XT; ++ti) pab[ti] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KX; ++ks) {
#pragma unroll
                for (int ti = 0; ti < XT; ++ti) pab[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[ks >> 2][ti][ks & 3], ab[ks][tj], pab[ti], 0, 0, 0);
            }
            if (tj < XT) {
#pragma unroll
                for (int ks = 0; ks < KX; ++ks) {
#pragma unroll
                    for (int ti = 0; ti <= tj; ++ti) Hxx[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks][ti], pab[ks >> 2][ks & 3], Hxx[ti][tj], 0, 0, 0);
#pragma unroll
                    for (int tu = 0; tu < UT; ++tu) Hux[tu][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks][XT + tu], pab[ks >> 2][ks & 3], Hux[tu][tj], 0, 0, 0);
                }
            } else {
                const int tv = tj - XT;
#pragma unroll
                for (int ks = 0; ks < KX; ++ks)
#pragma unroll
                    for (int tu = tv; tu < UT; ++tu) Huu[tu][tv] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab[ks][XT + tu], pab[ks >> 2][ks & 3], Huu[tu][tv], 0, 0, 0);
            }
        }
        if (k > 0) storeShare(NT - 1);
        mark(0);  // products

        // ---- tableau [R | H_ux h_u] -> LDS (row-major), both triangles of R from the tiles on and below the diagonal -> lane = column: t[i] = row i of
        // [R | -H_ux -h_u], the rows in kTableauPasses groups through a buffer of one group
        double t[NU];
#pragma unroll
        for (int pass = 0; pass < S::kTableauPasses; ++pass) {
            constexpr int rows = S::kTableauRows;
            const int first = pass * rows;
            auto put = [&](int row, int col, double v) {
                if (S::kTableauPasses == 1 || (row >= first && row < first + rows)) T[(row - first) * TLD + col] = v;
            };
#pragma unroll
            for (int tu = 0; tu < UT; ++tu) {
#pragma unroll
                for (int tv = 0; tv <= tu; ++tv)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int u = 16 * tu + 4 * r + lk, v = 16 * tv + lj;
                        if (u < NU && v < NU) {
                            if (tv < tu || v <= u) put(u, v, Huu[tu][tv][r]);
                            if (tv < tu || v < u) put(v, u, Huu[tu][tv][r]);
                        }
                    }
#pragma unroll
                for (int tx = 0; tx < XT; ++tx)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int u = 16 * tu + 4 * r + lk, c = 16 * tx + lj;
                        if (u < NU && c < XE) put(u, NU + c, Hux[tu][tx][r]);
                    }
            }
            WaveLdsFence();
#pragma unroll
            for (int i = 0; i < rows; ++i)
                if (first + i < NU) {
                    const double v = T[i * TLD + (lane < TW ? lane : TW - 1)];
                    t[first + i] = lane < NU ? v : (lane < TW ? -v : 0.0);
                }
            if (pass + 1 < S::kTableauPasses) WaveLdsFence();
        }
        // (Measured and dropped, round 4: per-knot words flagging the inputs an assembly eliminated -- decoupled dummies, 8-12 of the quadruped's 24: zero column of
        // [A|B], unit row of the Hessian -- so that their row updates and back-substitution columns are skipped behind scalar branches.  Synthetic 25 + 24 with
        // half the inputs flagged: 1.97 -> 1.93 ms; inside the quadruped iteration 1.98-2.02 -> 2.09 ms.  Skipping the flagged pivots' whole steps costs 40+
        // registers of control flow: 37 + 12 spills (1.4 -> 3.1 ms).  The section is not bound by those instructions.)
        // (Measured and dropped for the elimination below: the pivot's reciprocal chain dealt out between the batches of row updates -- no change, the section is
        // bound by the issue of its ~1100 vector instructions at ~8 cycles each in a lone wavefront, not by that chain; the multipliers broadcast through LDS, the
        // pivot ROW stored once and read back entry by entry, one LDS instruction + one multiply-add per row update instead of two v_readlane + one -- 9 k -> 17 k
        // cycles per knot for 24 inputs: a store-to-load round trip per pivot on the critical path of a lone wavefront.)
        bool bad = false;
        auto reciprocalOfPivot = [&](double d) {  // a pivot that is not positive is replaced by 1 and reported (ocp_riccati.hpp)
            const bool neg = !(d > 0.0);
            bad = bad || neg;
            // hardware reciprocal and two Newton steps (a last-bit-or-so reciprocal) instead of the IEEE division: the pivots' reciprocals are the serial
            // chain of the elimination -- NU of them, each behind the first row update of the previous pivot
            const double x = neg ? 1.0 : d;
            double r = __builtin_amdgcn_rcp(x);
            r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
            return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
        };
        double rd = reciprocalOfPivot(ReadLaneF64(t[0], 0));
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const double prow = t[j] * rd;  // row j of L^T (unit diagonal) next to its forward-substituted, scaled right-hand sides
            // the next pivot is final after the first row update: its reciprocal (a long dependent chain) is started at once, beside the other rows' updates
            if (j + 1 < NU) {
                t[j + 1] = __builtin_fma(-ReadLaneF64(t[j + 1], j), prow, t[j + 1]);
                rd = reciprocalOfPivot(ReadLaneF64(t[j + 1], j + 1));
            }
            // the multipliers of a pivot are read in batches of eight, then applied: one scalar register pair reused for every row made each update wait
            // for the previous one (two v_readlane, the wait states between a scalar write and its use, the multiply-add: ~30 cycles per row)
#pragma unroll
            for (int i0 = j + 2; i0 < NU; i0 += 8) {
                double m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i0 + q < NU) m[q] = ReadLaneF64(t[i0 + q], j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (i0 + q < NU) t[i0 + q] = __builtin_fma(-m[q], prow, t[i0 + q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            t[j] = prow;
        }
        if (bad) failed = failed ? failed : k + 1;
        mark(2);  // forward elimination
        // back substitution L^T X = Y, column by column: once x_c is final, x_j -= L^T[j][c] x_c for the rows above.  Only the right-hand-side lanes take
        // part: row c is zeroed elsewhere first (its entries of L^T are not read again), so that the updates need no select and leave L^T intact.
#pragma unroll
        for (int c = NU - 1; c > 0; --c) {
            const double xc = lane >= NU ? t[c] : 0.0;
#pragma unroll
            for (int j0 = 0; j0 < c; j0 += 8) {
                double m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (j0 + q < c) m[q] = ReadLaneF64(t[j0 + q], c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (j0 + q < c) t[j0 + q] = __builtin_fma(-m[q], xc, t[j0 + q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mark(3);  // back substitution
        // gains: global (forward pass) and LDS (B operand of the cost-to-go update)
        {
            double* gk = gains + static_cast<long long>(k) * NU * nk;
            const int c = lane - NU;
#pragma unroll
            for (int i = 0; i < NU; ++i)
                if (c >= 0 && c < XE) {
                    gk[i * nk + c] = t[i];
                    Kb[i * KLD + c] = t[i];
                }
        }
        WaveLdsFence();
        double kreg[KU][XT];
#pragma unroll
        for (int ks = 0; ks < KU; ++ks)
#pragma unroll
            for (int tj = 0; tj < XT; ++tj) {
                const int m = 4 * ks + lk, c = 16 * tj + lj;
                kreg[ks][tj] = masked(Kb[(m < NU ? m : NU - 1) * KLD + (c < XE ? c : XE - 1)], m < NU && c < XE);
            }
        WaveLdsFence();
        mark(4);  // gains out and back
        // ---- P_e' = H_e,xx + H_e,xu [K | kff] on the matrix cores (accumulated onto H_e,xx)
#pragma unroll
        for (int ti = 0; ti < XT; ++ti)
#pragma unroll
            for (int tj = ti; tj < XT; ++tj) {
                // (one orientation of T = H_e,xu [K | kff] is enough: what makes P_e' symmetric to the last bit is the mirroring below, and the other
                // orientation, averaged in, was 18 more matrix instructions per knot)
                f64x4 t1 = Hxx[ti][tj];
#pragma unroll
                for (int ks = 0; ks < KU; ++ks) t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Hux[ks >> 2][ti][ks & 3], kreg[ks][tj], t1, 0, 0, 0);
                P[ti][tj] = t1;
            }
        // symmetric to the last bit (see above): the tiles below the diagonal are the transposes of the ones above, a diagonal tile is averaged with its own
        // transpose (through LDS, one tile at a time)
#pragma unroll
        for (int ti = 0; ti < XT; ++ti)
#pragma unroll
            for (int tj = ti; tj < XT; ++tj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tr[(4 * r + lk) * 17 + lj] = P[ti][tj][r];
                WaveLdsFence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double mirrored = tr[lj * 17 + 4 * r + lk];
                    if (ti == tj) P[ti][ti][r] = 0.5 * (P[ti][ti][r] + mirrored);
                    else P[tj][ti][r] = mirrored;
                }
                WaveLdsFence();
            }
        mark(5);  // cost-to-go update and its transposes
    }

    // ---- forward pass: du_k = [K | kff] [dx_k; 1],  dx_(k+1) = [A b B] [dx_k; 1; du_k]; a row's inner product is shared by the four lanes of a quad
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the gains written above are read back below (same wavefront: program order + the fence for the compiler)
    if (lane < XE) {
        const double v = lane < NX ? a.dx0.at(inst, 0, lane) : 1.0;
        dxv[lane] = v;
        if (lane < NX) a.dX.at(inst, 0, lane) = v;
    }
    WaveLdsFence();
    const int part = lane & 3, q = lane >> 2;  // row q of a round of 16 rows, terms part, part + 4, ...
    constexpr int XQ = (XE + 3) / 4, UQ = (NU + 3) / 4, RU = (NU + 15) / 16, RX = (NX + 15) / 16;
    const int vg = (q * nk + part) * 8, vj = (q * n + part) * 8;  // lane parts: entry (q, part) of a row-major NU x nk / NX x n block
    // the operands of knot k + 1 are requested (into registers) before knot k is computed: one knot of lead instead of a round trip to memory per knot
    double gvN[RU][XQ], avN[RX][XQ], bvN[RX][UQ], b0N[RX];
    auto requestForward = [&](int k) {
        const __amdgpu_buffer_rsrc_t rsK = BufferOver(gains + static_cast<long long>(k) * NU * nk, NU * nk), rsJ = BufferOver(&a.jac.at(inst, k, 0), NX * n),
                                     rsB = BufferOver(&a.b.at(inst, k, 0), NX);
#pragma unroll
        for (int rr = 0; rr < RU; ++rr)
#pragma unroll
            for (int m = 0; m < XQ; ++m) {
                const bool ok = (16 * rr + 15 < NU || 16 * rr + q < NU) && (4 * m + 3 < XE || 4 * m + part < XE);
                gvN[rr][m] = BufferLoad(rsK, ok ? vg : kOutOfRange, (16 * rr * nk + 4 * m) * 8);
            }
#pragma unroll
        for (int rr = 0; rr < RX; ++rr) {
            const bool rowOk = 16 * rr + 15 < NX || 16 * rr + q < NX;
#pragma unroll
            for (int m = 0; m < XQ; ++m) {
                const bool ok = rowOk && (4 * m + 3 < NX || 4 * m + part < NX);
                avN[rr][m] = 4 * m < NX ? BufferLoad(rsJ, ok ? vj : kOutOfRange, (16 * rr * n + 4 * m) * 8) : 0.0;
            }
#pragma unroll
            for (int m = 0; m < UQ; ++m) {
                const bool ok = rowOk && (4 * m + 3 < NU || 4 * m + part < NU);
                bvN[rr][m] = BufferLoad(rsJ, ok ? vj : kOutOfRange, (16 * rr * n + NX + 4 * m) * 8);
            }
            b0N[rr] = BufferLoad(rsB, rowOk && part == 0 ? q * 8 : kOutOfRange, 16 * rr * 8);
        }
    };
    requestForward(0);
    for (int k = 0; k < N; ++k) {
        double gv[RU][XQ], av[RX][XQ], bv[RX][UQ], b0[RX];
#pragma unroll
        for (int rr = 0; rr < RU; ++rr)
#pragma unroll
            for (int m = 0; m < XQ; ++m) gv[rr][m] = gvN[rr][m];
#pragma unroll
        for (int rr = 0; rr < RX; ++rr) {
#pragma unroll
            for (int m = 0; m < XQ; ++m) av[rr][m] = avN[rr][m];
#pragma unroll
            for (int m = 0; m < UQ; ++m) bv[rr][m] = bvN[rr][m];
            b0[rr] = b0N[rr];
        }
        if (k + 1 < N) requestForward(k + 1);
        double xs[XQ];
#pragma unroll
        for (int m = 0; m < XQ; ++m) xs[m] = dxv[part + 4 * m < XE ? part + 4 * m : XE - 1];  // (a term beyond XE has a zero coefficient)
#pragma unroll
        for (int rr = 0; rr < RU; ++rr) {
            const int i = 16 * rr + q;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < XQ; ++m) s = __builtin_fma(gv[rr][m], xs[m], s);
            s += QuadPermuteF64<0xB1>(s);
            s += QuadPermuteF64<0x4E>(s);
            if (i < NU && part == 0) {
                duv[i] = s;
                a.dU.at(inst, k, i) = s;
            }
        }
        WaveLdsFence();
        double us[UQ];
#pragma unroll
        for (int m = 0; m < UQ; ++m) us[m] = duv[part + 4 * m < NU ? part + 4 * m : NU - 1];
        double next[RX];
#pragma unroll
        for (int rr = 0; rr < RX; ++rr) {
            double s = b0[rr];
#pragma unroll
            for (int m = 0; m < XQ; ++m) s = __builtin_fma(av[rr][m], xs[m], s);
#pragma unroll
            for (int m = 0; m < UQ; ++m) s = __builtin_fma(bv[rr][m], us[m], s);
            s += QuadPermuteF64<0xB1>(s);
            s += QuadPermuteF64<0x4E>(s);
            next[rr] = s;
        }
        WaveLdsFence();  // every lane has read dx_k
#pragma unroll
        for (int rr = 0; rr < RX; ++rr) {
            const int i = 16 * rr + q;
            if (i < NX && part == 0) {
                dxv[i] = next[rr];
                a.dX.at(inst, k + 1, i) = next[rr];
            }
        }
        WaveLdsFence();
    }
    mark(7);  // forward pass
    if (a.status && lane == 0) a.status[inst] = failed;
    if constexpr (CLOCKS)
        if (inst == 0 && lane == 0)
            printf("[riccati wave clocks %d+%d, cycles per knot] products %llu, operands to registers %llu, tableau + elimination %llu, back substitution %llu, gains %llu, cost-to-go %llu, operand wait %llu; forward pass %llu per knot\n",
                   NX, NU, clocks[0] / N, clocks[1] / N, clocks[2] / N, clocks[3] / N, clocks[4] / N, clocks[5] / N, clocks[6] / N, clocks[7] / N);
}

/// WAVES_PER_EU resident wavefronts per SIMD: the register budget of the instantiation (512 / WAVES_PER_EU).
template <int NX, int NU, int WAVES_PER_EU, bool CLOCKS = false>
__global__ __launch_bounds__(64, WAVES_PER_EU) void RiccatiWaveKernel(const RiccatiArgs a) {
    RiccatiWaveBody<NX, NU, CLOCKS>(a);
}

/// Dynamic LDS of one wavefront of the <NX, NU> instantiation, and whether the sizes fit the kernel at all (tableau [R | H_ux h_u]: one lane per column).
constexpr bool RiccatiWaveFits(int nx, int nu) { return nx >= 1 && nu >= 1 && nx + nu + 1 <= 64; }

}  // namespace
}  // namespace ungar_amd::kernels
