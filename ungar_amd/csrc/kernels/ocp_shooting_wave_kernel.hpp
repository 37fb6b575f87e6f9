// ungar_amd :: one-wavefront assembly kernel of the batched SQP for stage problems WITH equality rows (ocp_shooting.hip): the
// kernel TEMPLATE <NZ, NU, NE>, instantiated by the library for the reference's quadruped OCP (25 + 24, 16 rows) and by the kernel factory
// (runtime/kernel_jit.cpp) for the sizes any other problem declares -- reference optimization/concepts.hpp:153-262: the optimiser takes ANY problem.
#pragma once

#include <hip/hip_runtime.h>

#include "ocp_barrier.hpp"
#include "ocp_shooting.hpp"

namespace ungar_amd::kernels {
namespace {


template <int CTRL>
__device__ __forceinline__ double QuadPermute(double v) {  // DPP quad_perm: register to register
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned long long ReadLaneU64(unsigned long long v, int sourceLane) {  // sourceLane wave-uniform
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), sourceLane), hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), sourceLane);
    return (static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | static_cast<unsigned>(lo);
}

__device__ __forceinline__ double ReadLane(double v, int sourceLane) {
    return __longlong_as_double(static_cast<long long>(ReadLaneU64(static_cast<unsigned long long>(__double_as_longlong(v)), sourceLane)));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double DppOrZero(double v) {  // lanes without a source (or outside ROW_MASK) read +0.0, the identity of the max below
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double MaxOfNonNegative(double x, double y) {  // the instruction itself: fmax() canonicalises both operands first (two more v_max_f64)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

/// Largest of the 64 lanes' values of TWO quantities at once (the two chains fill each other's issue gaps), left in scalar registers.  The values are bit
/// patterns of NON-NEGATIVE finite doubles, which order like the doubles: v_max_f64 returns one of its operands unchanged (FP64 denormals are kept), so the
/// result is the integer maximum -- three instructions per stage instead of five.  Stages: butterfly inside each 16-lane DPP row, then row_bcast:15 /
/// row_bcast:31 carry the row maxima forward; lane 63 holds the maximum of the wavefront.
__device__ __forceinline__ void WaveMaxPair(unsigned long long& x, unsigned long long& y) {
    double u = __longlong_as_double(static_cast<long long>(x)), v = __longlong_as_double(static_cast<long long>(y));
#define UNGAR_MAX_STAGE(CTRL, ROWS)              \
    {                                            \
        const double su = DppOrZero<CTRL, ROWS>(u), sv = DppOrZero<CTRL, ROWS>(v); \
        u = MaxOfNonNegative(u, su);             \
        v = MaxOfNonNegative(v, sv);             \
    }
    UNGAR_MAX_STAGE(0xB1, 0xF)   // quad_perm [1, 0, 3, 2]
    UNGAR_MAX_STAGE(0x4E, 0xF)   // quad_perm [2, 3, 0, 1]
    UNGAR_MAX_STAGE(0x141, 0xF)  // row_half_mirror
    UNGAR_MAX_STAGE(0x140, 0xF)  // row_mirror: every lane of a row holds the row's maximum
    UNGAR_MAX_STAGE(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
    UNGAR_MAX_STAGE(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef UNGAR_MAX_STAGE
    x = ReadLaneU64(static_cast<unsigned long long>(__double_as_longlong(u)), 63);
    y = ReadLaneU64(static_cast<unsigned long long>(__double_as_longlong(v)), 63);
}

/// The same for ONE quantity (the kernel with three nodes per SIMD: its issue gaps belong to the other wavefronts).
__device__ __forceinline__ unsigned long long WaveMaxKey(unsigned long long x) {
    double u = __longlong_as_double(static_cast<long long>(x));
#define UNGAR_MAX_STAGE(CTRL, ROWS) u = MaxOfNonNegative(u, DppOrZero<CTRL, ROWS>(u));
    UNGAR_MAX_STAGE(0xB1, 0xF)   // quad_perm [1, 0, 3, 2]
    UNGAR_MAX_STAGE(0x4E, 0xF)   // quad_perm [2, 3, 0, 1]
    UNGAR_MAX_STAGE(0x141, 0xF)  // row_half_mirror
    UNGAR_MAX_STAGE(0x140, 0xF)  // row_mirror: every lane of a row holds the row's maximum
    UNGAR_MAX_STAGE(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
    UNGAR_MAX_STAGE(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef UNGAR_MAX_STAGE
    return ReadLaneU64(static_cast<unsigned long long>(__double_as_longlong(u)), 63);
}

__device__ __forceinline__ int WaveMinInt(int v) {  // smallest of the 64 lanes' values (same stages as above), wave-uniform
#define UNGAR_MIN_STAGE(CTRL, ROWS) v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROWS, 0xF, false));
    UNGAR_MIN_STAGE(0xB1, 0xF)
    UNGAR_MIN_STAGE(0x4E, 0xF)
    UNGAR_MIN_STAGE(0x141, 0xF)
    UNGAR_MIN_STAGE(0x140, 0xF)
    UNGAR_MIN_STAGE(0x142, 0xA)
    UNGAR_MIN_STAGE(0x143, 0xC)
#undef UNGAR_MIN_STAGE
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned WaveOr(unsigned v) {  // bitwise OR of the 64 lanes' values, wave-uniform
#define UNGAR_OR_STAGE(CTRL, ROWS) v |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROWS, 0xF, true));
    UNGAR_OR_STAGE(0xB1, 0xF)
    UNGAR_OR_STAGE(0x4E, 0xF)
    UNGAR_OR_STAGE(0x141, 0xF)
    UNGAR_OR_STAGE(0x140, 0xF)
    UNGAR_OR_STAGE(0x142, 0xA)
    UNGAR_OR_STAGE(0x143, 0xC)
#undef UNGAR_OR_STAGE
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

constexpr int kBlock = 64;
constexpr int kRegisterRows = 16;  // equality rows of a node the one-wavefront Gauss-Jordan keeps in registers

__device__ __forceinline__ const double* RowOf(const double* rows, const ShootingDims& d, long long b, int k) {
    return rows + (b * (d.N + 1) + k) * static_cast<long long>(d.nv());
}

/// Barrier terms of ONE wavefront's node into the packed image R of W_e (column `last` = the linear terms):  W += J_h^T diag(b''(-h)) J_h,  w -= J_h^T b'(-h),
/// straight from the sparse inequality Jacobian -- lane e < nnz holds entry e (myRow < 0: none; the entries of a row are consecutive, columns ascending).
/// One lane per PAIR (e1, e2 >= e1) of a row where the pairs fit a wavefront (the quadruped: 28 entries in 12 rows, 46 pairs): entry e1 owns the pairs
/// [before(e1), before(e1) + partners(e1)) -- a prefix sum over the lanes (DPP), every entry writes its index into its pairs' slots of a 64-word LDS table and
/// every pair reads its slot (a loop over the entries with a v_readlane each was ~2 k cycles of this lone wavefront).  Then row by row, in order (rows share
/// targets): the LDS instructions of one wavefront execute in order, so the next row's reads see these writes without a barrier.
template <class Tri>
__device__ __forceinline__ void WaveBarrierTerms(double* R, const Tri& tri, int last, int nh, const double* d1, const double* d2, int* pairTable, int lane, int nnz, int myRow, int myCol,
                                                 double myValue, const int* patternCols, const double* nodeValues) {
    auto fence = [] { asm volatile("" ::: "memory"); };
    const int mine = myRow >= 0 ? lane : -1;
    const int nextRow = __shfl_down(myRow, 1);
    const unsigned long long rowEnds = __ballot(mine >= 0 && (mine == nnz - 1 || nextRow != myRow));
    const int partners = mine >= 0 ? __ffsll(static_cast<unsigned long long>(rowEnds >> mine)) : 0;  // entries of the same row from this one on
    int before = partners;
#define UNGAR_SCAN_STAGE(CTRL, ROWS) before += __builtin_amdgcn_update_dpp(0, before, CTRL, ROWS, 0xF, false);
    UNGAR_SCAN_STAGE(0x111, 0xF)  // row_shr:1
    UNGAR_SCAN_STAGE(0x112, 0xF)  // row_shr:2
    UNGAR_SCAN_STAGE(0x114, 0xF)  // row_shr:4
    UNGAR_SCAN_STAGE(0x118, 0xF)  // row_shr:8
    UNGAR_SCAN_STAGE(0x142, 0xA)  // row_bcast:15 into rows 1 and 3
    UNGAR_SCAN_STAGE(0x143, 0xC)  // row_bcast:31 into rows 2 and 3
#undef UNGAR_SCAN_STAGE
    const int pairs = __builtin_amdgcn_readlane(before, 63);
    before -= partners;  // (exclusive)
    if (pairs <= 64) {
        for (int q = 0; __ballot(q < partners) != 0ull; ++q)
            if (q < partners) pairTable[before + q] = mine | (q << 8);
        fence();
        const bool havePair = lane < pairs;
        const int slot = havePair ? pairTable[lane] : 0;
        const int first = slot & 255, second = first + (slot >> 8);
        const int pairRowAny = __shfl(myRow, first), c1 = __shfl(myCol, first), c2 = __shfl(myCol, second);
        const double v1 = __shfl(myValue, first), v2 = __shfl(myValue, second);
        const int pairRow = havePair ? pairRowAny : -1, target = havePair ? tri(c1, c2) : 0, gTarget = tri(myCol, last);
        const double d1Mine = myRow >= 0 ? d1[myRow] : 0.0, d2Mine = havePair ? d2[pairRow] : 0.0;
        for (int j = 0; j < nh; ++j) {
            const bool g = myRow == j, ww = pairRow == j;
            double g0 = 0.0, w0 = 0.0;
            if (g) g0 = R[gTarget];
            if (ww) w0 = R[target];
            if (g) R[gTarget] = __builtin_fma(-d1Mine, myValue, g0);
            if (ww) R[target] = __builtin_fma(d2Mine * v1, v2, w0);
            fence();
            __builtin_amdgcn_wave_barrier();
        }
    } else {  // (patterns whose pairs do not fit a wavefront: one lane per entry, its partners one after the other)
        for (int j = 0; j < nh; ++j) {
            if (myRow == j) {
                R[tri(myCol, last)] -= d1[j] * myValue;
                for (int q = 0; q < partners; ++q) R[tri(myCol, patternCols[mine + q])] += d2[j] * myValue * nodeValues[mine + q];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

/// Doubles of the one LDS region: the tableau [E | e], packed W_e (with its homogeneous row), [A|B]_e -- whichever is largest.
constexpr int ShootingAssembleWaveImage(int nz, int nu, int ne) {
    const int nd = nz + nu, nh = nd + 1, tri = nh * (nh + 1) / 2, ab = nz * nh, tableau = ne * nh;
    return tri > ab ? (tri > tableau ? tri : tableau) : (ab > tableau ? ab : tableau);
}

/// ONE WAVEFRONT per node, for stage problems whose equality tableau fits the lanes (NE <= 16 rows, ND + 1 <= 64 columns: the quadruped's 16 x 50) -- the
/// workgroup kernel above spends ~9.8 k vector instructions per node, most of them index arithmetic in front of LDS operands that every 16 x 16 tile of the
/// substitution fetches again (18 tiles x 5 operands x 4 k-steps), and a workgroup barrier between its sections.  Here
///   * the quadratic and the linear part travel together in HOMOGENEOUS coordinates z_e = [z; 1]:  W_e = [W w; w^T 0],  [A|B]_e = [A|B  b],  G_e = [G | g0] --
///     the substitution  z = (I - E_J G_e) z_e  is then  W_e' = W_e - W_e[:,J] G_e - G_e^T W_e[J,:] + G_e^T W_JJ G_e,  [A|B]_e' = [A|B]_e - [A|B]_e[:,J] G_e,  with
///     w' and b' as column ND of the tiles (three separate matrix-vector sections before);
///   * the operands of the products are fetched ONCE per (tile row, k-step): by symmetry -W_e[:,J] in A layout is also -W_e[J,:] in B layout, and G_e in B layout
///     is also G_e^T in A layout (lane 16 k + i in both);
///   * V = W_JJ G_e stays in the accumulators it was computed in: element r of a tile is the B operand of k-step r;
///   * ONE LDS region holds, one after the other, the tableau [C | D | e] (scatter target of the sparse equality Jacobian; the elimination itself runs in
///     registers), the reduced tableau (from which G_e is gathered into operand layout), the packed image of W_e (cost Hessian, barrier terms, regularisation;
///     its tiles are fetched one at a time in front of their matrix instructions and go to memory from the accumulators) and the image of [A|B]_e:
///     10.9 KB of LDS and <= 168 registers per node, THREE nodes per SIMD (the serial chains of a lone wavefront -- elimination, barrier terms -- leave the SIMD
///     idle nine cycles in ten: the kernel's rate follows the number of resident wavefronts; with the tableau in a region of its own and every tile of W_e in
///     registers before the region changed hands it was 17 KB, 256 registers, two per SIMD);
///   * results go to memory from the registers (the dummies' identity rows / zero columns are written, never formed).
/// Same pivot rule, same arithmetic for the reduced rows, the pivots and the tiles of W' and [A|B]' as the kernel above (bitwise); w' and b' are summed by the
/// matrix cores in a different order (rounding).
template <int NZ, int NU, int NE, bool CLOCKS = false>
__device__ __forceinline__ void ShootingAssembleWaveBody(const ShootingAssembleArgs& a) {
    using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;
    constexpr int ND = NZ + NU, NH = ND + 1, LD = ND + 1, TD = (NH + 15) / 16, TZ = (NZ + 15) / 16, KS = (NE + 3) / 4;
    constexpr int kImage = ShootingAssembleWaveImage(NZ, NU, NE);  // the tableau, packed W_e, [A|B]_e (row stride NH): whichever is largest
    constexpr int kImagePadded = (kImage + 1) & ~1;
    static_assert(NE <= kRegisterRows && LD <= 64 && TD <= 4, "one wavefront holds the tableau");
    extern __shared__ double lds[];
    const ShootingDims& d = a.d;
    const long long node = blockIdx.x;
    const long long b = node / (d.N + 1);
    const int k = static_cast<int>(node - b * (d.N + 1));
    const int lane = static_cast<int>(threadIdx.x), lj = lane & 15, lk = lane >> 4;
    const int nc = d.nc, nx = d.nx;
    const bool stage = k < d.N;
    double* R = lds;                       // kImagePadded
    double* d1 = R + kImagePadded;         // nh
    double* d2 = d1 + a.nh;                // nh
    unsigned long long* rowScale = reinterpret_cast<unsigned long long*>(d2 + a.nh);  // NE
    int* pivL = reinterpret_cast<int*>(rowScale + NE);                                  // NE
    int* pairTable = pivL + NE;                                                         // 64: pairs of inequality-Jacobian entries (barrier terms)
    auto tri = [](int r, int c) { return ((r * (2 * NH + 1 - r)) >> 1) + (c - r); };  // r <= c < NH
    auto sym = [&tri](int r, int c) { return r <= c ? tri(r, c) : tri(c, r); };
    auto fence = [] { asm volatile("" ::: "memory"); };  // (for the compiler: the LDS instructions of one wavefront execute in order)
    const long long nodeOff = node, stageOff = b * d.N + k;
    unsigned long long marks[CLOCKS ? 12 : 1];  // UNGAR_AMD_ASSEMBLE_CLOCKS=1: cycles of the sections of two nodes, printed by their first lane
    int markCount = 0;
    auto mark = [&] {
        if constexpr (CLOCKS) {
            __builtin_amdgcn_sched_barrier(0);
            marks[markCount++] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    mark();
    // ---- requests: every sparse value of the node, up to kSlots per lane and output (the launcher checks the patterns against these bounds)
    constexpr int kSlotsH = 4, kSlotsF = 4, kSlotsC = 2, kSlotsE = 4;
    struct Entry {
        int target;  // < 0: none
        double value;
    };
    struct Raw {
        int r, c;
        double value;
        bool valid;
    };
    // (MUBUF loads against resources of exactly the pattern's size: a lane beyond the pattern -- or a whole output that is absent -- reads zeros from the range
    // check, unconditionally, and nothing is computed from a loaded value before every request of its batch is out.  Written with conditions, every slot became a
    // branch around its loads with a full wait behind it: 15 round trips to memory in a row, 24-43 k of a node's ~85 k cycles.)
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
    auto resolve = [](const Raw& f, auto targetOf) { return Entry{f.valid ? targetOf(f.r, f.c) : -1, f.value}; };
    auto hessianTarget = [&](int r, int c) { return r <= c ? tri(r, c) : -1; };
    auto gradientTarget = [&](int, int c) { return tri(c, ND); };
    auto dynamicsTarget = [&](int r, int c) { return (nc + r) * NH + nc + c; };
    auto carryTarget = [&](int r, int c) { return r * NH + nc + c; };
    auto equalityTarget = [&](int r, int c) { return r * LD + c; };
    // first batch: what the tableau, the image of W_e and the vectors need (the dynamics' values are asked for when the image of W_e has gone to memory: held
    // from here on, they were 24 registers across the sections that need the most)
    Raw rH[kSlotsH], rE[kSlotsE];
#pragma unroll
    for (int s = 0; s < kSlotsE; ++s) rE[s] = request(a.pe, a.eJ, stage, s);
    const double residual = loadDouble(resourceOver(a.e ? a.e + nodeOff * NE : nullptr, stage ? NE * 8 : 0), lane * 8);
#pragma unroll
    for (int s = 0; s < kSlotsH; ++s) rH[s] = request(a.pH, a.lH, true, s);
    const Raw rG = request(a.pg, a.lg, true, 0);
    const double hMine = loadDouble(resourceOver(a.h ? a.h + nodeOff * a.nh : nullptr, stage ? a.nh * 8 : 0), lane * 8);
    const Raw rI = request(a.ph, a.hJ, stage && a.nh > 0, 0);
    // lane i < NZ: dz0[i] = [0; x_m - x_0]  (a lane below nc reaches before the values: out of range, zero)
    const double xmMine = loadDouble(resourceOver(a.xm + b * nx, k == 0 ? nx * 8 : 0), (lane - nc) * 8);
    const double row0Mine = loadDouble(resourceOver(RowOf(a.rows, d, b, 0), k == 0 ? NZ * 8 : 0), lane * 8);
    double* W = a.W + nodeOff * ND * ND;
    double* w = a.w + nodeOff * ND;
    auto masked = [](double v, bool keep) { return __longlong_as_double(keep ? __double_as_longlong(v) : 0ll); };
    // ---- the tableau [C | D | e] in the region, then in registers: Gauss-Jordan, lane = column (the elimination job of the kernel above, statement by statement)
    unsigned long long taken = 0ull;  // inputs that are pivots already (wave-uniform)
    unsigned pivotRows = 0u;          // rows that took a pivot (wave-uniform)
    double gm[KS][TD];                // G_e in B layout = G_e^T in A layout: row 4 ks + lk, column 16 tj + lj
    bool stepHasPivot[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        stepHasPivot[ks] = false;
#pragma unroll
        for (int tj = 0; tj < TD; ++tj) gm[ks][tj] = 0.0;
    }
    if (stage) {
        for (int i = lane; i < NE * LD; i += 64) R[i] = 0.0;
        if (lane < NE) rowScale[lane] = 0ull;
        fence();
        auto offerScale = [&](int row, double value) {  // (non-negative doubles order like their bit patterns)
            const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(fabs(value)));
            if (bits) atomicMax(&rowScale[row], bits);
        };
#pragma unroll
        for (int s = 0; s < kSlotsE; ++s)
            if (rE[s].valid) {
                R[equalityTarget(rE[s].r, rE[s].c)] = rE[s].value;
                offerScale(rE[s].r, rE[s].value);
            }
        if (lane < NE) {
            R[lane * LD + ND] = residual;
            offerScale(lane, residual);
        }
        fence();
        double t[NE];
#pragma unroll
        for (int r = 0; r < NE; ++r) t[r] = lane < LD ? R[r * LD + lane] : 0.0;
        mark();  // 1: first requests answered, tableau in registers
        unsigned nonZeroRows = 0u;
#pragma unroll
        for (int r = 0; r < NE; ++r) nonZeroRows |= (__double_as_longlong(t[r]) << 1) != 0ll ? 1u << r : 0u;
        nonZeroRows = WaveOr(nonZeroRows);
        const double scaleOfMyRow = lane < NE ? __longlong_as_double(static_cast<long long>(rowScale[lane])) : 0.0;
        int myPivot = -1;  // lane i < NE: pivot input of row i
        const int myInput = lane - NZ;
        const bool inputLane = lane >= NZ && lane < ND;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if ((nonZeroRows >> i) & 1u) {  // (uniform; an identically-zero row takes no pivot: -1)
                // (the largest entry of the row decides in three places; each is a question every lane answers for its own entry -- x -> 1e-12 x rounds monotonically,
                // so  best <= 1e-12 max_l |t_l|  is  best <= 1e-12 |t_l| for some l -- and one reduction over the lanes, 20 vector instructions, is left of two)
                const double magnitude = fabs(t[i]);
                const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(magnitude));
                const bool candidate = bits != 0ull && inputLane && !((taken >> (myInput & 63)) & 1ull);
                const unsigned long long key = WaveMaxKey(candidate ? (bits & ~0xFFull) | static_cast<unsigned long long>(255 - myInput) : 0ull);
                const double best = __longlong_as_double(static_cast<long long>(key & ~0xFFull));
                int j = key ? 255 - static_cast<int>(key & 0xFFull) : -1;
                if (__ballot(magnitude > 1e-12 * ReadLane(scaleOfMyRow, i)) == 0ull) {  // row maximum <= 1e-12 x the scale of the row as it was given
                    j = -1;
                } else {
                    if (j >= 0 && __ballot(best <= 1e-12 * magnitude) != 0ull) j = -2;
                    if (j == -1 && __ballot(bits != 0ull) != 0ull) j = -2;
                }
                if (lane == i) myPivot = j;
                if (j >= 0) {
                    taken |= 1ull << j;
                    pivotRows |= 1u << i;
                    const int J = NZ + j;
                    // (hardware reciprocal + two Newton steps instead of the IEEE division -- the pivots' reciprocals are the serial chain of the elimination; the
                    // multipliers of a pivot -- column J of the tableau, 16 registers of lane J -- reach the lanes through the LDS crossbar (ds_bpermute), in batches:
                    // as v_readlane pairs they were 30 of a pivot's 137 vector instructions, and the vector unit is what three resident wavefronts compete for)
                    const double pivot = ReadLane(t[i], J);
                    double rpiv = __builtin_amdgcn_rcp(pivot);
                    rpiv = __builtin_fma(__builtin_fma(-pivot, rpiv, 1.0), rpiv, rpiv);
                    rpiv = __builtin_fma(__builtin_fma(-pivot, rpiv, 1.0), rpiv, rpiv);
                    const double p = lane == J ? 1.0 : t[i] * rpiv;
#pragma unroll
                    for (int r0 = 0; r0 < NE; r0 += 8) {
                        double m[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (r0 + q < NE && r0 + q != i) m[q] = __shfl(t[r0 + q], J);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (r0 + q < NE && r0 + q != i) t[r0 + q] = t[r0 + q] - m[q] * p;  // (lane J: m - m * 1 = +0 exactly)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    t[i] = p;
                }
            }
        }
        mark();  // 2: elimination
        // the reduced rows, their residuals and pivots: to memory from the registers, and back to the region the operand G_e is gathered from
        double* E = a.E + stageOff * NE * ND;
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            if (lane < ND) E[r * ND + lane] = t[r];
            if (lane < LD) R[r * LD + lane] = t[r];
            if (lane == ND) a.er[stageOff * NE + r] = t[r];
        }
        if (lane < NE) {
            a.pivots[stageOff * NE + lane] = myPivot;
            pivL[lane] = myPivot;
        }
        fence();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {  // row t = 4 ks + lk of the tableau (rows without a pivot are masked out)
            const int tRow = 4 * ks + lk;
            stepHasPivot[ks] = tRow < NE && ((pivotRows >> tRow) & 1u) != 0u;
#pragma unroll
            for (int tj = 0; tj < TD; ++tj) {
                const int col = 16 * tj + lj;
                gm[ks][tj] = masked(R[(tRow < NE ? tRow : NE - 1) * LD + (col < LD ? col : LD - 1)], col < LD && stepHasPivot[ks]);
            }
        }
        fence();
        mark();  // 3: reduced rows stored, G_e in operand layout
    }
    // ---- the region becomes the packed image of W_e
    for (int i = lane; i < kImagePadded; i += 64) R[i] = 0.0;
    if (stage && lane < a.nh) {
        d1[lane] = BarrierD1(a.barrier, -hMine);
        d2[lane] = BarrierD2(a.barrier, -hMine);
    }
    fence();
#pragma unroll
    for (int s = 0; s < kSlotsH; ++s) {
        const Entry e = resolve(rH[s], hessianTarget);
        if (e.target >= 0) R[e.target] = e.value;
    }
    {
        const Entry e = resolve(rG, gradientTarget);
        if (e.target >= 0) R[e.target] = e.value;
    }
    if (!stage) {  // knot N: the terminal cost as it is (regularised over its state), no dynamics, no rows
        fence();
        if (lane >= nc && lane < NZ) R[tri(lane, lane)] += a.regularization;
        fence();
        const float ndInv = 1.0f / static_cast<float>(ND);
        for (int idx = lane; idx < ND * ND; idx += 64) {
            const int r = static_cast<int>((static_cast<float>(idx) + 0.5f) * ndInv);
            if (r <= idx - r * ND) W[idx] = R[tri(r, idx - r * ND)];
        }
        if (lane < ND) w[lane] = R[tri(lane, ND)];
        return;
    }
    const int inequalityRow = rI.valid ? rI.r : -1, inequalityColumn = rI.valid ? rI.c : 0;
    const double inequalityValue = rI.value;
    const double dz0 = lane >= nc && lane < NZ ? xmMine - row0Mine : 0.0;
    if (k == 0 && lane < NZ) a.dz0[b * NZ + lane] = dz0;
    fence();
    mark();  // 4: image of W_e filled
    // ---- barrier terms  W += J_h^T diag(b''(-h)) J_h,  w -= J_h^T b'(-h)  from the sparse inequality Jacobian
    if (a.nh > 0) WaveBarrierTerms(R, tri, ND, a.nh, d1, d2, pairTable, lane, a.ph.nnz, inequalityRow, inequalityColumn, inequalityValue, a.ph.cols, a.hJ + nodeOff * a.ph.nnz);
    if (lane >= nc && lane < ND) R[tri(lane, lane)] += a.regularization;  // the reference's 1e-6 I over its decision variables (soft_sqp.hpp:149-151)
    fence();
    mark();  // 5: barrier terms, regularisation
    auto isPivot = [&](int c) { return c >= NZ && c < ND && ((taken >> ((c - NZ) & 63)) & 1ull) != 0ull; };
    // ---- operands out of the image, once per (tile row, k-step): J_t the pivot input of row t = 4 ks + lk
    int pivotOfStep[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int tRow = 4 * ks + lk;
        const int j = pivL[tRow < NE ? tRow : NE - 1];
        pivotOfStep[ks] = NZ + (stepHasPivot[ks] ? j : 0);
    }
    const bool anyPivot = pivotRows != 0u;
    double a1m[TD][KS];  // -W_e[16 ti + lj][J_t] in A layout = -W_e[J_t][16 tj + lj] in B layout
    double wjj[KS];      // W[J_lj][J_t]
    const int pivotOfMyRow = NZ + (lj < NE && ((pivotRows >> lj) & 1u) ? pivL[lj < NE ? lj : 0] : 0);
    const bool myRowHasPivot = lj < NE && ((pivotRows >> lj) & 1u) != 0u;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int ti = 0; ti < TD; ++ti) {
            const int row = 16 * ti + lj;
            const bool in = row < NH && stepHasPivot[ks];
            a1m[ti][ks] = masked(-R[sym(row < NH ? row : NH - 1, pivotOfStep[ks])], in);
        }
        wjj[ks] = masked(R[sym(pivotOfMyRow, pivotOfStep[ks])], myRowHasPivot && stepHasPivot[ks]);
    }
    // ---- V = W_JJ G_e; the B operand of the second product of a tile is V_t[c] - W[J_t][c]
    double vb[TD][KS];
    {
        f64x4 Vt[TD];
#pragma unroll
        for (int tj = 0; tj < TD; ++tj) Vt[tj] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (anyPivot) {  // (uniform)
#pragma unroll
            for (int tj = 0; tj < TD; ++tj)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Vt[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(wjj[ks], gm[ks][tj], Vt[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < TD; ++tj)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vb[tj][ks] = Vt[tj][ks] + a1m[tj][ks];
    }
    mark();  // 6: operands, V
    // ---- W_e' tile by tile (on and above the diagonal; element r: row 16 ti + lk + 4 r, column 16 tj + lj; below the diagonal of a diagonal tile the transposed
    // entry): out of the image, through the matrix cores, to memory.  The eliminated inputs are decoupled dummies: identity rows / columns, zero gradient.
    bool rowPivot[TD][4], colPivot[TD];
#pragma unroll
    for (int ti = 0; ti < TD; ++ti) {
        colPivot[ti] = isPivot(16 * ti + lj);
#pragma unroll
        for (int r = 0; r < 4; ++r) rowPivot[ti][r] = isPivot(16 * ti + lk + 4 * r);
    }
    Raw rF[kSlotsF], rC[kSlotsC];
    double fMine = 0.0, nextMine = 0.0;
#pragma unroll
    for (int ti = 0; ti < TD; ++ti)  // (tile row by tile row: neighbouring tiles write neighbouring pieces of the same rows of W')
#pragma unroll
        for (int tj = ti; tj < TD; ++tj) {
            if (tj == TD - 1 && ti == (TD > 2 ? TD - 2 : 0)) {
                // second batch of requests, in front of the last two tiles: the operands of the earlier tile rows and columns are dead by now, and the round trip to
                // memory (2-3 k cycles) is over when the image of [A|B]_e is filled
                fence();
#pragma unroll
                for (int s = 0; s < kSlotsF; ++s) rF[s] = request(a.pf, a.fJ, true, s);
#pragma unroll
                for (int s = 0; s < kSlotsC; ++s) rC[s] = request(a.pc, a.cJ, !d.carryInputs, s);
                // lane i < NZ: b[i] = [0; f - x_next]
                fMine = loadDouble(resourceOver(a.f + nodeOff * nx, nx * 8), (lane - nc) * 8);
                nextMine = loadDouble(resourceOver(RowOf(a.rows, d, b, k + 1), NZ * 8), lane * 8);
                fence();
            }
            f64x4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lk + 4 * r, col = 16 * tj + lj;
                const bool in = row < NH && col < NH;
                const int rc = row < NH ? row : NH - 1, cc = col < NH ? col : NH - 1;
                acc[r] = masked(R[ti == tj ? sym(rc, cc) : tri(rc, cc)], in);
            }
            if (anyPivot) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1m[ti][ks], gm[ks][tj], acc, 0, 0, 0);   // - W[a][J_t] G_t[c]
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(gm[ks][ti], vb[tj][ks], acc2, 0, 0, 0);  // + G_t[a] (V_t[c] - W[J_t][c])
                }
                acc = acc + acc2;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lk + 4 * r, col = 16 * tj + lj;
                if (row > col || row >= ND || col > ND) continue;
                const bool dummy = rowPivot[ti][r] || colPivot[tj];
                const double v = dummy ? (row == col ? 1.0 : 0.0) : acc[r];
                if (col == ND) w[row] = v;
                else W[row * ND + col] = v;
            }
        }
    mark();  // 7: W', w' stored
    // ---- the region becomes the image of [A|B]_e = [A|B  b] (row stride NH)
    fence();
    for (int i = lane; i < NZ * NH; i += 64) R[i] = 0.0;
    fence();
#pragma unroll
    for (int s = 0; s < kSlotsF; ++s) {
        const Entry e = resolve(rF[s], dynamicsTarget);
        if (e.target >= 0) R[e.target] = e.value;
    }
    if (d.carryInputs) {
        if (lane < nc) R[lane * NH + NZ + lane] = 1.0;
    } else {
#pragma unroll
        for (int s = 0; s < kSlotsC; ++s) {
            const Entry e = resolve(rC[s], carryTarget);
            if (e.target >= 0) R[e.target] = e.value;
        }
    }
    if (lane < NZ) R[lane * NH + ND] = lane >= nc ? fMine - nextMine : 0.0;
    fence();
    mark();  // 8: image of [A|B]_e
    // ---- [A|B]_e' = [A|B]_e - [A|B]_e[:,J] G_e
    double aAB[TZ][KS];
    f64x4 ABt[TZ][TD];
#pragma unroll
    for (int ti = 0; ti < TZ; ++ti) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int row = 16 * ti + lj;
            aAB[ti][ks] = masked(-R[(row < NZ ? row : NZ - 1) * NH + pivotOfStep[ks]], row < NZ && stepHasPivot[ks]);
        }
#pragma unroll
        for (int tj = 0; tj < TD; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lk + 4 * r, col = 16 * tj + lj;
                ABt[ti][tj][r] = masked(R[(row < NZ ? row : NZ - 1) * NH + (col < NH ? col : NH - 1)], row < NZ && col < NH);
            }
    }
    if (anyPivot) {
#pragma unroll
        for (int ti = 0; ti < TZ; ++ti)
#pragma unroll
            for (int tj = 0; tj < TD; ++tj)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) ABt[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aAB[ti][ks], gm[ks][tj], ABt[ti][tj], 0, 0, 0);
    }
    mark();  // 9: tiles and products of [A|B]_e'
    double* AB = a.AB + stageOff * NZ * ND;
    double* bo = a.b + stageOff * NZ;
#pragma unroll
    for (int ti = 0; ti < TZ; ++ti)
#pragma unroll
        for (int tj = 0; tj < TD; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lk + 4 * r, col = 16 * tj + lj;
                if (row >= NZ || col > ND) continue;
                if (col == ND) bo[row] = ABt[ti][tj][r];
                else AB[row * ND + col] = colPivot[tj] ? 0.0 : ABt[ti][tj][r];
            }
    if constexpr (CLOCKS) {
        __builtin_amdgcn_s_waitcnt(0);
        mark();  // 10: [A|B]', b' stored
        if (lane == 0 && (node == 1 || node == d.batch * (d.N + 1) / 2 + 1))
            printf("[assemble wave clocks] node %lld (%d pivots): requests+tableau %llu, elimination %llu, rows stored+G_e %llu, W image %llu, barrier terms %llu, operands+V %llu, W tiles: products+stores %llu, [A|B] requests+image %llu, [A|B] tiles+products %llu, [A|B] stored %llu; total %llu\n",
                   node, __popc(pivotRows), marks[1] - marks[0], marks[2] - marks[1], marks[3] - marks[2], marks[4] - marks[3], marks[5] - marks[4], marks[6] - marks[5], marks[7] - marks[6], marks[8] - marks[7],
                   marks[9] - marks[8], marks[10] - marks[9], marks[10] - marks[0]);
    }
}

#ifndef UNGAR_ASSEMBLE_WAVE_EU
#define UNGAR_ASSEMBLE_WAVE_EU 3  // (measurement knob: tools/make_shooting_variants.sh)
#endif
template <int NZ, int NU, int NE, bool CLOCKS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(UNGAR_ASSEMBLE_WAVE_EU, UNGAR_ASSEMBLE_WAVE_EU))) void ShootingAssembleWaveKernel(const ShootingAssembleArgs a) {
    ShootingAssembleWaveBody<NZ, NU, NE, CLOCKS>(a);
}

/// Sizes the template can be instantiated for (the tableau [E | e] and the homogeneous row of W_e in one wavefront) and its dynamic LDS.
constexpr bool ShootingAssembleWaveFits(int nz, int nu, int ne) { return ne >= 1 && ne <= kRegisterRows && nz >= 1 && nu >= 1 && nz + nu + 1 <= 64; }
constexpr std::size_t ShootingAssembleWaveLds(int nz, int nu, int ne, int nh) {
    const int padded = (ShootingAssembleWaveImage(nz, nu, ne) + 1) & ~1;
    return (static_cast<std::size_t>(padded) + 2 * static_cast<std::size_t>(nh) + static_cast<std::size_t>(ne)) * sizeof(double) + (static_cast<std::size_t>(ne) + 64) * sizeof(int);
}

}  // namespace
}  // namespace ungar_amd::kernels
