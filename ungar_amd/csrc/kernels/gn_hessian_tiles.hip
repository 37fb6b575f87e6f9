// ungar_amd :: Gauss-Newton contraction  G = J^T diag(d) J  (upper triangle) for UNIT-FASTEST Jacobians,
// one LANE per (node, TILE x TILE block of G), Jacobian rows streamed ONCE through LDS.
//
// Why not the matrix cores: v_mfma_f64_16x16x4_f64 has no rate advantage over the FP64 vector ALU on gfx950 (78.6 TFLOP/s data
// sheet for both, 47 measured for the matrix instruction, tools/mfma_f64_peak.hip) and a 37 x 49 block pads to 40 x 64, so 56 %
// of the issued matrix flops are padding.  Why not one lane per node (gn_hessian_lanes.hip): a lane can
// hold one 7 x 7 block of G in registers, so the 28 blocks of a node are 28 passes over its Jacobian, all through the vector
// memory path (14 loads per 49 multiply-adds; 9.5 GB of L1 / L2 traffic per launch for ANYmal).
//
// This kernel gives the 28 blocks of a node to 28 LANES instead.  A workgroup owns NODES = 16 consecutive nodes; lane
// l = slot * 16 + n holds block `slot` of node n (ANYmal: 28 x 16 = 448 lanes = 7 wavefronts).  The Jacobian is consumed
// row by row,    acc[a][b] += (d_r J[r][A + a]) * J[r][B + b],
// so only a few ROWS of the 16 nodes are ever on chip: STAGE rows (STAGE x cols x 16 doubles, 25 KB for ANYmal) are copied
// global -> registers -> LDS by all lanes (element e of a stage goes to LDS double e: the unit-fastest layout IS the LDS layout,
// both sides fully coalesced / conflict free), double buffered, ONE barrier per stage, the loads of stage s + 1 in flight while
// stage s is contracted.  Every Jacobian byte is read from HBM exactly once and from LDS 2 x 28 / 7 = 8 times at 256 B / clk.
// LDS banks: a column of 16 nodes is 128 B = half of the 64 banks; the two blocks of a 32-lane read group are neighbours in the
// row-major block order, (A, B) and (A, B + 1), so their `a` operands are the same addresses (broadcast) and their `b` operands
// are columns 7 apart -- odd, i.e. the other half of the banks.
// Arithmetic: 49 fused multiply-adds + 7 multiplies per lane and row, 100 % useful except the lower halves of the 7 diagonal
// blocks (11 %).  Budget for 81 920 ANYmal nodes: vector ALU 0.12 ms, LDS 0.03 ms, HBM (2.0 GB) 0.25 ms.
// Output strides are the caller's, as for the lanes kernel: unit-fastest (16 lanes = 128 contiguous bytes per entry) or
// node-major.  Reference analogue: soft_sqp.hpp:257-264.
#include "../runtime/measurement.hpp"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

namespace ungar_amd::kernels {

namespace {

constexpr int kTile = 7;
constexpr int kNodes = 16;

template <int COLS>
struct GnTilesShape {
    static constexpr int side = (COLS + kTile - 1) / kTile;
    static constexpr int slots = side * (side + 1) / 2;
    static constexpr int lanes = slots * kNodes;
    static constexpr int block = (lanes + 63) / 64 * 64;
};

}  // namespace

/// DIAG (tools/gn_tiles_bench.hip only): 1 = no result stores, 2 = no arithmetic, 3 = no global loads, 4 = neither loads nor stores,
/// 5 = arithmetic from LDS only (no loads, stores, staging writes or barriers).
/// STAGE rows per barrier.  COLS is the Jacobian's column count (compile time: it fixes the workgroup shape).
/// PERSISTENT launch: gridDim.x workgroups (one per compute unit) walk the node groups g = blockIdx.x, blockIdx.x + gridDim.x, ...
/// as ONE pipeline: the loads of the next stage -- the first stage of the NEXT group included -- are in flight while a stage is
/// contracted, and a finished group's blocks are stored after the next stage's loads have been issued, so neither the load
/// latency at the head of a group nor the store drain at its tail is exposed (a workgroup per group left both in the open:
/// 0.52 ms against 0.44 ms of pure streaming and 0.36 ms of arithmetic + stores for 81 920 ANYmal nodes).
template <int COLS, int STAGE, bool WEIGHTED, int DIAG = 0>
__global__ __launch_bounds__(GnTilesShape<COLS>::block) void GnHessianTilesKernel(const double* __restrict__ jac, long long jes, const double* __restrict__ d, long long des,
                                                                                 double* __restrict__ g, long long ges, long long gns, long long ldg, int rows,
                                                                                 long long count, long long tileNodes = 0, long long jTile = 0, long long dTile = 0,
                                                                                 long long gTile = 0) {
    using Shape = GnTilesShape<COLS>;
    constexpr int BLOCK = Shape::block;
    constexpr int ROW = COLS * kNodes;                       // doubles of one Jacobian row of the node group in LDS
    constexpr int STAGE_ELEMS = STAGE * ROW;                 // doubles per stage (Jacobian part)
    constexpr int PER_LANE = (STAGE_ELEMS + BLOCK - 1) / BLOCK;
    constexpr bool EXACT = PER_LANE * BLOCK == STAGE_ELEMS;  // every lane stages exactly PER_LANE elements (ANYmal: 7 x 448 = 4 x 784)
    constexpr int PAD = (Shape::side * kTile - COLS) * kNodes;  // blocks of the last block column read past the row: keep it inside the allocation
    constexpr int BUF = STAGE_ELEMS + PAD + STAGE * kNodes;      // + the weights of the stage
    constexpr int W_PER_LANE = (STAGE * kNodes + BLOCK - 1) / BLOCK;  // weights of a stage: STAGE x 16 values
    __shared__ double lds[2 * BUF];

    const int t = threadIdx.x;
    const int n = t & (kNodes - 1);
    const int slot = t >> 4;
    const bool computes = slot < Shape::slots;
    const long long groups = (count + kNodes - 1) / kNodes;

    // unrank the slot: blocks (A <= B) of the upper block triangle in row-major order
    int A = 0, B = 0;
    {
        int s = computes ? slot : 0;
        while (s >= Shape::side - A) {
            s -= Shape::side - A;
            ++A;
        }
        B = A + s;
    }
    // One LDS address register per operand: the compiler would otherwise pair neighbouring columns into ds_read2_b64, which has
    // half the rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table).
    unsigned aPtr[kTile], bPtr[kTile];
#pragma unroll
    for (int i = 0; i < kTile; ++i) {
        aPtr[i] = static_cast<unsigned>(((A * kTile + i) * kNodes + n) * sizeof(double));
        bPtr[i] = static_cast<unsigned>(((B * kTile + i) * kNodes + n) * sizeof(double));
        asm volatile("" : "+v"(aPtr[i]));
        asm volatile("" : "+v"(bPtr[i]));
    }
    const char* ldsBytes = reinterpret_cast<const char*>(lds);

    // staging: element e = t + i * BLOCK of a stage is row-column index e / 16 of node e % 16
    const long long total = static_cast<long long>(rows) * COLS;  // row-column indices of the whole Jacobian
    const int rc0 = t >> 4;
    const int stages = (rows + STAGE - 1) / STAGE;
    double pre[PER_LANE];
    double preW[W_PER_LANE];
    unsigned preZero = 0;  // bit i: element i of the stage in flight lies past the last Jacobian row
    static_assert(PER_LANE <= 32);
    auto fetch = [&](long long group, int stage) {
        preZero = 0;
        const long long node = group * kNodes + n;
        long long nodeRead = node < count ? node : count - 1;  // out-of-range lanes read a valid node and discard
        const double* __restrict__ dsrc = d;
        const double* __restrict__ src = jac;
        if (tileNodes > 0) {  // [tile][element][node in tile]: operands of `tileNodes` nodes each (a multiple of 16), tile strides jTile / dTile
            const long long tile = (group * kNodes) / tileNodes;
            nodeRead -= tile * tileNodes;
            src += tile * jTile;
            dsrc += tile * dTile;
        }
        src += nodeRead;
        const long long base = static_cast<long long>(stage) * STAGE * COLS;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            const long long rc = base + rc0 + i * (BLOCK / kNodes);
            if constexpr (DIAG >= 3) {
                pre[i] = 1.0 + i;
            } else {  // branch-free: out-of-range elements read the last valid one; publish() replaces them by zero (a select here
                      // would wait for the load it has just issued)
                pre[i] = __builtin_nontemporal_load(src + (rc < total ? rc : total - 1) * jes);
                if (rc >= total) preZero |= 1u << i;
            }
        }
        if constexpr (WEIGHTED) {
#pragma unroll
            for (int i = 0; i < W_PER_LANE; ++i) {
                const long long r = static_cast<long long>(stage) * STAGE + rc0 + i * (BLOCK / kNodes);
                preW[i] = ((t + i * BLOCK) < STAGE * kNodes && r < rows) ? dsrc[r * des + nodeRead] : 0.0;
            }
        }
    };
    auto publish = [&](int buf) {
        double* __restrict__ dst = lds + buf * BUF;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i)
            if (EXACT || (t + i * BLOCK) < STAGE_ELEMS) dst[t + i * BLOCK] = (preZero >> i) & 1u ? 0.0 : pre[i];
        if constexpr (WEIGHTED) {
#pragma unroll
            for (int i = 0; i < W_PER_LANE; ++i)
                if ((t + i * BLOCK) < STAGE * kNodes) dst[STAGE_ELEMS + PAD + t + i * BLOCK] = preW[i];
        }
    };

    double acc[kTile][kTile];
    auto clear = [&] {
#pragma unroll
        for (int a = 0; a < kTile; ++a)
#pragma unroll
            for (int b = 0; b < kTile; ++b) acc[a][b] = 0.0;
    };
    auto store = [&](long long group) {
        const long long node = group * kNodes + n;
        if (!computes || node >= count) return;
        if ((DIAG == 1 || DIAG >= 4) && acc[0][0] != 12345.678) return;
        // lane part of the address once (it changes with the group, so it is not hoisted out of the pipeline loop into 49 live
        // register pairs); the per-entry part (a * ldg + b) * ges is uniform and stays on the scalar unit
        long long nodeOut = node;
        double* __restrict__ gt = g;
        if (tileNodes > 0) {
            const long long tile = (group * kNodes) / tileNodes;
            nodeOut -= tile * tileNodes;
            gt += tile * gTile;
        }
        double* __restrict__ gn = gt + (nodeOut * gns + (static_cast<long long>(A * kTile) * ldg + B * kTile) * ges);
#pragma unroll
        for (int a = 0; a < kTile; ++a)
#pragma unroll
            for (int b = 0; b < kTile; ++b) {
                const int ga = A * kTile + a, gb = B * kTile + b;
                if (ga < COLS && gb < COLS && ga <= gb) __builtin_nontemporal_store(acc[a][b], gn + (static_cast<long long>(a) * ldg + b) * ges);
            }
    };
    clear();

    long long group = blockIdx.x;
    if (group >= groups) return;
    fetch(group, 0);
    int parity = 0;
    long long finished = -1;  // a group whose blocks are complete but not yet stored
    for (;;) {
        for (int s = 0; s < stages; ++s) {
            if constexpr (DIAG != 5) {
                publish(parity);
                __syncthreads();
            }
            // next stage of the pipeline: this group's, or the first of the next group
            if (s + 1 < stages) fetch(group, s + 1);
            else if (group + gridDim.x < groups) fetch(group + gridDim.x, 0);
            if (s == 0 && finished >= 0) {  // the previous group's result leaves AFTER the loads above were issued
                store(finished);
                clear();
                finished = -1;
            }
            if (computes && (DIAG != 2 || rows < 0)) {
                const unsigned bufBytes = static_cast<unsigned>(parity * BUF * sizeof(double));
                // rows past the end of the Jacobian were staged as zeros: they contribute nothing
#pragma unroll 2
                for (int r = 0; r < STAGE; ++r) {
                    const unsigned rowBytes = bufBytes + static_cast<unsigned>(r * ROW * sizeof(double));
                    double ja[kTile], jb[kTile];
#pragma unroll
                    for (int a = 0; a < kTile; ++a) ja[a] = *reinterpret_cast<const double*>(ldsBytes + (aPtr[a] + rowBytes));
#pragma unroll
                    for (int b = 0; b < kTile; ++b) jb[b] = *reinterpret_cast<const double*>(ldsBytes + (bPtr[b] + rowBytes));
                    if constexpr (WEIGHTED) {
                        const double w = *reinterpret_cast<const double*>(ldsBytes + (bufBytes + static_cast<unsigned>((STAGE_ELEMS + PAD + r * kNodes + n) * sizeof(double))));
#pragma unroll
                        for (int a = 0; a < kTile; ++a) ja[a] *= w;
                    }
#pragma unroll
                    for (int a = 0; a < kTile; ++a)
#pragma unroll
                        for (int b = 0; b < kTile; ++b) acc[a][b] = fma(ja[a], jb[b], acc[a][b]);
                }
            }
            parity ^= 1;
        }
        finished = group;
        group += gridDim.x;
        if (group >= groups) break;
    }
    store(finished);
}

/// 1 KiB of zeros: what the LDS-DMA copies of the rows past the end of a Jacobian read (they cannot be predicated into zeros).
__device__ const double kGnTilesZeros[128] = {};

/// The same pipeline with the global -> LDS copies done by the LDS-DMA path (global_load_lds_dwordx4: 16 bytes per lane, 1 KiB
/// per wavefront instruction straight into LDS -- no staging registers, no ds_write pass, no wait-then-write in the instruction
/// stream of the arithmetic).  A stage is 8 Jacobian rows = 8 * COLS row-column indices of 16 nodes = COLS KiB = COLS
/// instructions, dealt round-robin to the wavefronts; the weights of the stage are one more instruction.  DEPTH buffers; the
/// copies of stage q + DEPTH - 1 are issued right after the barrier that opens stage q, and stage q is waited for with a
/// COUNTED s_waitcnt vmcnt (the copies of the younger stages stay in flight across the barrier -- __syncthreads would drain
/// them).  Result stores also count in vmcnt; loads complete in order among themselves, so "at most n outstanding" with n = the
/// copies of the younger stages still implies that the stage is complete (a pending store only makes the wait longer).
/// The arithmetic of a stage is straight-line code over its 8 rows with two operand sets (the LDS reads of row r + 1 are issued
/// under the multiply-adds of row r) and immediate row offsets; a last stage with fewer rows contracts zeros.
/// Needs 16-byte aligned operands: even `count`, `jes`, `des`, 16-byte aligned bases (the launcher checks).
template <int COLS, int DEPTH, bool WEIGHTED, int DIAG = 0>
__global__ __launch_bounds__(GnTilesShape<COLS>::block) void GnHessianTilesDmaKernel(const double* __restrict__ jac, long long jes, const double* __restrict__ d, long long des,
                                                                                    double* __restrict__ g, long long ges, long long gns, long long ldg, int rows,
                                                                                    long long count) {
    using Shape = GnTilesShape<COLS>;
    constexpr int BLOCK = Shape::block;
    constexpr int WAVES = BLOCK / 64;
    constexpr int STAGE = 8;
    constexpr int ROW = COLS * kNodes;                           // doubles of one Jacobian row of the node group in LDS
    constexpr int STAGE_ELEMS = STAGE * ROW;                     // doubles per stage = COLS KiB
    constexpr int PAD = (Shape::side * kTile - COLS) * kNodes;   // blocks of the last block column read past the row: keep it inside the allocation
    constexpr int BUF = STAGE_ELEMS + STAGE * kNodes + PAD;      // Jacobian rows, weights, slack
    constexpr int COPIES = COLS;                                 // 1 KiB copies per stage (Jacobian part)
    constexpr int PER_WAVE = (COPIES + WAVES - 1) / WAVES;
    static_assert(DEPTH == 2 || DEPTH == 3);
    static_assert(DEPTH * BUF * sizeof(double) <= 160 * 1024);
    extern __shared__ double lds[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = t & (kNodes - 1);
    const int slot = t >> 4;
    const bool computes = slot < Shape::slots;
    const long long groups = (count + kNodes - 1) / kNodes;

    int A = 0, B = 0;
    {
        int s = computes ? slot : 0;
        while (s >= Shape::side - A) {
            s -= Shape::side - A;
            ++A;
        }
        B = A + s;
    }
    unsigned aPtr[kTile], bPtr[kTile];  // one LDS address register per operand: no ds_read2_b64 pairing (half rate)
#pragma unroll
    for (int i = 0; i < kTile; ++i) {
        aPtr[i] = static_cast<unsigned>(((A * kTile + i) * kNodes + n) * sizeof(double));
        bPtr[i] = static_cast<unsigned>(((B * kTile + i) * kNodes + n) * sizeof(double));
        asm volatile("" : "+v"(aPtr[i]));
        asm volatile("" : "+v"(bPtr[i]));
    }
    const char* ldsBytes = reinterpret_cast<const char*>(lds);

    const long long total = static_cast<long long>(rows) * COLS;
    const int stages = (rows + STAGE - 1) / STAGE;
    // copy lane: 8 lanes per row-column index (16 nodes = 128 bytes), 8 indices per instruction
    const int pairNode = (lane & 7) * 2, subIndex = lane >> 3;
    auto copy = [&](long long group, int stage, int buf) {
        if constexpr (DIAG >= 3) return;
        long long node = group * kNodes + pairNode;
        if (node > count - 2) node = count - 2;  // lanes of a ragged last group read the last pair (never used)
        const long long base = static_cast<long long>(stage) * STAGE * COLS;
        double* dst = lds + buf * BUF;
        // address = uniform part (scalar registers) + ONE 32-bit lane offset shared by all copies of the group
        const unsigned laneBytes = static_cast<unsigned>((subIndex * jes + node) * sizeof(double));
        const char* jacBytes = reinterpret_cast<const char*>(jac);
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int c = wave + i * WAVES;  // uniform
            if (c < COPIES) {
                const long long first = base + c * 8;  // uniform: first row-column index of the copy
                auto* to = (__attribute__((address_space(3))) void*)(dst + c * 128);
                if (first + 8 <= total) {
                    __builtin_amdgcn_global_load_lds(jacBytes + first * jes * static_cast<long long>(sizeof(double)) + laneBytes, to, 16, 0, 0);
                } else {  // last stage of a Jacobian whose row count is not a multiple of 8: rows past the end are copied from zeros
                    const long long rc = first + subIndex;
                    const double* from = rc < total ? jac + (rc * jes + node) : kGnTilesZeros + lane * 2;
                    __builtin_amdgcn_global_load_lds(from, to, 16, 0, 0);
                }
            }
        }
        if constexpr (WEIGHTED) {
            if (wave == 0) {
                const long long r = static_cast<long long>(stage) * STAGE + subIndex;
                const double* from = r < rows ? d + (r * des + node) : kGnTilesZeros + lane * 2;
                __builtin_amdgcn_global_load_lds(from, (__attribute__((address_space(3))) void*)(dst + STAGE_ELEMS), 16, 0, 0);
            }
        }
    };
    // copies a wavefront issues per stage: what may stay outstanding per younger stage
    const int mine = __builtin_amdgcn_readfirstlane((COPIES - wave + WAVES - 1) / WAVES + ((WEIGHTED && wave == 0) ? 1 : 0));

    double acc[kTile][kTile];
    auto clear = [&] {
#pragma unroll
        for (int a = 0; a < kTile; ++a)
#pragma unroll
            for (int b = 0; b < kTile; ++b) acc[a][b] = 0.0;
    };
    auto store = [&](long long group) {
        const long long node = group * kNodes + n;
        if (!computes || node >= count) return;
        if ((DIAG == 1 || DIAG == 4 || DIAG == 5) && acc[0][0] != 12345.678) return;
        double* __restrict__ gn = g + (node * gns + (static_cast<long long>(A * kTile) * ldg + B * kTile) * ges);
#pragma unroll
        for (int a = 0; a < kTile; ++a)
#pragma unroll
            for (int b = 0; b < kTile; ++b) {
                const int ga = A * kTile + a, gb = B * kTile + b;
                if (ga < COLS && gb < COLS && ga <= gb) __builtin_nontemporal_store(acc[a][b], gn + (static_cast<long long>(a) * ldg + b) * ges);
            }
    };
    // operands of row r of the current stage: base registers (advanced once per stage) + IMMEDIATE row offsets
    unsigned pa[kTile], pb[kTile], pw = 0;
    double ja[2][kTile], jb[2][kTile], jw[2];  // two operand sets: the reads of row r + 1 are in flight under the arithmetic of row r
    auto fetchRow = [&](auto rIndex) {
        constexpr int r = decltype(rIndex)::value;
#pragma unroll
        for (int a = 0; a < kTile; ++a) ja[r & 1][a] = *reinterpret_cast<const double*>(ldsBytes + pa[a] + r * ROW * sizeof(double));
#pragma unroll
        for (int b = 0; b < kTile; ++b) jb[r & 1][b] = *reinterpret_cast<const double*>(ldsBytes + pb[b] + r * ROW * sizeof(double));
        if constexpr (WEIGHTED) jw[r & 1] = *reinterpret_cast<const double*>(ldsBytes + pw + r * kNodes * sizeof(double));
    };
    auto contractRow = [&](auto rIndex) {
        constexpr int r = decltype(rIndex)::value;
        if constexpr (WEIGHTED) {
#pragma unroll
            for (int a = 0; a < kTile; ++a) ja[r & 1][a] *= jw[r & 1];
        }
#pragma unroll
        for (int a = 0; a < kTile; ++a)
#pragma unroll
            for (int b = 0; b < kTile; ++b) acc[a][b] = fma(ja[r & 1][a], jb[r & 1][b], acc[a][b]);
    };
    clear();

    // the pipeline runs over (group, stage) pairs in order; `ahead` is the pair DEPTH - 1 steps in front of the one being contracted
    long long group = blockIdx.x;
    if (group >= groups) return;
    long long aheadGroup = group;
    int aheadStage = 0, aheadBuf = 0;
    auto issueAhead = [&] {
        if (aheadGroup < groups) copy(aheadGroup, aheadStage, aheadBuf);
        aheadBuf = aheadBuf + 1 == DEPTH ? 0 : aheadBuf + 1;
        if (++aheadStage == stages) {
            aheadStage = 0;
            aheadGroup += gridDim.x;
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) issueAhead();
    int buf = 0;
    long long finished = -1;
    for (;;) {
        for (int s = 0; s < stages; ++s) {
            // stage (group, s) has landed when at most the copies of the DEPTH - 2 younger stages are outstanding
            if constexpr (DIAG < 3) {
                if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (mine == PER_WAVE + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE + 1) : "memory");
                else if (mine == PER_WAVE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE - 1 > 0 ? PER_WAVE - 1 : 0) : "memory");
            }
            if constexpr (DIAG != 5) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
            issueAhead();  // into the buffer whose stage was contracted before this barrier
            if (s == 0 && finished >= 0) {  // the previous group's result leaves AFTER the copies above were issued
                store(finished);
                clear();
                finished = -1;
            }
            if (computes && (DIAG != 2 || rows < 0)) {
                const unsigned bufBytes = static_cast<unsigned>(buf * BUF * sizeof(double));
#pragma unroll
                for (int i = 0; i < kTile; ++i) {
                    pa[i] = aPtr[i] + bufBytes;
                    pb[i] = bPtr[i] + bufBytes;
                }
                pw = bufBytes + static_cast<unsigned>((STAGE_ELEMS + n) * sizeof(double));
                // straight line, no branches: rows past the end of the Jacobian were copied from a block of zeros
                fetchRow(std::integral_constant<int, 0>{});
                fetchRow(std::integral_constant<int, 1>{});
                contractRow(std::integral_constant<int, 0>{});
                fetchRow(std::integral_constant<int, 2>{});
                contractRow(std::integral_constant<int, 1>{});
                fetchRow(std::integral_constant<int, 3>{});
                contractRow(std::integral_constant<int, 2>{});
                fetchRow(std::integral_constant<int, 4>{});
                contractRow(std::integral_constant<int, 3>{});
                fetchRow(std::integral_constant<int, 5>{});
                contractRow(std::integral_constant<int, 4>{});
                fetchRow(std::integral_constant<int, 6>{});
                contractRow(std::integral_constant<int, 5>{});
                fetchRow(std::integral_constant<int, 7>{});
                contractRow(std::integral_constant<int, 6>{});
                contractRow(std::integral_constant<int, 7>{});
            }
            buf = buf + 1 == DEPTH ? 0 : buf + 1;
        }
        finished = group;
        group += gridDim.x;
        if (group >= groups) break;
    }
    store(finished);
}

}  // namespace ungar_amd::kernels

namespace {

using ungar_amd::kernels::GnHessianTilesDmaKernel;
using ungar_amd::kernels::GnHessianTilesKernel;
using ungar_amd::kernels::GnTilesShape;
using ungar_amd::kernels::kNodes;
using ungar_amd::kernels::kTile;

struct GnTilesCall {
    const double* jac;
    long long jes;
    const double* d;
    long long des;
    double* g;
    long long ges, gns, ldg;
    int rows;
    long long count;
    hipStream_t stream;
    unsigned workgroups;
};

/// LDS-DMA pipeline (DEPTH stage buffers of 8 rows).
template <int C, int DEPTH>
void LaunchDma(const GnTilesCall& c) {
    const size_t bytes = static_cast<size_t>(DEPTH) * (8 * C * kNodes + 8 * kNodes + (GnTilesShape<C>::side * kTile - C) * kNodes) * sizeof(double);
    auto launch = [&](auto kernel) {
        static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
        (void)once;
        hipLaunchKernelGGL(kernel, dim3(c.workgroups), dim3(GnTilesShape<C>::block), bytes, c.stream, c.jac, c.jes, c.d, c.des, c.g, c.ges, c.gns, c.ldg, c.rows, c.count);
    };
    if (c.d) launch(GnHessianTilesDmaKernel<C, DEPTH, true>);
    else launch(GnHessianTilesDmaKernel<C, DEPTH, false>);
}

/// Register-staged pipeline (STAGE rows per barrier): any alignment.
template <int C, int STAGE>
void LaunchStaged(const GnTilesCall& c) {
    if (c.d) hipLaunchKernelGGL((GnHessianTilesKernel<C, STAGE, true>), dim3(c.workgroups), dim3(GnTilesShape<C>::block), 0, c.stream, c.jac, c.jes, c.d, c.des, c.g, c.ges, c.gns, c.ldg, c.rows, c.count);
    else hipLaunchKernelGGL((GnHessianTilesKernel<C, STAGE, false>), dim3(c.workgroups), dim3(GnTilesShape<C>::block), 0, c.stream, c.jac, c.jes, c.d, c.des, c.g, c.ges, c.gns, c.ldg, c.rows, c.count);
}

}  // namespace

/// 1 if the (cols) shape has a compiled instance: the widths of the built-in models' Jacobians with at least three 7-wide blocks
/// (for 8 columns -- rc_car -- the lane-per-node kernel is the faster one: 0.73 against 0.90 ms per 3.3 M nodes).
extern "C" int ungar_amd_gn_hessian_tiles_supported(int cols) { return cols == 49 || cols == 37 || cols == 17; }

/// G(a, b) of node i at g[(a * ldg + b) * ges + i * gns]  (unit-fastest: ges >= count, gns = 1; node-major: ges = 1, gns = block stride).
extern "C" int ungar_amd_launch_gn_hessian_tiles(const double* jac, long long jes, const double* d, long long des, double* g, long long ges, long long gns,
                                                  long long ldg, int rows, int cols, long long count, void* stream) {
    if (count <= 0) return 0;
    static const int computeUnits = [] {
        int device = 0, cus = 256;
        if (hipGetDevice(&device) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        return cus;
    }();
    const long long groups = (count + kNodes - 1) / kNodes;
    // persistent workgroups: one per compute unit for the 6-7 wavefront blocks of the wide Jacobians (their registers fill a compute
    // unit), four of the 2-wavefront blocks of cols = 17
    const int side = (cols + kTile - 1) / kTile, wavesPerGroup = (side * (side + 1) / 2 * kNodes + 63) / 64;
    const long long resident = static_cast<long long>(computeUnits) * (wavesPerGroup >= 5 ? 1 : wavesPerGroup >= 3 ? 2 : 4);
    static const int depth = [] {  // tuning knob: 0 = register-staged pipeline only, 2 / 3 = LDS-DMA pipeline with that many stage buffers
        const char* e = UNGAR_MEASUREMENT_SWITCH("UNGAR_GN_TILES_DMA");
        return e ? atoi(e) : 2;  // measured (81 920 ANYmal nodes): 0.427 ms with 2 buffers, 0.437 with 3; register-staged 0.56
    }();
    const GnTilesCall call{jac, jes, d, des, g, ges, gns, ldg, rows, count, static_cast<hipStream_t>(stream), static_cast<unsigned>(groups < resident ? groups : resident)};
    // the LDS-DMA path copies 16-byte pairs of nodes: even count and strides, 16-byte aligned bases
    const bool aligned = count % 2 == 0 && jes % 2 == 0 && reinterpret_cast<unsigned long long>(jac) % 16 == 0 &&
                         (!d || (des % 2 == 0 && reinterpret_cast<unsigned long long>(d) % 16 == 0));
    const bool laneOffsetsFit32 = jes < (1LL << 26);  // (7 rows x jes + node) x 8 bytes is the 32-bit lane part of a copy's address
    const bool dma = depth >= 2 && aligned && laneOffsetsFit32;
    switch (cols) {
        case 49:
            if (dma && depth == 3) LaunchDma<49, 3>(call);
            else if (dma) LaunchDma<49, 2>(call);
            else LaunchStaged<49, 4>(call);
            break;
        case 37:
            if (dma) LaunchDma<37, 2>(call);
            else LaunchStaged<37, 4>(call);
            break;
        case 17:
            if (dma) LaunchDma<17, 2>(call);
            else LaunchStaged<17, 8>(call);
            break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}
