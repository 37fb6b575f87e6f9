// ungar_amd :: Gauss-Newton contraction  G = J^T diag(d) J  (upper triangle) for Jacobians in the
// UNIT-FASTEST layout the node kernels write fastest (element e of node i at jac[e * jes + i]).
//
// gn_hessian.hip streams node-major blocks straight into MFMA operands; a unit-fastest Jacobian cannot be
// read that way (the operand of one node would be 64 scattered 8-byte loads), and transposing 1.2 GB
// between the two kernels costs more than either.  Here a workgroup owns 16 CONSECUTIVE nodes:
//   * all 512 lanes stage a chunk of KR rows of those 16 nodes through LDS -- one element of 16 nodes is
//     one aligned 128-byte line in HBM, so the loads are fully coalesced; the tile is kept as
//     [element][16 nodes + 1 pad] (the pad spreads the per-node operand reads over all banks);
//   * each of the 16 / NPW wavefronts accumulates NPW nodes with v_mfma_f64_16x16x4_f64, reading
//     its operands (A = (J^T D)[16 ta + l&15][4 k + l>>4], B = J[4 k + l>>4][16 tb + l&15]) from the tile;
//   * a three-deep software pipeline (two register stages, three LDS buffers) keeps chunks c+2 and c+3 in
//     flight from HBM while chunk c is multiplied;
//   * only the T (T + 1) / 2 tile products on/above the diagonal are formed and only entries with
//     row <= col are written (node-major, 128-byte row segments), as in ungar_gn_hessian_upper.
// Reference analogue: soft_sqp.hpp:257-264.
#include <hip/hip_runtime.h>

namespace ungar_amd::kernels {

using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;

constexpr int kGnNodes = 16;   // nodes per workgroup = doubles per 128-byte line
constexpr int kGnPad = 17;     // node dimension of the LDS tile
constexpr int kGnRows = 4;     // rows per chunk (one k-step): keeps the staging registers small

template <int T, int NPW, int LOADS>  // LOADS = ceil(4 cols / (threads / 16)) staged elements per lane and chunk; NPW = nodes per wavefront (1: 16 wavefronts, 4 per SIMD; 2: 8 wavefronts, 2 per SIMD)
__global__ __launch_bounds__(64 * kGnNodes / NPW) void GnHessianUpperSoaKernel(const double* __restrict__ jac, long long jes, const double* __restrict__ d,
                                                                      long long des, double* __restrict__ g, long long gs, long long ldg, int rows,
                                                                      int cols, long long count) {
    constexpr int kGnThreads = 64 * kGnNodes / NPW, kWaves = kGnNodes / NPW;
    extern __shared__ double lds[];
    const int tileDoubles = kGnRows * cols * kGnPad;      // one Jacobian chunk
    const int weightDoubles = kGnRows * kGnNodes;         // its row weights
    double* const tile[3] = {lds, lds + (tileDoubles + weightDoubles), lds + 2 * (tileDoubles + weightDoubles)};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, lk = lane >> 4;
    const long long nodeBase = static_cast<long long>(blockIdx.x) * kGnNodes;
    const int ln = tid & 15;                       // node this lane loads for
    const bool nodeOk = nodeBase + ln < count;
    const int chunkElems = kGnRows * cols, totalElems = rows * cols;
    constexpr int kMaxLoads = LOADS;
    const int chunks = (rows + kGnRows - 1) / kGnRows;

    f64x4 acc[NPW][T][T];
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) acc[n][a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

    // Two register stages + three LDS buffers: while chunk c is multiplied, chunks c+2 and c+3 are in flight
    // from HBM (one workgroup owns a CU -- the accumulators of 16 nodes fill most of its register file -- so
    // nothing but this software pipeline hides the memory latency).
    double stageA[kMaxLoads], stageB[kMaxLoads], wA = 0.0, wB = 0.0;
    auto fetch = [&](int c, double (&stage)[kMaxLoads], double& wstage) {  // global -> registers (16 lanes = one 128-byte line)
        if (c >= chunks) return;
#pragma unroll
        for (int i = 0; i < kMaxLoads; ++i) {
            const int e = (tid >> 4) + i * (kGnThreads / 16);  // element within the chunk
            const int ge = c * chunkElems + e;                 // element of the whole block: row < rows <=> ge < rows * cols
            stage[i] = (e < chunkElems && ge < totalElems && nodeOk) ? jac[static_cast<long long>(ge) * jes + nodeBase + ln] : 0.0;
        }
        if (tid < weightDoubles) {
            const int r = c * kGnRows + (tid >> 4);
            wstage = (r < rows && nodeOk) ? (d ? d[static_cast<long long>(r) * des + nodeBase + ln] : 1.0) : 0.0;
        }
    };
    auto park = [&](int c, const double (&stage)[kMaxLoads], double wstage) {  // registers -> LDS buffer c % 3
        if (c >= chunks) return;
        double* const t = tile[c % 3];
#pragma unroll
        for (int i = 0; i < kMaxLoads; ++i) {
            const int e = (tid >> 4) + i * (kGnThreads / 16);
            if (e < chunkElems) t[e * kGnPad + ln] = stage[i];
        }
        if (tid < weightDoubles) t[tileDoubles + tid] = wstage;
    };
    auto multiply = [&](int c) {
        const double* __restrict__ t = tile[c % 3];
#pragma unroll
        for (int ks = 0; ks < kGnRows / 4; ++ks) {
            const int r = 4 * ks + lk;  // row within the chunk
#pragma unroll
            for (int n = 0; n < NPW; ++n) {
                const int node = wave + kWaves * n;
                const double w = t[tileDoubles + r * kGnNodes + node];
                double jv[T];
#pragma unroll
                for (int a = 0; a < T; ++a) {
                    const int col = 16 * a + lc;
                    jv[a] = col < cols ? t[(r * cols + col) * kGnPad + node] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < T; ++a) {
                    const double av = jv[a] * w;
#pragma unroll
                    for (int b = a; b < T; ++b) acc[n][a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, jv[b], acc[n][a][b], 0, 0, 0);
                }
            }
        }
    };

    fetch(0, stageA, wA);
    fetch(1, stageB, wB);
    park(0, stageA, wA);
    fetch(2, stageA, wA);
    park(1, stageB, wB);
    fetch(3, stageB, wB);
    __syncthreads();
    for (int c = 0; c < chunks; c += 2) {
        // LDS holds chunks c and c+1; stage A carries c+2, stage B carries c+3
        multiply(c);
        park(c + 2, stageA, wA);  // buffer (c+2) % 3 last held chunk c-1: every wavefront is past it (barrier below)
        fetch(c + 4, stageA, wA);
        __syncthreads();
        if (c + 1 < chunks) multiply(c + 1);
        park(c + 3, stageB, wB);
        fetch(c + 5, stageB, wB);
        __syncthreads();
    }

    // D-fragment of v_mfma_f64_16x16x4_f64: element reg of lane l is C[4 reg + (l >> 4)][l & 15]
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const long long node = nodeBase + wave + kWaves * n;
        if (node >= count) continue;
        double* __restrict__ G = g + node * gs;
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = a; b < T; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * a + 4 * reg + lk, col = 16 * b + lc;
                    if (row < cols && col < cols && row <= col) G[static_cast<long long>(row) * ldg + col] = acc[n][a][b][reg];
                }
    }
}

}  // namespace ungar_amd::kernels

#ifndef UNGAR_GN_SOA_NODES_PER_WAVE
#define UNGAR_GN_SOA_NODES_PER_WAVE 2
#endif

extern "C" int ungar_amd_launch_gn_hessian_upper_soa(const double* jac, long long jes, const double* d, long long des, double* g, long long gs,
                                                      long long ldg, int rows, int cols, long long count, void* stream) {
    using namespace ungar_amd::kernels;
    constexpr int kNpw = UNGAR_GN_SOA_NODES_PER_WAVE;
    const dim3 grid(static_cast<unsigned>((count + kGnNodes - 1) / kGnNodes)), block(64 * kGnNodes / kNpw);
    const size_t ldsBytes = 3 * static_cast<size_t>(kGnRows * cols * kGnPad + kGnRows * kGnNodes) * sizeof(double);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int perPass = 64 * kGnNodes / kNpw / 16;  // elements covered by one load of all lanes
    const int loads = (kGnRows * cols + perPass - 1) / perPass;
#define UNGAR_GN_SOA_CASE(TT, LL)                                                                                                        \
    if ((cols + 15) / 16 == TT && loads <= LL) {                                                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&GnHessianUpperSoaKernel<TT, kNpw, LL>),                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes));                      \
        if (e != hipSuccess) return static_cast<int>(e);                                                                                 \
        hipLaunchKernelGGL((GnHessianUpperSoaKernel<TT, kNpw, LL>), grid, block, ldsBytes, s, jac, jes, d, des, g, gs, ldg, rows, cols, count); \
        return static_cast<int>(hipGetLastError());                                                                                      \
    }
    UNGAR_GN_SOA_CASE(1, 2)
    UNGAR_GN_SOA_CASE(2, 4)
    UNGAR_GN_SOA_CASE(3, 6)
    UNGAR_GN_SOA_CASE(4, 7)
    UNGAR_GN_SOA_CASE(4, 8)
#undef UNGAR_GN_SOA_CASE
    return static_cast<int>(hipErrorInvalidValue);
}
