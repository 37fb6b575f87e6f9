// ungar_amd :: Gauss-Newton contraction  G = J^T diag(d) J  (upper triangle) for UNIT-FASTEST Jacobians, one LANE per node.
//
// The FP64 matrix instruction has no rate advantage over the FP64 vector ALU on gfx950 (78.6 TFLOP/s data sheet for both,
// 47 TFLOP/s measured for v_mfma_f64_16x16x4_f64, tools/mfma_f64_peak.hip), and what it costs to feed it from the layout the
// node kernels write -- element e of 64 consecutive nodes contiguous -- is a transposition through LDS that left
// gn_hessian_soa.hip at 0.81 ms per 81 920 ANYmal blocks, 43 % of the matrix rate.  In that layout
// the contraction is a plain structure-of-arrays computation instead: a wavefront holds 64 consecutive nodes, one per lane,
// and accumulates a TILE x TILE block of G for all of them,
//     acc[a][b] += (d_r J[r][A + a]) * J[r][B + b]        r = 0 .. rows - 1,
// every operand a fully coalesced 512-byte load, no LDS, no cross-lane traffic, TILE^2 fused multiply-adds per 2 TILE loads.
// The (cols / TILE)(cols / TILE + 1) / 2 block pairs on or above the diagonal are spread over the wavefronts of consecutive
// workgroups, so the rows of a node group are shared through the L1 / L2 caches while they are hot.  The result is written
// with caller-chosen strides: unit-fastest (coalesced; what the batched Riccati solve and a second contraction read) or
// node-major blocks (the layout of ungar_gn_hessian_upper; lane-strided stores, 5x slower -- kept for completeness).
// Measured (MI355X, 37 x 49, 81 920 nodes): 0.90 ms with the XCD renumbering (0.95 without) against 0.78-0.82 ms of the
// LDS-staged matrix-core kernel: the 8x re-fetch of every Jacobian entry through L1 / L2 (9.5 GB per launch) is the limit,
// not the arithmetic (8.2 of 78.6 TFLOP/s).  A variant staging each row through LDS (one barrier per row, four passes of 8
// wavefronts) exposed the HBM latency of every row and took 2.5 ms; it was not kept.  This kernel is the one that writes a
// unit-fastest G, which is what a lane-per-instance consumer wants.  Reference analogue: soft_sqp.hpp:257-264.
#include "../runtime/measurement.hpp"
#include <hip/hip_runtime.h>

#include <cstdlib>

namespace ungar_amd::kernels {

template <int TILE, bool WEIGHTED, int UNROLL>
__global__ __launch_bounds__(256) void GnHessianLanesKernel(const double* __restrict__ jac, long long jes, const double* __restrict__ d, long long des,
                                                            double* __restrict__ g, long long ges, long long gns, long long ldg, int rows, int cols, long long count,
                                                            int blocksPerSide, int pairs) {
    // Consecutive workgroups are dealt round-robin to the 8 XCDs, each with its own L2: renumber them so that workgroups
    // with consecutive LOGICAL ids -- the block pairs of one node group, which read the same Jacobian rows -- share an XCD.
    const long long perXcd = (static_cast<long long>(gridDim.x) + 7) / 8;
    const long long logical = (static_cast<long long>(blockIdx.x) & 7) * perXcd + (static_cast<long long>(blockIdx.x) >> 3);
    // wavefront w of the launch: node group w / pairs (64 nodes), block pair w % pairs
    const long long wave = logical * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long group = wave / pairs;
    int pair = static_cast<int>(wave - group * pairs);
    const long long node = group * 64 + lane;
    if (group * 64 >= count) return;
    // unrank the pair (A <= B) of the upper block triangle, row by row
    int A = 0;
    while (pair >= blocksPerSide - A) {
        pair -= blocksPerSide - A;
        ++A;
    }
    const int B = A + pair;
    const int a0 = A * TILE, b0 = B * TILE;
    const bool live = node < count;
    const double* __restrict__ jn = jac + (live ? node : count - 1);  // out-of-range lanes read a valid node and discard
    const double* __restrict__ dn = WEIGHTED ? d + (live ? node : count - 1) : nullptr;

    double acc[TILE][TILE];
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) acc[a][b] = 0.0;

#pragma unroll UNROLL  // rows of operands in flight: the kernel's own ILP hides the load latency (1: 184 VGPRs, 2 wavefronts / SIMD; 4: 382, one)
    for (int r = 0; r < rows; ++r) {
        const double* __restrict__ row = jn + static_cast<long long>(r) * cols * jes;
        double ja[TILE], jb[TILE];
        const double w = WEIGHTED ? dn[static_cast<long long>(r) * des] : 1.0;
#pragma unroll
        for (int a = 0; a < TILE; ++a) ja[a] = a0 + a < cols ? row[static_cast<long long>(a0 + a) * jes] : 0.0;
#pragma unroll
        for (int b = 0; b < TILE; ++b) jb[b] = b0 + b < cols ? row[static_cast<long long>(b0 + b) * jes] : 0.0;
        if constexpr (WEIGHTED) {
#pragma unroll
            for (int a = 0; a < TILE; ++a) ja[a] *= w;
        }
#pragma unroll
        for (int a = 0; a < TILE; ++a)
#pragma unroll
            for (int b = 0; b < TILE; ++b) acc[a][b] = fma(ja[a], jb[b], acc[a][b]);
    }
    if (!live) return;
    double* __restrict__ gn = g + node * gns;
#pragma unroll
    for (int a = 0; a < TILE; ++a)
#pragma unroll
        for (int b = 0; b < TILE; ++b) {
            const int ga = a0 + a, gb = b0 + b;
            if (ga < cols && gb < cols && ga <= gb) __builtin_nontemporal_store(acc[a][b], gn + (static_cast<long long>(ga) * ldg + gb) * ges);
        }
}

}  // namespace ungar_amd::kernels

/// G(a, b) of node i at g[(a * ldg + b) * ges + i * gns]  (unit-fastest: ges = count, gns = 1; node-major: ges = 1, gns = block stride).
extern "C" int ungar_amd_launch_gn_hessian_lanes(const double* jac, long long jes, const double* d, long long des, double* g, long long ges, long long gns,
                                                  long long ldg, int rows, int cols, long long count, void* stream) {
    using namespace ungar_amd::kernels;
    constexpr int kTile = 7;
    const int side = (cols + kTile - 1) / kTile, pairs = side * (side + 1) / 2;
    const long long groups = (count + 63) / 64, waves = groups * pairs;
    const dim3 grid(static_cast<unsigned>((((waves + 3) / 4 + 7) / 8) * 8)), block(256);  // a multiple of 8: the XCD renumbering is a bijection
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int unroll = [] {  // tuning knob (tools/bench_gn_hessian.py sweeps it); default = the measured best
        const char* e = UNGAR_MEASUREMENT_SWITCH("UNGAR_GN_LANES_UNROLL");
        return e ? atoi(e) : 1;  // measured on MI355X (ANYmal block, 81 920 nodes): 0.90 / 1.13 / 1.24 ms for 1 / 2 / 4 -- occupancy beats unrolling
    }();
#define UNGAR_GN_LANES_LAUNCH(W, U) \
    hipLaunchKernelGGL((GnHessianLanesKernel<kTile, W, U>), grid, block, 0, s, jac, jes, d, des, g, ges, gns, ldg, rows, cols, count, side, pairs)
    if (d) {
        if (unroll >= 4) UNGAR_GN_LANES_LAUNCH(true, 4);
        else if (unroll >= 2) UNGAR_GN_LANES_LAUNCH(true, 2);
        else UNGAR_GN_LANES_LAUNCH(true, 1);
    } else {
        if (unroll >= 4) UNGAR_GN_LANES_LAUNCH(false, 4);
        else if (unroll >= 2) UNGAR_GN_LANES_LAUNCH(false, 2);
        else UNGAR_GN_LANES_LAUNCH(false, 1);
    }
#undef UNGAR_GN_LANES_LAUNCH
    return static_cast<int>(hipGetLastError());
}
