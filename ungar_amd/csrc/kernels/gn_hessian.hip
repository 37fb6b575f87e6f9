// ungar_amd :: Gauss-Newton contraction  G = J^T diag(d) J  on the FP64 matrix cores (gfx950).
//
// Reference analogue: the Eigen sparse triple product
//   include/ungar/optimization/soft_sqp.hpp:257-264   J_i^T * H_barrier(h) * J_i,  H diagonal
// evaluated single-threaded on one instance.  Here: one wavefront per shooting node, the node's
// dense rows x cols Jacobian block (node-major, row-major) is streamed from HBM exactly once in
// 128-byte row segments straight into MFMA operand registers, and the cols x cols result is
// accumulated in registers with v_mfma_f64_16x16x4_f64:
//     A-operand (16x4)  = (J^T D)[16 ta + (l & 15)][4 ks + (l >> 4)]
//     B-operand (4x16)  =  J     [4 ks + (l >> 4)][16 tb + (l & 15)]
// so both operands of every tile pair come from the SAME four loads per k-step (one per column
// tile); no LDS is needed.  UPPER = false computes all T x T tiles and writes the full symmetric block
// with coalesced 128-byte row segments; UPPER = true computes only the T (T + 1) / 2 tiles on and above
// the diagonal and writes only the entries with row <= col (what the QP's objective matrix uses:
// Hessians are upper-triangular throughout the reference, function.hpp:236-274) -- 37 % fewer MFMAs,
// half the bytes written, fewer accumulators and hence one more wavefront per SIMD.
#include <hip/hip_runtime.h>

namespace ungar_amd::kernels {

using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;

template <int T, bool UPPER>  // T = number of 16-wide column tiles, cols <= 16 T
__global__ __launch_bounds__(256) void GnHessianKernel(const double* __restrict__ jac, long long js, long long ldj,
                                                       const double* __restrict__ d, long long ds, double* __restrict__ g,
                                                       long long gs, long long ldg, int rows, int cols, long long count) {
    const int lane = threadIdx.x & 63;
    const long long node = static_cast<long long>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (node >= count) return;  // whole wavefront exits together
    const double* __restrict__ J = jac + node * js;
    const double* __restrict__ D = d ? d + node * ds : nullptr;
    const int lc = lane & 15, lk = lane >> 4;

    f64x4 acc[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

    // The node's block is consumed in chunks of KC k-steps (KC * 4 rows): all loads of a chunk are issued
    // before its first MFMA, so that up to KC * T independent 128-byte row segments are in flight per
    // lane group instead of T (one wavefront per node has nothing else to hide HBM latency with).
    constexpr int KC = 5;   // rows beyond `rows` are loaded as zeros: padded k-steps are harmless
    const int ksteps = (rows + 3) >> 2;
    for (int k0 = 0; k0 < ksteps; k0 += KC) {
        double jv[KC][T], wv[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int r = 4 * (k0 + kk) + lk;
            const bool rowOk = r < rows;
            wv[kk] = rowOk ? (D ? D[r] : 1.0) : 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int c = 16 * t + lc;
                jv[kk][t] = (rowOk && c < cols) ? J[static_cast<long long>(r) * ldj + c] : 0.0;
            }
        }
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
#pragma unroll
            for (int a = 0; a < T; ++a) {
                const double av = jv[kk][a] * wv[kk];
#pragma unroll
                for (int b = UPPER ? a : 0; b < T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, jv[kk][b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // D-fragment map of v_mfma_f64_16x16x4_f64 (group size 1, 4 groups per block -- unlike the f32
    // 16x16 family): element `reg` of lane l is C[4 reg + (l >> 4)][l & 15].
    double* __restrict__ G = g + node * gs;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = UPPER ? a : 0; b < T; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * a + 4 * reg + lk, col = 16 * b + lc;
                if (row < cols && col < cols && (!UPPER || row <= col)) G[static_cast<long long>(row) * ldg + col] = acc[a][b][reg];
            }
}

}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_gn_hessian(const double* jac, long long js, long long ldj, const double* d, long long ds, double* g,
                                            long long gs, long long ldg, int rows, int cols, long long count, int upperOnly, void* stream) {
    using namespace ungar_amd::kernels;
    const int wavesPerBlock = 4;
    const dim3 grid(static_cast<unsigned>((count + wavesPerBlock - 1) / wavesPerBlock)), block(64 * wavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch ((cols + 15) / 16) {
        case 1:
            if (upperOnly) hipLaunchKernelGGL((GnHessianKernel<1, true>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            else hipLaunchKernelGGL((GnHessianKernel<1, false>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            break;
        case 2:
            if (upperOnly) hipLaunchKernelGGL((GnHessianKernel<2, true>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            else hipLaunchKernelGGL((GnHessianKernel<2, false>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            break;
        case 3:
            if (upperOnly) hipLaunchKernelGGL((GnHessianKernel<3, true>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            else hipLaunchKernelGGL((GnHessianKernel<3, false>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            break;
        case 4:
            if (upperOnly) hipLaunchKernelGGL((GnHessianKernel<4, true>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            else hipLaunchKernelGGL((GnHessianKernel<4, false>), grid, block, 0, s, jac, js, ldj, d, ds, g, gs, ldg, rows, cols, count);
            break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}
