// ungar_amd :: layout conversion between the two operand layouts of the C ABI, at streaming rate.
//
// The node kernels run fastest on UNIT-FASTEST operands (element e of consecutive nodes contiguous); a reference-side caller holds
// INSTANCE-MAJOR data (one VariableMap buffer per problem instance: the elements of a node contiguous, variable_map.hpp / function.hpp:186-257
// hand `Eigen::Map`s of exactly that).  Writing a wide dense Jacobian straight into instance-major storage makes every store
// instruction touch 64 different cache lines (ANYmal: 1.76 ms per 81 920 nodes against 0.26-0.31 ms unit-fastest), so the recipe
// for such callers is: transpose the (small) inputs in, run the unit-fastest kernel, transpose the Jacobian out -- this kernel.
//
//   dst[n * dns + e * des] = src[n * sns + e * ses]      n < count, e < elements
// A workgroup moves a 64 x 64 tile through LDS (row pitch 65 doubles: conflict-free both ways): reads are coalesced along whichever
// of (n, e) is unit-stride in src, writes along whichever is unit-stride in dst; strides are arbitrary otherwise (padded element
// strides, leading dimensions).
#include <hip/hip_runtime.h>

namespace ungar_amd::kernels {

/// SRC_NODE_FAST: the node index is the unit-stride axis of src (unit-fastest -> instance-major); otherwise the element index is.
template <bool SRC_NODE_FAST>
__global__ __launch_bounds__(256) void TransposeNodesKernel(const double* __restrict__ src, long long sns, long long ses, double* __restrict__ dst, long long dns, long long des,
                                                            long long count, int elements, int elementTiles) {
    __shared__ double tile[64][65];
    const long long t = blockIdx.x;
    const long long n0 = (t / elementTiles) * 64;
    const int e0 = static_cast<int>(t % elementTiles) * 64;
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;  // 4 rows of 64 lanes per pass
    // load: lanes run along the unit-stride axis of src
#pragma unroll 4
    for (int r = row; r < 64; r += 4) {
        const long long n = SRC_NODE_FAST ? n0 + lane : n0 + r;
        const int e = SRC_NODE_FAST ? e0 + r : e0 + lane;
        if (n < count && e < elements) tile[SRC_NODE_FAST ? r : lane][SRC_NODE_FAST ? lane : r] = __builtin_nontemporal_load(src + n * sns + e * ses);  // tile[e][n]
    }
    __syncthreads();
    // store: lanes run along the other axis (the unit-stride axis of dst)
#pragma unroll 4
    for (int r = row; r < 64; r += 4) {
        const long long n = SRC_NODE_FAST ? n0 + r : n0 + lane;
        const int e = SRC_NODE_FAST ? e0 + lane : e0 + r;
        if (n < count && e < elements) __builtin_nontemporal_store(tile[SRC_NODE_FAST ? lane : r][SRC_NODE_FAST ? r : lane], dst + n * dns + e * des);
    }
}

}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_transpose_nodes(const double* src, long long sns, long long ses, double* dst, long long dns, long long des, long long count, int elements,
                                                 void* stream) {
    using namespace ungar_amd::kernels;
    if (count <= 0 || elements <= 0) return 0;
    const int elementTiles = (elements + 63) / 64;
    const long long tiles = ((count + 63) / 64) * elementTiles;
    const dim3 grid(static_cast<unsigned>(tiles)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // lanes follow the unit-stride axis of the source on the way in and of the destination on the way out; when both have the same
    // unit-stride axis (a strided copy) either instance is correct and one of the two passes is simply not coalesced
    if (sns == 1) hipLaunchKernelGGL(TransposeNodesKernel<true>, grid, block, 0, s, src, sns, ses, dst, dns, des, count, elements, elementTiles);
    else hipLaunchKernelGGL(TransposeNodesKernel<false>, grid, block, 0, s, src, sns, ses, dst, dns, des, count, elements, elementTiles);
    return static_cast<int>(hipGetLastError());
}
