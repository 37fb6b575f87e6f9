// ungar_amd :: 64-bit-offset instantiations of the lane-per-leg ANYmal program (quad_kernel.hpp).
// A unit-fastest Jacobian operand of more than 2.37 M nodes has element offsets (row * 49 + col) * stride beyond
// 2^32; these variants carry the wave-uniform offsets as 64-bit scalars (one more s_mul per store) so that such
// launches keep the headline kernel instead of dropping to the lane-per-node fallback (12 % of the HBM roofline).
// Outputs that large never fit the last-level cache, so only the streaming-store variants are instantiated.
#include "../gen/anymal_quad_gen.hpp"
#include "quad_kernel.hpp"

namespace ungar_amd::kernels {
struct AnymalQuadWideBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { gen::anymal_quad::ValueJacobianQuad<double>(io); }
};
}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_anymal_quad_wide(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {
    using namespace ungar_amd::kernels;
    if (a->count <= 0) return 0;
    constexpr int kBlock = 64;
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ungar_amd::gen::anymal_quad::kLegConstantsDev));
    if (e != hipSuccess) return static_cast<int>(e);
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((a->count + kBlock / 4 - 1) / (kBlock / 4))), block(kBlock);
    namespace Q = ungar_amd::gen::anymal_quad;
    using Wide = unsigned long long;
    if (mode == kModeSparseJacobian)
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, true, true, AnymalQuadWideBody, Q::SparsePlan, Wide>), grid, block, 0,
                           static_cast<hipStream_t>(stream), *a, ctab, AnymalQuadWideBody{});
    else
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, AnymalQuadWideBody, NoSparsePlan, Wide>), grid, block, 0,
                           static_cast<hipStream_t>(stream), *a, ctab, AnymalQuadWideBody{});
    return static_cast<int>(hipGetLastError());
}
