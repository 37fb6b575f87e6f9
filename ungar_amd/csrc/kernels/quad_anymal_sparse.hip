// ungar_amd :: sparse-Jacobian instantiation of the lane-per-leg ANYmal program (quad_kernel.hpp):
// same generated body as the dense kernel of model_anymal.hip, every entry stored at its index in the
// CSR value array of ungar_model_jacobian_sparsity("anymal") instead of its (row, column) position.
// Its own translation unit so that the two ~12k-statement bodies compile in parallel.
#include "../gen/anymal_quad_gen.hpp"
#include "quad_kernel.hpp"

namespace ungar_amd::kernels {
struct AnymalQuadSparseBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { gen::anymal_quad::ValueJacobianQuad<double>(io); }
};
}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_anymal_quad_sparse(const ungar_amd::kernels::NodeLaunch* a, void* stream) {
    using namespace ungar_amd::kernels;
    if (a->count <= 0) return 0;
    constexpr int kBlock = 64;
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ungar_amd::gen::anymal_quad::kLegConstantsDev));
    if (e != hipSuccess) return static_cast<int>(e);
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((a->count + kBlock / 4 - 1) / (kBlock / 4))), block(kBlock);
    namespace Q = ungar_amd::gen::anymal_quad;
    if (UseStreamingStores(*a, kModeSparseJacobian, Q::kJacNnz, 37))
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, true, true, AnymalQuadSparseBody, Q::SparsePlan>), grid, block, 0, static_cast<hipStream_t>(stream), *a,
                           ctab, AnymalQuadSparseBody{});
    else
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, true, false, AnymalQuadSparseBody, Q::SparsePlan>), grid, block, 0, static_cast<hipStream_t>(stream), *a,
                           ctab, AnymalQuadSparseBody{});
    return static_cast<int>(hipGetLastError());
}
