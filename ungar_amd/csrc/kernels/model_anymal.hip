// ungar_amd :: kernels for the built-in 'anymal' (ANYmal B full-body) shooting-node model.
//   dense [A|B] block  -> lane-per-leg SPMD program (quad_kernel.hpp), 4 wavefronts / CU
//   sparse CSR values  -> the same program, CSR addressing (quad_anymal_sparse.hip)
//   value only         -> plain lane-per-node body
//   operands beyond 32-bit element offsets -> structured implicit differentiation, lane-per-node phased
//                         body with an LDS home (§4.3-4.4)
#include "../runtime/measurement.hpp"
#include "../gen/anymal_gen.hpp"
#include "../gen/anymal_quad_gen.hpp"
#include <cstdlib>

#include "quad_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_TRAITS(anymal)

namespace ungar_amd::kernels {
struct AnymalQuadBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { gen::anymal_quad::ValueJacobianQuad<double>(io); }
};
/// Value only (forward_zero; the SQP's stacked line search): the value sinks of the same lane-per-leg program, 2.2 k statements per
/// lane instead of the 7.3 k-statement lane-per-node body (1.4 KB of scratch per lane).
struct AnymalQuadValueBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { gen::anymal_quad::ValueQuad<double>(io); }
};
}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_anymal_quad_sparse(const ungar_amd::kernels::NodeLaunch* a, void* stream);  // quad_anymal_sparse.hip
extern "C" int ungar_amd_launch_anymal_quad_wide(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream);  // quad_anymal_wide.hip

extern "C" int ungar_amd_launch_anymal(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {
    using namespace ungar_amd::kernels;
    // the quad kernel addresses the block with wave-uniform element offsets (< 1813 * element stride): 32-bit ones up
    // to 2.37 M nodes per unit-fastest operand, 64-bit ones (quad_anymal_wide.hip, same program) beyond; only operands
    // with a negative element stride are left to the lane-per-node kernel
    const bool jacobian = mode == kModeDenseJacobian || mode == kModeSparseJacobian;
    if (jacobian && !QuadOffsetsFit32(*a) && a->jac.es >= 0) return ungar_amd_launch_anymal_quad_wide(mode, a, stream);
    static const bool laneValue = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ANYMAL_VALUE_LANE_PER_NODE") != nullptr;  // A/B switch (tools/bench_anymal_value.py)
    if (mode == kModeValue && !laneValue) {
        if (a->count <= 0 || !a->f.base) return 0;
        void* vsym = nullptr;
        const hipError_t ve = hipGetSymbolAddress(&vsym, HIP_SYMBOL(ungar_amd::gen::anymal_quad::kLegConstantsDev));
        if (ve != hipSuccess) return static_cast<int>(ve);
        const dim3 vgrid(static_cast<unsigned>((a->count + 15) / 16)), vblock(64);
        hipLaunchKernelGGL((QuadNodeKernel<64, 0, 0, false, false, AnymalQuadValueBody>), vgrid, vblock, 0, static_cast<hipStream_t>(stream), *a,
                           static_cast<const double(*)[4]>(vsym), AnymalQuadValueBody{});
        return static_cast<int>(hipGetLastError());
    }
    const bool quadOk = jacobian && QuadOffsetsFit32(*a);
    if (quadOk && mode == kModeSparseJacobian) return ungar_amd_launch_anymal_quad_sparse(a, stream);
    if (!quadOk) return static_cast<int>(LaunchNodeModel<Model_anymal, 64>(mode, *a, static_cast<hipStream_t>(stream)));
    if (a->count <= 0) return 0;
    constexpr int kBlock = 64;
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ungar_amd::gen::anymal_quad::kLegConstantsDev));
    if (e != hipSuccess) return static_cast<int>(e);
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((a->count + kBlock / 4 - 1) / (kBlock / 4))), block(kBlock);
    namespace Q = ungar_amd::gen::anymal_quad;
    static const bool noBuffer = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_NO_BUFFER_STORES") != nullptr;  // A/B switch for the store path (tools/)
    static const bool noPairs = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_NO_PAIRED_STORES") != nullptr;  // A/B switch: 8-byte stores where the paired 16-byte ones apply
    // Two entries of a column per store instruction where consecutive nodes lie at consecutive addresses and their number is even (partner nodes exchange one
    // value each with v_permlane16_swap: DESIGN.md section 4.13 (ii)); bit-identical to the 8-byte kernel below.
    if (UseStreamingStores(*a, mode, 37 * 49, 37) && QuadBufferStoresApply(*a) && QuadPairStoresApply(*a) && !noBuffer && !noPairs)
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, AnymalQuadBody, NoSparsePlan, unsigned, true, true>), grid, block, 0,
                           static_cast<hipStream_t>(stream), *a, ctab, AnymalQuadBody{});
    else if (UseStreamingStores(*a, mode, 37 * 49, 37) && QuadBufferStoresApply(*a) && !noBuffer)
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, AnymalQuadBody, NoSparsePlan, unsigned, true>), grid, block, 0,
                           static_cast<hipStream_t>(stream), *a, ctab, AnymalQuadBody{});
    else if (UseStreamingStores(*a, mode, 37 * 49, 37))
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, AnymalQuadBody>), grid, block, 0, static_cast<hipStream_t>(stream), *a, ctab,
                           AnymalQuadBody{});
    else
        hipLaunchKernelGGL((QuadNodeKernel<kBlock, Q::kLdsSlots, Q::kLdsUniformSlots, false, false, AnymalQuadBody>), grid, block, 0, static_cast<hipStream_t>(stream), *a, ctab,
                           AnymalQuadBody{});
    return static_cast<int>(hipGetLastError());
}
