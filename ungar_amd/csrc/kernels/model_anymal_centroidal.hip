// ungar_amd :: built-in rigid-body quantity node 'anymal_centroidal' (SURVEY.md section 8(f) N4): centroidal momentum h_G of ANYmal B (rbd/quantities/centroidal_momentum.hpp:42-43) and d h_G / d (q, v),
// whole batch per launch.
//   value + Jacobian (dense block or CSR values) -> lane-per-leg SPMD program (quad_rnea_kernel.hpp skeleton, csrc/codegen/quad_centroidal_program.hpp)
//   value only, or operands beyond 32-bit element offsets / with negative strides -> one lane per configuration (body lowered from the tape of csrc/models/rbd_nodes.hpp)
#include "../runtime/measurement.hpp"
#include "../gen/anymal_centroidal_gen.hpp"
#include "../gen/anymal_centroidal_quad_gen.hpp"
#include <cstdlib>

#include "quad_rnea_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_TRAITS(anymal_centroidal)

namespace ungar_amd::kernels {
struct AnymalCentroidalQuadBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { gen::anymal_centroidal_quad::ValueJacobianQuad<double>(io); }
};
}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_anymal_centroidal(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {
    using namespace ungar_amd::kernels;
    namespace Q = ungar_amd::gen::anymal_centroidal_quad;
    static const bool lanePerNode = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_CENTROIDAL_LANE_PER_NODE") != nullptr;  // A/B switch (tools/bench_rbd_nodes.py)
    const bool jacobian = mode == kModeDenseJacobian || mode == kModeSparseJacobian;
    const long long entries = mode == kModeDenseJacobian ? 6 * 37 : Q::kJacNnz;
    if (!jacobian || lanePerNode || a->jac.es < 0 || a->jac.es * entries >= (1LL << 32))
        return static_cast<int>(LaunchNodeModel<Model_anymal_centroidal, 64>(mode, *a, static_cast<hipStream_t>(stream)));
    if (a->count <= 0) return 0;
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ungar_amd::gen::anymal_centroidal_quad::kLegConstantsDev));
    if (e != hipSuccess) return static_cast<int>(e);
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((a->count + 15) / 16)), block(64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool streaming = UseStreamingStores(*a, mode, entries, 6);
    const AnymalCentroidalQuadBody body{};
    if (mode == kModeDenseJacobian) {
        if (streaming) hipLaunchKernelGGL((QuadRneaKernel<Q::kLdsSlots, Q::kLdsUniformSlots, false, true, AnymalCentroidalQuadBody, NoSparsePlan, 37>), grid, block, 0, s, *a, ctab, body);
        else hipLaunchKernelGGL((QuadRneaKernel<Q::kLdsSlots, Q::kLdsUniformSlots, false, false, AnymalCentroidalQuadBody, NoSparsePlan, 37>), grid, block, 0, s, *a, ctab, body);
    } else {
        if (streaming) hipLaunchKernelGGL((QuadRneaKernel<Q::kLdsSlots, Q::kLdsUniformSlots, true, true, AnymalCentroidalQuadBody, Q::SparsePlan, 37>), grid, block, 0, s, *a, ctab, body);
        else hipLaunchKernelGGL((QuadRneaKernel<Q::kLdsSlots, Q::kLdsUniformSlots, true, false, AnymalCentroidalQuadBody, Q::SparsePlan, 37>), grid, block, 0, s, *a, ctab, body);
    }
    return static_cast<int>(hipGetLastError());
}
