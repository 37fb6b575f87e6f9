// ungar_amd :: built-in rigid-body quantity node 'anymal_crba' (SURVEY.md section 8(f) N4): joint-space inertia matrix M(q) of ANYmal B, 18 x 18 row-major (rbd/quantities/joint_space_inertia_matrix.hpp:42-43) and d M / d q,
// whole batch per launch.
//   value + CSR Jacobian values -> lane-per-leg SPMD program (quad_crba_kernel.hpp, csrc/codegen/quad_crba_program.hpp)
//   value only, the dense 324 x 19 block (85 % structural zeros), operands beyond 32-bit element offsets -> one lane per configuration (body lowered from the tape of csrc/models/rbd_nodes.hpp)
#include "../runtime/measurement.hpp"
#include "../gen/anymal_crba_gen.hpp"
#include "../gen/anymal_crba_quad_gen.hpp"
#include <cstdlib>

#include "quad_crba_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_TRAITS(anymal_crba)

namespace ungar_amd::kernels {
struct AnymalCrbaQuadBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { gen::anymal_crba_quad::ValueJacobianQuad<double>(io); }
};
}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_anymal_crba(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {
    using namespace ungar_amd::kernels;
    namespace Q = ungar_amd::gen::anymal_crba_quad;
    static const bool lanePerNode = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_CRBA_LANE_PER_NODE") != nullptr;  // A/B switch (tools/bench_rbd_nodes.py)
    if (mode != kModeSparseJacobian || lanePerNode || a->jac.es < 0 || a->jac.es * static_cast<long long>(Q::kJacNnz) >= (1LL << 32))
        return static_cast<int>(LaunchNodeModel<Model_anymal_crba, 64>(mode, *a, static_cast<hipStream_t>(stream)));
    if (a->count <= 0) return 0;
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ungar_amd::gen::anymal_crba_quad::kLegConstantsDev));
    if (e != hipSuccess) return static_cast<int>(e);
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((a->count + 15) / 16)), block(64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const AnymalCrbaQuadBody body{};
    if (UseStreamingStores(*a, mode, Q::kJacNnz, 324)) hipLaunchKernelGGL((QuadCrbaKernel<Q::kLdsSlots, true, AnymalCrbaQuadBody, Q::SparsePlan>), grid, block, 0, s, *a, ctab, body);
    else hipLaunchKernelGGL((QuadCrbaKernel<Q::kLdsSlots, false, AnymalCrbaQuadBody, Q::SparsePlan>), grid, block, 0, s, *a, ctab, body);
    return static_cast<int>(hipGetLastError());
}
