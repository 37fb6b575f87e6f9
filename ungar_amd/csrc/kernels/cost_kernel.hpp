// ungar_amd :: kernel skeleton of the scalar stage-cost node models: value, gradient and
// the upper triangle of the Hessian w.r.t. (x, u) per shooting node, one lane per node.
//   mode 0: value;  mode 1 / 2: value + gradient (the 1 x (nx+nu) gradient row, dense);
//   mode 3: value + gradient + Hessian values in the CSR order of ungar_model_hessian_sparsity.
// Reference analogue: Function::operator() / Jacobian / Hessian of a scalar objective (function.hpp:206-274).
#pragma once

#include "node_kernel.hpp"

namespace ungar_amd::kernels {

template <int NCOLS, bool STREAM>
struct CostIO : StridedIO<NCOLS, true, STREAM> {
    double* __restrict__ hb;
    long long he;
    int mode;
    using Base = StridedIO<NCOLS, true, STREAM>;
    __device__ __forceinline__ void j(int k, int r, int c, double v) const {
        if (mode >= kModeSparseJacobian && this->jb) Base::j(k, r, c, v);
    }
    __device__ __forceinline__ void h(int k, int /*row*/, int /*col*/, double v) const {
        if (mode == kModeHessian) StoreResult<STREAM>(hb + k * he, v);
    }
};

/// M: struct with kJacCols and  template <class IO> static void Body(IO&)  (the generated ValueGradientHessian).
template <class M, bool STREAM>
__global__ __launch_bounds__(128) void CostKernel(const NodeLaunch a, int mode) {
    const long long i = static_cast<long long>(blockIdx.x) * 128 + threadIdx.x;
    if (i >= a.count) return;
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    CostIO<M::kJacCols, STREAM> io{{a.x.base + b * a.x.bs + k * a.x.ks, a.u.base + b * a.u.bs + k * a.u.ks, nullptr, a.p.base + b * a.p.bs + k * a.p.ks,
                                    a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr, a.jac.base ? a.jac.base + b * a.jac.bs + k * a.jac.ks : nullptr,
                                    a.x.es, a.u.es, 0, a.p.es, a.f.es, a.jac.es},
                                   a.hes.base ? a.hes.base + b * a.hes.bs + k * a.hes.ks : nullptr,
                                   a.hes.es,
                                   mode};
    M::Body(io);  // the mode is wave-uniform: unused sinks are skipped by a scalar branch
}

template <class M>
inline int LaunchCostModel(int mode, const NodeLaunch& a, void* stream) {
    if (a.count <= 0) return 0;
    const dim3 grid(static_cast<unsigned>((a.count + 127) / 128)), block(128);
    // streaming stores only when the output of the mode is large and unit-fastest (node_kernel.hpp: StoreResult)
    const OperandView& out = mode == kModeHessian ? a.hes : mode == kModeValue ? a.f : a.jac;
    const bool streaming = out.es != 1 && a.count * 8 * (mode == kModeHessian ? M::kHesNnz : mode == kModeValue ? 1 : M::kJacNnz) > (256LL << 20);
    if (streaming) hipLaunchKernelGGL((CostKernel<M, true>), grid, block, 0, static_cast<hipStream_t>(stream), a, mode);
    else hipLaunchKernelGGL((CostKernel<M, false>), grid, block, 0, static_cast<hipStream_t>(stream), a, mode);
    return static_cast<int>(hipGetLastError());
}

}  // namespace ungar_amd::kernels

/// Binds a generated cost namespace (gen::<ns>) to the skeleton and defines the entry points c_api.cpp looks up.
#define UNGAR_AMD_DEFINE_COST_MODEL(ns)                                                                      \
    namespace ungar_amd::kernels {                                                                           \
    struct Cost_##ns {                                                                                       \
        static constexpr int kJacCols = gen::ns::kJacCols, kJacNnz = gen::ns::kJacNnz, kHesNnz = gen::ns::kHesNnz; \
        template <class IO>                                                                                  \
        __device__ __forceinline__ static void Body(IO& io) { gen::ns::ValueGradientHessian(io); }          \
    };                                                                                                       \
    }                                                                                                        \
    extern "C" int ungar_amd_launch_##ns(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {  \
        return ungar_amd::kernels::LaunchCostModel<ungar_amd::kernels::Cost_##ns>(mode, *a, stream);        \
    }                                                                                                        \
    extern "C" const int* ungar_amd_pattern_##ns(int which, int* nnz) {                                      \
        namespace G = ungar_amd::gen::ns;                                                                    \
        *nnz = which < 2 ? G::kJacNnz : G::kHesNnz;                                                          \
        return which == 0 ? G::kJacRow : which == 1 ? G::kJacCol : which == 2 ? G::kHesRow : G::kHesCol;     \
    }                                                                                                        \
    extern "C" void ungar_amd_dims_##ns(int* d) {                                                            \
        namespace G = ungar_amd::gen::ns;                                                                    \
        d[0] = G::kNx;                                                                                       \
        d[1] = G::kNu;                                                                                       \
        d[2] = G::kNw;                                                                                       \
        d[3] = G::kNp;                                                                                       \
        d[4] = 1;                                                                                            \
    }
