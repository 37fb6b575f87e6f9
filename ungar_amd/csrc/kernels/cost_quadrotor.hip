// ungar_amd :: built-in scalar stage-cost node 'quadrotor_cost' (SURVEY.md section 8(f) N2): value, gradient
// and the upper triangle of the Hessian w.r.t. (x, u) per shooting node, one lane per node.
//   mode 0: value;  mode 1 / 2: value + gradient (sparse = dense: the 1 x (nx+nu) gradient is full);
//   mode 3: value + gradient + Hessian values in the CSR order of ungar_model_hessian_sparsity.
// Reference analogue: Function::operator() / Jacobian / Hessian of a scalar objective (function.hpp:206-274).
#include "../gen/quadrotor_cost_gen.hpp"
#include "node_kernel.hpp"

namespace ungar_amd::kernels {

namespace G = ungar_amd::gen::quadrotor_cost;

template <bool STREAM>
struct CostIO : StridedIO<G::kJacCols, true, STREAM> {
    double* __restrict__ hb;
    long long he;
    int mode;
    using Base = StridedIO<G::kJacCols, true, STREAM>;
    __device__ __forceinline__ void j(int k, int r, int c, double v) const {
        if (mode >= kModeSparseJacobian && this->jb) Base::j(k, r, c, v);
    }
    __device__ __forceinline__ void h(int k, int /*row*/, int /*col*/, double v) const {
        if (mode == kModeHessian) StoreResult<STREAM>(hb + k * he, v);
    }
};

template <bool STREAM>
__global__ __launch_bounds__(128) void CostKernel(const NodeLaunch a, int mode) {
    const long long i = static_cast<long long>(blockIdx.x) * 128 + threadIdx.x;
    if (i >= a.count) return;
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    CostIO<STREAM> io{{a.x.base + b * a.x.bs + k * a.x.ks, a.u.base + b * a.u.bs + k * a.u.ks, nullptr, a.p.base + b * a.p.bs + k * a.p.ks,
                       a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr, a.jac.base ? a.jac.base + b * a.jac.bs + k * a.jac.ks : nullptr, a.x.es, a.u.es,
                       0, a.p.es, a.f.es, a.jac.es},
                      a.hes.base ? a.hes.base + b * a.hes.bs + k * a.hes.ks : nullptr,
                      a.hes.es,
                      mode};
    G::ValueGradientHessian(io);  // the mode is wave-uniform: unused sinks are skipped by a scalar branch
}

}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_quadrotor_cost(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {
    using namespace ungar_amd::kernels;
    if (a->count <= 0) return 0;
    const dim3 grid(static_cast<unsigned>((a->count + 127) / 128)), block(128);
    // streaming stores only for the large output of the mode (Hessian values)
    const OperandView& out = mode == kModeHessian ? a->hes : mode == kModeValue ? a->f : a->jac;
    const bool streaming = out.es != 1 && a->count * 8 * (mode == kModeHessian ? G::kHesNnz : mode == kModeValue ? 1 : G::kJacNnz) > (256LL << 20);
    if (streaming) hipLaunchKernelGGL(CostKernel<true>, grid, block, 0, static_cast<hipStream_t>(stream), *a, mode);
    else hipLaunchKernelGGL(CostKernel<false>, grid, block, 0, static_cast<hipStream_t>(stream), *a, mode);
    return static_cast<int>(hipGetLastError());
}
extern "C" const int* ungar_amd_pattern_quadrotor_cost(int which, int* nnz) {
    namespace G = ungar_amd::gen::quadrotor_cost;
    *nnz = which < 2 ? G::kJacNnz : G::kHesNnz;
    return which == 0 ? G::kJacRow : which == 1 ? G::kJacCol : which == 2 ? G::kHesRow : G::kHesCol;
}
extern "C" void ungar_amd_dims_quadrotor_cost(int* d) {
    namespace G = ungar_amd::gen::quadrotor_cost;
    d[0] = G::kNx;
    d[1] = G::kNu;
    d[2] = G::kNw;
    d[3] = G::kNp;
    d[4] = 1;
}
