// ungar_amd :: batched SQP kernels for optimal-control structure: the Riccati solve of the
// QP subproblem (ocp_riccati.hpp; replaces the OSQP call of soft_sqp.hpp:143-158 for this structure), the merit terms
// phi = objective + barrier and theta = c |g| of the line search (soft_sqp.hpp:68-87), trial points, and the three-way
// acceptance test of backtracking_line_search.hpp:116-151.  One 64-lane workgroup (= one wavefront) per MPC instance;
// instances are independent, so a launch of thousands of them fills the device.
#include <hip/hip_runtime.h>

#include "ocp_sqp.hpp"

namespace ungar_amd::kernels {

namespace {

constexpr int kBlock = 64;

/// Lanes of the workgroup stride over the index range and meet at a barrier.
struct DeviceExec {
    template <class F>
    __device__ __forceinline__ void ForEach(int n, F f) {
        for (int i = static_cast<int>(threadIdx.x); i < n; i += kBlock) f(i);
        __syncthreads();
    }
};

__global__ __launch_bounds__(kBlock) void RiccatiKernel(const RiccatiArgs a) {
    extern __shared__ double scratch[];
    const long long inst = blockIdx.x;
    if (inst >= a.batch) return;
    DeviceExec ex;
    RiccatiInstance(a, inst, scratch, ex);
}

/// Wavefront sum (64 lanes), result in every lane.
__device__ __forceinline__ double WaveSum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double Barrier(const BarrierParams& p, double z) {
    const double k = p.stiffness, eps = p.epsilon;
    if (p.type == 1) {  // relaxed log barrier (soft_inequality_constraint.hpp:98-105)
        if (z >= eps) return -k * log(z);
        const double t = (z - 2.0 * eps) / eps;
        return 0.5 * k * (t * t - 1.0) - k * log(eps);
    }
    // relaxed polynomial barrier (:131-190): quadratic below 0, cubic on [0, eps), 0 above
    const double a1 = k, b1 = -0.5 * k * eps;
    const double c1 = -(1.0 / 3.0) * (-b1 - a1 * eps) * eps - 0.5 * a1 * eps * eps - b1 * eps;
    if (z < 0.0) return 0.5 * a1 * z * z + b1 * z + c1;
    if (z < eps) {
        const double a2 = (-b1 - a1 * eps) / (eps * eps);
        return (1.0 / 3.0) * a2 * z * z * z + 0.5 * a1 * z * z + b1 * z + c1;
    }
    return 0.0;
}

__device__ __forceinline__ double BarrierD1(const BarrierParams& p, double z) {
    const double k = p.stiffness, eps = p.epsilon;
    if (p.type == 1) return z >= eps ? -k / z : k * (z - 2.0 * eps) / (eps * eps);
    const double a1 = k, b1 = -0.5 * k * eps;
    if (z < 0.0) return a1 * z + b1;
    if (z < eps) return (-b1 - a1 * eps) / (eps * eps) * z * z + a1 * z + b1;
    return 0.0;
}
__device__ __forceinline__ double BarrierD2(const BarrierParams& p, double z) {
    const double k = p.stiffness, eps = p.epsilon;
    if (p.type == 1) return z >= eps ? k / (z * z) : k / (eps * eps);
    const double a1 = k, b1 = -0.5 * k * eps;
    if (z < 0.0) return a1;
    if (z < eps) return 2.0 * (-b1 - a1 * eps) / (eps * eps) * z + a1;
    return 0.0;
}

/// One workgroup per instance, knots in sequence.
__global__ __launch_bounds__(kBlock) void StageQpKernel(const StageQpArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    const int lane = static_cast<int>(threadIdx.x), n = a.nx + a.nu;
    __shared__ double d1[64], d2[64];
    for (int i = lane; i < a.nx; i += kBlock) a.dx0.at(b, 0, i) = a.xm.at(b, 0, i) - a.X.at(b, 0, i);
    for (int k = 0; k < a.N; ++k) {
        for (int i = lane; i < a.nx; i += kBlock) a.b.at(b, k, i) = a.f.at(b, k, i) - a.X.at(b, k + 1, i);
        const bool ineq = a.h.base != nullptr;
        if (ineq)
            for (int j = lane; j < a.nh; j += kBlock) {
                const double z = -a.h.at(b, k, j);
                d1[j] = BarrierD1(a.barrier, z);
                d2[j] = BarrierD2(a.barrier, z);
            }
        __syncthreads();
        for (int idx = lane; idx < n * n; idx += kBlock) {
            const int r = idx / n, c = idx % n;
            double acc = 0.0;
            if (r <= c && ineq)
                for (int j = 0; j < a.nh; ++j) acc += d2[j] * a.hJac.at(b, k, j * n + r) * a.hJac.at(b, k, j * n + c);
            a.hess.at(b, k, idx) = acc;
        }
        for (int c = lane; c < n; c += kBlock) {
            double acc = a.costGrad.at(b, k, c);
            if (ineq)
                for (int j = 0; j < a.nh; ++j) acc -= d1[j] * a.hJac.at(b, k, j * n + c);  // d/dz b(-h) = -b'(-h) dh/dz
            a.grad.at(b, k, c) = acc;
        }
        __syncthreads();
        for (int e = lane; e < a.hesNnz; e += kBlock) a.hess.at(b, k, a.hesRow[e] * n + a.hesCol[e]) += a.costHes.at(b, k, e);  // distinct (row, col) per entry
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void MeritKernel(const MeritArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    const int lane = static_cast<int>(threadIdx.x);
    double g2 = 0.0, phi = 0.0, slope = 0.0;
    for (int i = lane; i < a.nx; i += kBlock) {
        const double r = a.X.at(b, 0, i) - a.xm.at(b, 0, i);
        g2 += r * r;
    }
    for (int idx = lane; idx < a.N * a.nx; idx += kBlock) {
        const int k = idx / a.nx, i = idx % a.nx;
        const double r = a.X.at(b, k + 1, i) - a.f.at(b, k, i);
        g2 += r * r;
    }
    if (a.cost.base)
        for (int k = lane; k < a.N; k += kBlock) phi += a.cost.at(b, k, 0);
    if (a.costN.base && lane == 0) phi += a.costN.at(b, 0, 0);
    if (a.h.base)
        for (int idx = lane; idx < a.N * a.nh; idx += kBlock) phi += Barrier(a.barrier, -a.h.at(b, idx / a.nh, idx % a.nh));
    const bool wantSlope = a.grad.base && a.dX.base && a.slope;
    if (wantSlope) {
        const int n = a.nx + a.nu;
        for (int idx = lane; idx < a.N * n; idx += kBlock) {
            const int k = idx / n, c = idx % n;
            slope += a.grad.at(b, k, c) * (c < a.nx ? a.dX.at(b, k, c) : a.dU.at(b, k, c - a.nx));
        }
        if (a.gradN.base)
            for (int i = lane; i < a.nx; i += kBlock) slope += a.gradN.at(b, 0, i) * a.dX.at(b, a.N, i);
    }
    g2 = WaveSum(g2);
    phi = WaveSum(phi);
    slope = WaveSum(slope);
    if (lane == 0) {
        a.theta[b] = a.violationMultiplier * sqrt(g2);
        a.phi[b] = phi;
        if (wantSlope) a.slope[b] = slope;
    }
}

__global__ __launch_bounds__(kBlock) void TrialKernel(const TrialArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    for (int idx = static_cast<int>(threadIdx.x); idx < (a.N + 1) * a.nx; idx += kBlock) {
        const int k = idx / a.nx, i = idx % a.nx;
        a.Xt.at(b, k, i) = a.X.at(b, k, i) + a.alpha * a.dX.at(b, k, i);
    }
    for (int idx = static_cast<int>(threadIdx.x); idx < a.N * a.nu; idx += kBlock) {
        const int k = idx / a.nu, i = idx % a.nu;
        a.Ut.at(b, k, i) = a.U.at(b, k, i) + a.alpha * a.dU.at(b, k, i);
    }
}

__global__ __launch_bounds__(kBlock) void AcceptKernel(const AcceptArgs a) {
    const long long b = blockIdx.x;
    if (b >= a.batch) return;
    if (a.accepted[b] != 0.0) return;  // uniform over the workgroup
    const double theta = a.theta0[b], phi = a.phi0[b], thetaNext = a.thetaT[b], phiNext = a.phiT[b], slope = a.slope[b];
    bool ok;
    if (thetaNext > a.thetaMax) ok = thetaNext < (1.0 - a.gammaTheta) * theta;
    else if (fmax(theta, thetaNext) < a.thetaMin && slope < 0.0) ok = phiNext < phi + a.eta * a.alpha * slope;
    else ok = phiNext < (1.0 - a.gammaPhi) * phi || thetaNext < (1.0 - a.gammaTheta) * theta;
    if (!ok) return;
    for (int idx = static_cast<int>(threadIdx.x); idx < (a.N + 1) * a.nx; idx += kBlock) a.X.at(b, idx / a.nx, idx % a.nx) = a.Xt.at(b, idx / a.nx, idx % a.nx);
    for (int idx = static_cast<int>(threadIdx.x); idx < a.N * a.nu; idx += kBlock) a.U.at(b, idx / a.nu, idx % a.nu) = a.Ut.at(b, idx / a.nu, idx % a.nu);
    __syncthreads();
    if (threadIdx.x == 0) a.accepted[b] = a.alpha;
}

}  // namespace
}  // namespace ungar_amd::kernels

using namespace ungar_amd::kernels;

extern "C" int ungar_amd_launch_riccati(const RiccatiArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    const std::size_t lds = static_cast<std::size_t>(RiccatiScratchDoubles(a->nx, a->nu)) * sizeof(double);
    if (lds > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    if (lds > 64 * 1024) {  // above the default dynamic-LDS limit the kernel has to opt in (160 KiB per CU on gfx950)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(RiccatiKernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(RiccatiKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), lds, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_stage_qp(const StageQpArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    if (a->nh > 64 || a->hesNnz > 160) return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(StageQpKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_merit(const MeritArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    hipLaunchKernelGGL(MeritKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_trial(const TrialArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    hipLaunchKernelGGL(TrialKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_ocp_accept(const AcceptArgs* a, void* stream) {
    if (a->batch <= 0) return 0;
    hipLaunchKernelGGL(AcceptKernel, dim3(static_cast<unsigned>(a->batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}
