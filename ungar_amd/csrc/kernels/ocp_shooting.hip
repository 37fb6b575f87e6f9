// ungar_amd :: batched SQP kernels for shooting problems with carried quantities and stage equality rows (ocp_shooting.hpp):
// QP data from the stage functions' sparse outputs, merit terms, stacked trial rows, and the per-instance line search + iteration
// bookkeeping of SoftSQPOptimizer::Optimize (reference include/ungar/optimization/soft_sqp.hpp:62-112, 143-158, 247-264;
// backtracking_line_search.hpp:116-151).  The QP itself is solved by the Riccati kernel (ocp_riccati.hip) on what is assembled here.
#include <hip/hip_runtime.h>

#include "ocp_barrier.hpp"
#include "ocp_shooting.hpp"

namespace ungar_amd::kernels {
namespace {

constexpr int kBlock = 64;

__device__ __forceinline__ const double* RowOf(const double* rows, const ShootingDims& d, long long b, int k) {
    return rows + (b * (d.N + 1) + k) * static_cast<long long>(d.nv());
}

/// One workgroup per node (instance, knot <= N).  Sparse values are scattered into dense LDS images first (a dense block is written
/// to global memory exactly once, coalesced; zero-filling and scattering in global memory would race), then the barrier terms are added.
__global__ __launch_bounds__(256) void ShootingAssembleKernel(const ShootingAssembleArgs a) {
    extern __shared__ double lds[];
    const ShootingDims& d = a.d;
    const long long node = blockIdx.x;
    const long long b = node / (d.N + 1);
    const int k = static_cast<int>(node - b * (d.N + 1));
    if (b >= d.batch) return;
    const int lane = static_cast<int>(threadIdx.x), lanes = static_cast<int>(blockDim.x);
    const int nz = d.nz(), nd = d.nd(), nc = d.nc, nx = d.nx;
    const bool stage = k < d.N;  // knot N: terminal cost only
    double* Wd = lds;                 // nd x nd
    double* gd = Wd + nd * nd;        // nd
    double* Jh = gd + nd;             // nh x nd
    double* d1 = Jh + a.nh * nd;      // nh
    double* d2 = d1 + a.nh;           // nh
    double* ABd = d2 + a.nh;          // nz x nd
    double* Ed = ABd + nz * nd;       // ne x nd
    const int total = nd * nd + nd + a.nh * nd + 2 * a.nh + nz * nd + a.ne * nd;
    for (int i = lane; i < total; i += lanes) lds[i] = 0.0;
    __syncthreads();
    const long long nodeOff = b * (d.N + 1) + k;
    for (int e = lane; e < a.pH.nnz; e += lanes) Wd[a.pH.rows[e] * nd + a.pH.cols[e]] = a.lH[nodeOff * a.pH.nnz + e];
    for (int e = lane; e < a.pg.nnz; e += lanes) gd[a.pg.cols[e]] = a.lg[nodeOff * a.pg.nnz + e];
    if (stage) {
        for (int e = lane; e < a.ph.nnz; e += lanes) Jh[a.ph.rows[e] * nd + a.ph.cols[e]] = a.hJ[nodeOff * a.ph.nnz + e];
        for (int j = lane; j < a.nh; j += lanes) {
            const double z = -a.h[nodeOff * a.nh + j];
            d1[j] = BarrierD1(a.barrier, z);
            d2[j] = BarrierD2(a.barrier, z);
        }
        // [A|B]: function column j over [x|u] is column nc + j of [c|x|u]
        for (int e = lane; e < a.pf.nnz; e += lanes) ABd[(nc + a.pf.rows[e]) * nd + nc + a.pf.cols[e]] = a.fJ[nodeOff * a.pf.nnz + e];
        if (d.carryInputs) {
            for (int r = lane; r < nc; r += lanes) ABd[r * nd + nz + r] = 1.0;
        } else {
            for (int e = lane; e < a.pc.nnz; e += lanes) ABd[a.pc.rows[e] * nd + nc + a.pc.cols[e]] = a.cJ[nodeOff * a.pc.nnz + e];
        }
        for (int e = lane; e < a.pe.nnz; e += lanes) Ed[a.pe.rows[e] * nd + a.pe.cols[e]] = a.eJ[nodeOff * a.pe.nnz + e];
    }
    __syncthreads();
    // W (upper triangle), w
    double* W = a.W + nodeOff * nd * nd;
    for (int idx = lane; idx < nd * nd; idx += lanes) {
        const int r = idx / nd, c = idx - r * nd;
        if (r > c) continue;
        double acc = Wd[idx];
        if (stage)
            for (int j = 0; j < a.nh; ++j) acc += d2[j] * Jh[j * nd + r] * Jh[j * nd + c];
        if (r == c && r >= nc && (stage || r < nz)) acc += a.regularization;  // the reference's 1e-6 I over its decision variables (soft_sqp.hpp:149-151)
        W[idx] = acc;
    }
    double* w = a.w + nodeOff * nd;
    for (int c = lane; c < nd; c += lanes) {
        double acc = gd[c];
        if (stage)
            for (int j = 0; j < a.nh; ++j) acc -= d1[j] * Jh[j * nd + c];  // d/dz b(-h) = -b'(-h) dh/dz
        w[c] = acc;
    }
    if (stage) {
        const long long stageOff = b * d.N + k;
        double* AB = a.AB + stageOff * nz * nd;
        for (int idx = lane; idx < nz * nd; idx += lanes) AB[idx] = ABd[idx];
        const double* next = RowOf(a.rows, d, b, k + 1);
        double* bo = a.b + stageOff * nz;
        for (int i = lane; i < nz; i += lanes) bo[i] = i < nc ? 0.0 : a.f[nodeOff * nx + (i - nc)] - next[i];
        if (a.ne > 0) {
            double* E = a.E + stageOff * a.ne * nd;
            for (int idx = lane; idx < a.ne * nd; idx += lanes) E[idx] = Ed[idx];
        }
        if (k == 0) {
            const double* row0 = RowOf(a.rows, d, b, 0);
            for (int i = lane; i < nz; i += lanes) a.dz0[b * nz + i] = i < nc ? 0.0 : a.xm[b * nx + (i - nc)] - row0[i];
        }
    }
}

__global__ __launch_bounds__(kBlock) void ShootingMeritKernel(const ShootingMeritArgs a) {
    const ShootingDims& d = a.d;
    const long long s = blockIdx.x;
    if (s >= d.batch) return;
    const long long b = a.period > 0 ? s % a.period : s;
    const int lane = static_cast<int>(threadIdx.x), nz = d.nz(), nd = d.nd(), nc = d.nc, nx = d.nx, N = d.N;
    double g2 = 0.0, obj = 0.0, bar = 0.0, slope = 0.0;
    const double* row0 = RowOf(a.rows, d, s, 0);
    for (int i = lane; i < nx; i += kBlock) {
        const double r = row0[nc + i] - a.xm[b * nx + i];
        g2 += r * r;
    }
    for (int idx = lane; idx < N * nx; idx += kBlock) {
        const int k = idx / nx, i = idx - k * nx;
        const double r = RowOf(a.rows, d, s, k + 1)[nc + i] - a.f[(s * (N + 1) + k) * nx + i];
        g2 += r * r;
    }
    if (a.e)
        for (int idx = lane; idx < N * a.ne; idx += kBlock) {
            const int k = idx / a.ne, j = idx - k * a.ne;
            const double r = a.e[(s * (N + 1) + k) * a.ne + j];
            g2 += r * r;
        }
    for (int k = lane; k <= N; k += kBlock) obj += a.l[s * (N + 1) + k];
    if (a.h)
        for (int idx = lane; idx < N * a.nh; idx += kBlock) {
            const int k = idx / a.nh, j = idx - k * a.nh;
            bar += Barrier(a.barrier, -a.h[(s * (N + 1) + k) * a.nh + j]);
        }
    const bool wantSlope = a.lg && a.dZ && a.slope;
    if (wantSlope)
        for (int idx = lane; idx < (N + 1) * a.pg.nnz; idx += kBlock) {
            const int k = idx / a.pg.nnz, e = idx - k * a.pg.nnz, c = a.pg.cols[e];
            double step = 0.0;
            if (c < nz) step = a.dZ[(b * (N + 1) + k) * nz + c];
            else if (k < N) step = a.dU[(b * N + k) * d.nu + (c - nz)];
            slope += a.lg[(s * (N + 1) + k) * a.pg.nnz + e] * step;
        }
    g2 = WaveSum(g2);
    obj = WaveSum(obj);
    bar = WaveSum(bar);
    slope = WaveSum(slope);
    if (lane == 0) {
        a.theta[s] = a.violationMultiplier * sqrt(g2);
        a.phi[s] = obj + bar;
        if (a.objective) a.objective[s] = obj;
        if (wantSlope) a.slope[s] = slope;
    }
}

__global__ __launch_bounds__(256) void ShootingTrialKernel(const ShootingTrialArgs a) {
    const ShootingDims& d = a.d;
    const long long node = blockIdx.x;  // stacked node: (candidate c, instance b, knot k)
    const long long s = node / (d.N + 1);
    const int k = static_cast<int>(node - s * (d.N + 1));
    if (s >= a.candidates * d.batch) return;
    const long long b = s % d.batch;
    const double alpha = a.alphas[s / d.batch];
    const int nz = d.nz(), nd = d.nd(), nv = d.nv(), nc = d.nc, N = d.N;
    const double* row = RowOf(a.rows, d, b, k);
    double* out = a.trial + node * nv;
    for (int j = static_cast<int>(threadIdx.x); j < nv; j += static_cast<int>(blockDim.x)) {
        double v = row[j];
        if (j < nc && d.carryInputs) {
            if (k > 0) {  // the trial input of the previous knot, the same bits the trial row k - 1 holds
                const double* prev = RowOf(a.rows, d, b, k - 1);
                v = fma(alpha, a.dU[(b * N + (k - 1)) * d.nu + j], prev[nz + j]);
            }
        } else if (j < nz) {
            v = fma(alpha, a.dZ[(b * (N + 1) + k) * nz + j], v);
        } else if (j < nd && k < N) {
            v = fma(alpha, a.dU[(b * N + k) * d.nu + (j - nz)], v);
        }
        out[j] = v;
    }
}

__global__ __launch_bounds__(kBlock) void ShootingSelectKernel(const ShootingSelectArgs a) {
    const ShootingDims& d = a.d;
    const long long b = blockIdx.x;
    if (b >= d.batch) return;
    if (a.active && a.active[b] == 0) {  // uniform over the workgroup
        if (threadIdx.x == 0) a.accepted[b] = 0.0;
        return;
    }
    const double theta = a.theta0[b], phi = a.phi0[b], slope = a.slope[b];
    int chosen = -1;
    const bool solved = !a.status || a.status[b] == 0;
    for (int c = 0; solved && c < a.candidates && chosen < 0; ++c)
        if (StepAcceptable(theta, phi, slope, a.thetaT[c * d.batch + b], a.phiT[c * d.batch + b], a.alphas[c], a.thetaMin, a.thetaMax, a.eta, a.gammaPhi, a.gammaTheta)) chosen = c;
    if (chosen < 0) {
        if (threadIdx.x == 0) {
            a.accepted[b] = 0.0;
            if (a.active) a.active[b] = 0;  // the reference's `break` on a rejected step (soft_sqp.hpp:88-90)
        }
        return;
    }
    const long long from = chosen * d.batch + b;
    const int nd = d.nd(), nv = d.nv();
    for (int idx = static_cast<int>(threadIdx.x); idx < (d.N + 1) * nd; idx += kBlock) {
        const int k = idx / nd, j = idx - k * nd;
        a.rows[(b * (d.N + 1) + k) * nv + j] = a.trial[(from * (d.N + 1) + k) * nv + j];
    }
    if (threadIdx.x == 0) {
        a.accepted[b] = a.alphas[chosen];
        const double difference = a.objectiveT[from] - a.objective0[b];
        if (a.active && difference < 0.0 && fabs(difference) < 1e-6) a.active[b] = 0;  // convergence criterion (soft_sqp.hpp:92-99)
    }
}

}  // namespace
}  // namespace ungar_amd::kernels

using namespace ungar_amd::kernels;

extern "C" int ungar_amd_launch_shooting_assemble(const ShootingAssembleArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    const std::size_t nd = static_cast<std::size_t>(a->d.nd()), nz = static_cast<std::size_t>(a->d.nz());
    const std::size_t lds = (nd * nd + nd + static_cast<std::size_t>(a->nh) * nd + 2 * static_cast<std::size_t>(a->nh) + nz * nd + static_cast<std::size_t>(a->ne) * nd) * sizeof(double);
    if (lds > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ShootingAssembleKernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(ShootingAssembleKernel, dim3(static_cast<unsigned>(a->d.batch * (a->d.N + 1))), dim3(nd >= 32 ? 256 : kBlock), lds, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_merit(const ShootingMeritArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    hipLaunchKernelGGL(ShootingMeritKernel, dim3(static_cast<unsigned>(a->d.batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_trial(const ShootingTrialArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    const long long nodes = static_cast<long long>(a->candidates) * a->d.batch * (a->d.N + 1);
    hipLaunchKernelGGL(ShootingTrialKernel, dim3(static_cast<unsigned>(nodes)), dim3(a->d.nv() > 64 ? 128 : kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_shooting_select(const ShootingSelectArgs* a, void* stream) {
    if (a->d.batch <= 0) return 0;
    hipLaunchKernelGGL(ShootingSelectKernel, dim3(static_cast<unsigned>(a->d.batch)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}
