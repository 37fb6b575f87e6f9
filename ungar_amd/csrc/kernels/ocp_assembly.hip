// ungar_amd :: whole-horizon assembly of the equality-constraint Jacobian from per-node blocks (row N1
// of SURVEY.md §8(f)).
//
// The reference records ONE tape for  g(X, U) = [x_0 - x_m ; x_{k+1} - f(x_k, u_k)]_{k<N}
// (example/mpc/quadrotor.example.cpp:246-266) and gets a block-bidiagonal sparse Jacobian over the
// decision variables [X | U]:
//     rows 0..nx-1                 : d/dx_0 = I
//     rows nx + k nx + i           : d/dx_k = -A_k,  d/dx_{k+1} = I,  d/du_k = -B_k
// Here that matrix is produced for a whole batch of instances from the node kernels' dense
// [A_k | B_k] blocks: per instance the CSR value array (canonical order: A entries, the identity
// entry, B entries within a row) and the constraint values.  One lane per shooting node reads its
// block (coalesced in the unit-fastest layout) and writes its nx rows.
#include <hip/hip_runtime.h>

#include "ocp_assembly.hpp"

namespace ungar_amd::kernels {

__global__ __launch_bounds__(256) void OcpAssembleEqualityKernel(const OcpAssemblyArgs a) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= a.batch * a.N) return;
    const long long b = i / a.N;
    const int k = static_cast<int>(i - b * a.N);
    const int ncols = a.nx + a.nu;
    const long long perKnot = a.nnzNode + a.nx;
    double* __restrict__ val = a.values + b * a.vbs + a.nx + static_cast<long long>(k) * perKnot;
    double* __restrict__ g = a.g + b * a.gbs;
    const double* __restrict__ jac = a.jac + b * a.jbs + k * a.jks;
    const double* __restrict__ f = a.f + b * a.fbs + k * a.fks;
    const double* __restrict__ xn = a.X + b * a.xbs + static_cast<long long>(k + 1) * a.xks;
    if (k == 0) {  // initial-state rows: x_0 - x_m, Jacobian = I
        const double* __restrict__ x0 = a.X + b * a.xbs;
        for (int r = 0; r < a.nx; ++r) {
            g[r] = x0[r * a.xes] - a.xm[b * a.mbs + r * a.mes];
            a.values[b * a.vbs + r] = 1.0;
        }
    }
    long long out = 0;
    for (int r = 0; r < a.nx; ++r) {
        g[a.nx + k * a.nx + r] = xn[r * a.xes] - f[r * a.fes];
        int e = a.rowStart[r];
        const int end = a.rowStart[r + 1];
        for (; e < end && a.nodeCol[e] < a.nx; ++e) val[out++] = -jac[static_cast<long long>(r * ncols + a.nodeCol[e]) * a.jes];  // -A_k
        val[out++] = 1.0;                                                                                                         // d/dx_{k+1}
        for (; e < end; ++e) val[out++] = -jac[static_cast<long long>(r * ncols + a.nodeCol[e]) * a.jes];                          // -B_k
    }
}

}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_ocp_assemble(const ungar_amd::kernels::OcpAssemblyArgs* a, void* stream) {
    const long long nodes = a->batch * a->N;
    if (nodes <= 0) return 0;
    hipLaunchKernelGGL(ungar_amd::kernels::OcpAssembleEqualityKernel, dim3(static_cast<unsigned>((nodes + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *a);
    return static_cast<int>(hipGetLastError());
}
