// ungar_amd :: argument block of the whole-horizon assembly kernel, shared by its launcher (ocp_assembly.hip) and the
// C ABI (runtime/c_api.cpp).
#pragma once

namespace ungar_amd::kernels {

struct OcpAssemblyArgs {
    const double* X;        // states x_k of every instance: element e of (b, k) at X[b * xbs + k * xks + e * xes], k = 0..N
    long long xbs, xks, xes;
    const double* xm;       // measured state per instance (nx), instance stride mbs, element stride mes
    long long mbs, mes;
    const double* f;        // node values f_k:   element e of node (b, k) at f[b * fbs + k * fks + e * fes]
    long long fbs, fks, fes;
    const double* jac;      // node dense blocks: element d = r * ncols + c of node (b, k) at jac[b * jbs + k * jks + d * jes]
    long long jbs, jks, jes;
    double* g;              // out: constraint values, (N+1) nx per instance, instance stride gbs
    long long gbs;
    double* values;         // out: CSR values, nnz per instance, instance stride vbs
    long long vbs;
    const int* nodeRow;     // node pattern (device), nnzNode entries, row-major
    const int* nodeCol;
    const int* rowStart;    // node pattern CSR starts (nx + 1)
    int nx, nu, N, nnzNode;
    long long batch;
};

}  // namespace ungar_amd::kernels
