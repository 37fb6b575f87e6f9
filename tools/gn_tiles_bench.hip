// Standalone timing of the (node, block)-per-lane Gauss-Newton kernel and its diagnostic variants (no stores / no arithmetic /
// no global loads): where do the cycles go?  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gn_tiles_bench.hip -o /tmp/gn_tiles_bench
#include "../ungar_amd/csrc/kernels/gn_hessian_tiles.hip"

#include <cstdio>
#include <vector>

using namespace ungar_amd::kernels;

template <int STAGE, int DIAG>
static float Time(const double* J, const double* d, double* G, long long count, int reps, long long T = 0) {
    constexpr int C = 49;
    const dim3 grid(256), block(GnTilesShape<C>::block);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((GnHessianTilesKernel<C, STAGE, true, DIAG>), grid, block, 0, 0, J, T ? T : count, d, T ? T : count, G, T ? T : count, 1LL, (long long)C, 37, count, T, T * 37 * 49, T * 37, T * 49 * 49);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((GnHessianTilesKernel<C, STAGE, true, DIAG>), grid, block, 0, 0, J, T ? T : count, d, T ? T : count, G, T ? T : count, 1LL, (long long)C, 37, count, T, T * 37 * 49, T * 37, T * 49 * 49);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int DEPTH, int DIAG>
static float TimeDma(const double* J, const double* d, double* G, long long count, int reps, long long pad = 0) {
    constexpr int C = 49;
    const dim3 grid(256), block(GnTilesShape<C>::block);
    const size_t bytes = static_cast<size_t>(DEPTH) * (8 * C * kNodes + 8 * kNodes) * sizeof(double);
    auto kernel = GnHessianTilesDmaKernel<C, DEPTH, true, DIAG>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, grid, block, bytes, 0, J, count + pad, d, count + pad, G, count + pad, 1LL, (long long)C, 37, count);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, grid, block, bytes, 0, J, count + pad, d, count + pad, G, count + pad, 1LL, (long long)C, 37, count);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / reps;
}

int main() {
    const long long count = 81920;
    double *J, *d, *G;
    hipMalloc(&J, sizeof(double) * 37 * 49 * (count + 64));
    hipMalloc(&d, sizeof(double) * 37 * (count + 64));
    hipMalloc(&G, sizeof(double) * 49 * 49 * (count + 64));
    hipMemset(J, 0, sizeof(double) * 37 * 49 * count);
    hipMemset(d, 0, sizeof(double) * 37 * count);
    printf("LDS-DMA, 2 buffers: full %.4f  no-store %.4f  no-math %.4f  no-load %.4f  no-load-no-store %.4f  lds-math-only %.4f ms\n", TimeDma<2, 0>(J, d, G, count, 20),
           TimeDma<2, 1>(J, d, G, count, 20), TimeDma<2, 2>(J, d, G, count, 20), TimeDma<2, 3>(J, d, G, count, 20), TimeDma<2, 4>(J, d, G, count, 20), TimeDma<2, 5>(J, d, G, count, 20));
    printf("LDS-DMA, 3 buffers: full %.4f  no-store %.4f  no-math %.4f  no-load %.4f  no-load-no-store %.4f  lds-math-only %.4f ms\n", TimeDma<3, 0>(J, d, G, count, 20),
           TimeDma<3, 1>(J, d, G, count, 20), TimeDma<3, 2>(J, d, G, count, 20), TimeDma<3, 3>(J, d, G, count, 20), TimeDma<3, 4>(J, d, G, count, 20), TimeDma<3, 5>(J, d, G, count, 20));
    for (long long pad : {0LL, 48LL})
        printf("stride + %lld | 2 buffers: %.4f | 3 buffers: %.4f ms\n", pad, TimeDma<2, 0>(J, d, G, count, 20, pad), TimeDma<3, 0>(J, d, G, count, 20, pad));
    return 0;
}
