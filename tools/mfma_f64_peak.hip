// Measured issue rate of v_mfma_f64_16x16x4_f64 on this box (context for the Gauss-Newton kernels' MFMA
// fractions): every wavefront runs independent accumulation chains, enough wavefronts to fill all SIMDs.
//   hipcc -O3 --offload-arch=gfx950 -o build/variants/mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>

#include <cstdio>

using f64x4 = __attribute__((__vector_size__(4 * sizeof(double)))) double;

__global__ __launch_bounds__(256) void Spin(double* out, int iters) {
    f64x4 acc[8];
    for (auto& a : acc) a = f64x4{0.0, 0.0, 0.0, 0.0};
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[k], 0, 0, 0);
    }
    double s = 0;
    for (auto& a : acc) s += a[0] + a[1] + a[2] + a[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 8, iters = 4000;
    double* out;
    if (hipMalloc(&out, blocks * 256 * 8) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(Spin, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(Spin, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = static_cast<double>(blocks) * 4 * iters * 8;  // 4 wavefronts per block
    std::printf("v_mfma_f64_16x16x4_f64: %.1f TFLOP/s (%.0f MFMA, %.3f ms)\n", mfmas * 2048 / (ms * 1e-3) / 1e12, mfmas, ms);
    return 0;
}
