// Measured issue rate of v_fma_f64 on this box (context for the FP64-vector Gauss-Newton kernels): every lane runs 32 independent
// accumulation chains; 1, 2 and 4 wavefronts per SIMD.   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/valu_f64_peak tools/valu_f64_peak.hip
#include <hip/hip_runtime.h>

#include <cstdio>

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void Spin(double* out, int iters) {
    double acc[32];
    for (int k = 0; k < 32; ++k) acc[k] = k * 1e-3;
    double x = 1.0 + threadIdx.x * 1e-9;
    const double y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = __builtin_fma(acc[k], x, y);
        asm volatile("" : "+v"(x));
    }
    double s = 0;
    for (int k = 0; k < 32; ++k) s += acc[k];
    out[static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x] = s;
}

template <int BLOCK>
static void Run(int blocks, const char* what) {
    const int iters = 4000;
    double* out;
    if (hipMalloc(&out, static_cast<size_t>(blocks) * BLOCK * 8) != hipSuccess) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(Spin<BLOCK>, dim3(blocks), dim3(BLOCK), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(Spin<BLOCK>, dim3(blocks), dim3(BLOCK), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fmas = static_cast<double>(blocks) * BLOCK * iters * 32.0;
    std::printf("v_fma_f64, %s: %.1f TFLOP/s (%.3f ms)\n", what, fmas * 2 / (ms * 1e-3) / 1e12, ms);
    hipFree(out);
}

int main() {
    Run<256>(256, "1 wavefront / SIMD (256 x 256 lanes)");
    Run<512>(256, "2 wavefronts / SIMD (256 x 512 lanes)");
    Run<448>(256, "7 wavefronts / CU (256 x 448 lanes)");
    Run<1024>(256, "4 wavefronts / SIMD (256 x 1024 lanes)");
    Run<256>(256 * 8, "2048 x 256 lanes");
    return 0;
}
