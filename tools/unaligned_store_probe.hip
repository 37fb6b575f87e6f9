#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void k(double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double* o = out + i * 5 + 1;  // 8-byte aligned, every other lane NOT 16-byte aligned
    *reinterpret_cast<double2*>(o) = make_double2(1.0 + i, 2.0 + i);
}
int main() {
    double* d; hipMalloc(&d, 8 * (64 * 5 + 8)); hipMemset(d, 0, 8 * (64 * 5 + 8));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    double h[64 * 5 + 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) bad += h[i * 5 + 1] != 1.0 + i || h[i * 5 + 2] != 2.0 + i || h[i * 5 + 3] != 0.0;
    printf("sync %s, %d lanes wrong\n", hipGetErrorString(e), bad);
    return bad;
}
