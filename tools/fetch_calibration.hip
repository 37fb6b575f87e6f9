// Calibration of rocprofv3's FETCH_SIZE on the access mix of the register-resident Riccati kernels (ocp_riccati_wave_kernel.hpp): the guide's correction
// (FETCH_SIZE reports half of the bytes) is established for 16-byte-per-lane streaming reads only.  Every kernel below reads a KNOWN number of bytes, once,
// from a buffer far larger than the 256 MiB memory-side cache:
//   b128        raw_buffer_load_b128, lanes contiguous (1 KiB per instruction)           -- the operand loads of a knot
//   b64         raw_buffer_load_b64, lanes contiguous (512 B per instruction)            -- gradients, right-hand sides
//   touch       global_load_lds_dword, ONE 4-byte word per 128-byte line and lane        -- TouchLine: the prefetch of the next knot's operands
//   touch_b128  the touch of a 64 KiB block, then (one block later) its b128 read        -- what the recursion does: every line is touched, then read
// usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- fetch_calibration     (and a second pass with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
#include <hip/hip_runtime.h>

#include <cstdio>

using v4i = __attribute__((__vector_size__(4 * sizeof(int)))) int;
using v2i = __attribute__((__vector_size__(2 * sizeof(int)))) int;

__device__ __forceinline__ void TouchLine(const void* lanePointer, unsigned ldsJunk) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lanePointer), "s"(ldsJunk) : "memory");
}

constexpr long long kBlock = 64 * 1024;  // bytes one wavefront handles per step

__global__ __launch_bounds__(64) void calib_b128(const double* in, long long bytes, double* sink) {
    const long long begin = static_cast<long long>(blockIdx.x) * kBlock;
    if (begin >= bytes) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(in) + begin / 8, 0, static_cast<int>(kBlock), 0x00020000);
    v4i acc{0, 0, 0, 0};
    for (int off = 0; off < kBlock; off += 1024) acc += __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(threadIdx.x) * 16, off, 0);
    if (acc[0] == 0x7fffffff) sink[0] = acc[1];
}

__global__ __launch_bounds__(64) void calib_b64(const double* in, long long bytes, double* sink) {
    const long long begin = static_cast<long long>(blockIdx.x) * kBlock;
    if (begin >= bytes) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(in) + begin / 8, 0, static_cast<int>(kBlock), 0x00020000);
    v2i acc{0, 0};
    for (int off = 0; off < kBlock; off += 512) acc += __builtin_amdgcn_raw_buffer_load_b64(r, static_cast<int>(threadIdx.x) * 8, off, 0);
    if (acc[0] == 0x7fffffff) sink[0] = acc[1];
}

__global__ __launch_bounds__(64) void calib_touch(const double* in, long long bytes, double* sink) {
    __shared__ double junk[64];
    const long long begin = static_cast<long long>(blockIdx.x) * kBlock;
    if (begin >= bytes) return;
    const unsigned ldsJunk = static_cast<unsigned>(reinterpret_cast<size_t>(junk));
    const char* base = reinterpret_cast<const char*>(in) + begin;
    for (int off = 0; off < kBlock; off += 64 * 128) TouchLine(base + off + threadIdx.x * 128, ldsJunk);  // 64 lines per instruction
    __builtin_amdgcn_s_waitcnt(0);
    if (junk[threadIdx.x] == 1.2345e300) sink[0] = 1.0;
}

__global__ __launch_bounds__(64) void calib_touch_b128(const double* in, long long bytes, double* sink) {
    __shared__ double junk[64];
    const long long begin = static_cast<long long>(blockIdx.x) * kBlock * 8;  // eight blocks per wavefront: touch block k + 1 while reading block k
    if (begin >= bytes) return;
    const unsigned ldsJunk = static_cast<unsigned>(reinterpret_cast<size_t>(junk));
    const char* base = reinterpret_cast<const char*>(in) + begin;
    v4i acc{0, 0, 0, 0};
    for (int off = 0; off < kBlock; off += 64 * 128) TouchLine(base + off + threadIdx.x * 128, ldsJunk);
    for (int blk = 0; blk < 8; ++blk) {
        if (blk + 1 < 8)
            for (int off = 0; off < kBlock; off += 64 * 128) TouchLine(base + (blk + 1) * kBlock + off + threadIdx.x * 128, ldsJunk);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base) + blk * kBlock, 0, static_cast<int>(kBlock), 0x00020000);
        for (int off = 0; off < kBlock; off += 1024) acc += __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(threadIdx.x) * 16, off, 0);
    }
    if (acc[0] == 0x7fffffff) sink[0] = acc[1];
}

// writes: 8-byte buffer stores, lanes contiguous (the steps and gains the recursion writes), write-back and non-temporal
template <int AUX>
__global__ __launch_bounds__(64) void calib_store_b64(double* out, long long bytes) {
    const long long begin = static_cast<long long>(blockIdx.x) * kBlock;
    if (begin >= bytes) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + begin / 8, 0, static_cast<int>(kBlock), 0x00020000);
    const v2i v{static_cast<int>(threadIdx.x), 1};
    for (int off = 0; off < kBlock; off += 512) __builtin_amdgcn_raw_buffer_store_b64(v, r, static_cast<int>(threadIdx.x) * 8, off, AUX);
}

int main() {
    const long long bytes = 2ll << 30;  // 2 GiB: eight times the memory-side cache
    double *in, *sink;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(in, 0, bytes);
    const unsigned blocks = static_cast<unsigned>(bytes / kBlock);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_b128, dim3(blocks), dim3(64), 0, 0, in, bytes, sink);
        hipLaunchKernelGGL(calib_b64, dim3(blocks), dim3(64), 0, 0, in, bytes, sink);
        hipLaunchKernelGGL(calib_touch, dim3(blocks), dim3(64), 0, 0, in, bytes, sink);
        hipLaunchKernelGGL(calib_touch_b128, dim3(blocks / 8), dim3(64), 0, 0, in, bytes, sink);
        hipLaunchKernelGGL((calib_store_b64<0>), dim3(blocks), dim3(64), 0, 0, in, bytes);
        hipLaunchKernelGGL((calib_store_b64<2>), dim3(blocks), dim3(64), 0, 0, in, bytes);
    }
    (void)hipDeviceSynchronize();
    std::printf("bytes per launch: %lld (touch: %lld lines of 128 B)\n", bytes, bytes / 128);
    return 0;
}
