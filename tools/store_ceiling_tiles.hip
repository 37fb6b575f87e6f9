// Store-only kernels for the TILED result layouts of the ANYmal node kernel (round 6; DESIGN.md "store path"): what the 37 x 49
// Jacobian block of 81 920 nodes costs to WRITE when a wavefront (16 nodes, lane = 16 * row + 4 * leg + j) owns one contiguous tile.
//   wave8   : [tile][sink][64 lanes]        one 8-byte store instruction = 512 contiguous bytes (the register image of the wavefront)
//   wave16  : [tile][sink pair][64 lanes x 2] one 16-byte store instruction = 1 KiB contiguous (partner nodes swapped beforehand)
//   col16   : [tile][column][row][16 nodes]  a column = 37 rows x 128 B staged somewhere and written as 4 x 1 KiB + 640 B (40 lanes)
//   quad8   : the product's present pattern in the unit-fastest layout (4 runs of 128 B per instruction), for the same-run comparison
// Every kernel holds 40 KiB of LDS so that ONE wavefront runs per SIMD as in the product kernel; SPACING dependent FMAs sit between stores.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int kEntries = 37 * 49;           // 1813
constexpr int kSinks = 454;                 // 1816 slots of 128 B: 3 pad
constexpr int kTileDoubles = kSinks * 64;   // 29 056 doubles = 227 KiB per 16 nodes

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define OCCUPANCY_ONE                                  \
    __shared__ double ldsHome[40 * 1024 / 8];           \
    if (seed == -1.0) ldsHome[threadIdx.x] = seed;      \
    __builtin_amdgcn_sched_barrier(0);

template <int SPACING>
__device__ __forceinline__ double Spin(double v) {
#pragma unroll
    for (int s = 0; s < SPACING; ++s) v = __builtin_fma(v, 1.0000001, 1e-9);
    return v;
}

template <int AUX, int SPACING>
__global__ __launch_bounds__(64) void Wave8(double* out, double seed) {
    OCCUPANCY_ONE
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + static_cast<long long>(blockIdx.x) * kTileDoubles, 0, 0xFFFFFFFF, 0x00020000);
    const int voff = static_cast<int>(threadIdx.x) * 8;
    double v = seed + static_cast<double>(threadIdx.x);
#pragma unroll 1
    for (int s = 0; s < kSinks; s += 2) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, voff, s * 512, AUX);
        v = Spin<SPACING>(v);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, voff, s * 512 + 512, AUX);
        v = Spin<SPACING>(v);
    }
}

// SHIFT: byte offset of the tile inside the buffer (is 1 KiB ALIGNMENT of the 1 KiB instruction needed, or only contiguity?)
template <int AUX, int SPACING, int SHIFT>
__global__ __launch_bounds__(64) void Wave16(double* out, double seed) {
    OCCUPANCY_ONE
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out + static_cast<long long>(blockIdx.x) * kTileDoubles) + SHIFT, 0, 0xFFFFFFFF, 0x00020000);
    const int voff = static_cast<int>(threadIdx.x) * 16;
    double v = seed + static_cast<double>(threadIdx.x);
#pragma unroll 1
    for (int s = 0; s < kSinks / 2; ++s) {
        const v2d q{v, v + 1.0};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q), r, voff, s * 1024, AUX);
        v = Spin<2 * SPACING>(v);
    }
}


// BAND: the tiles of BAND consecutive wavefronts are interleaved at the granularity of one store instruction -- [band][sink pair][wave in band][1 KiB] -- so
// that wavefronts that run at the same time fill whole DRAM pages together (BAND = 1: plain tiles; BAND = all wavefronts: "unit-fastest over register images")
template <int AUX, int SPACING>
__global__ __launch_bounds__(64) void Band16(double* out, double seed, int band) {
    OCCUPANCY_ONE
    const long long w = blockIdx.x;
    const long long first = (w / band) * (kSinks / 2) * band + (w % band);  // in KiB
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + first * 1024, 0, 0xFFFFFFFF, 0x00020000);
    const int voff = static_cast<int>(threadIdx.x) * 16;
    const int step = band * 1024;
    double v = seed + static_cast<double>(threadIdx.x);
    int soff = 0;
#pragma unroll 1
    for (int s = 0; s < kSinks / 2; ++s) {
        const v2d q{v, v + 1.0};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q), r, voff, soff, AUX);
        soff += step;
        v = Spin<2 * SPACING>(v);
    }
}
template <int AUX, int SPACING>
__global__ __launch_bounds__(64) void Band8(double* out, double seed, int band) {
    OCCUPANCY_ONE
    const long long w = blockIdx.x;
    const long long first = (w / band) * kSinks * band + (w % band);  // in 512 B
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + first * 512, 0, 0xFFFFFFFF, 0x00020000);
    const int voff = static_cast<int>(threadIdx.x) * 8;
    const int step = band * 512;
    double v = seed + static_cast<double>(threadIdx.x);
    int soff = 0;
#pragma unroll 1
    for (int s = 0; s < kSinks; ++s) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, voff, soff, AUX);
        soff += step;
        v = Spin<SPACING>(v);
    }
}

// the exchange of the product kernel in front of every 16-byte store (v_permlane16_swap x 2)
template <int AUX, int SPACING>
__global__ __launch_bounds__(64) void Wave16Swap(double* out, double seed) {
    OCCUPANCY_ONE
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + static_cast<long long>(blockIdx.x) * kTileDoubles, 0, 0xFFFFFFFF, 0x00020000);
    // even 16-lane rows write the first sink of the pair, odd rows the second: (row >> 1, leg, j) -> nodes 8 R + 2 j, + 1
    const int lane = static_cast<int>(threadIdx.x), row = lane >> 4, leg = (lane >> 2) & 3, j = lane & 3;
    const int voff = ((row & 1) * 64 + leg * 16 + 8 * (row >> 1) + 2 * j) * 8;
    double v = seed + static_cast<double>(threadIdx.x), w = v * 0.5;
#pragma unroll 1
    for (int s = 0; s < kSinks / 2; ++s) {
        const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(v)), static_cast<unsigned>(__double2loint(w)), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(v)), static_cast<unsigned>(__double2hiint(w)), false, false);
        const v4i q{static_cast<int>(lo[0]), static_cast<int>(hi[0]), static_cast<int>(lo[1]), static_cast<int>(hi[1])};
        __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, s * 1024, AUX);
        asm volatile("s_nop 1" ::"v"(q) : "memory");
        v = Spin<SPACING>(v);
        w = Spin<SPACING>(w);
    }
}

// [tile][column][row][16]: 49 columns x (4 x 1 KiB + 640 B by 40 lanes)
template <int AUX, int SPACING>
__global__ __launch_bounds__(64) void Col16(double* out, double seed) {
    OCCUPANCY_ONE
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + static_cast<long long>(blockIdx.x) * (kEntries * 16), 0, 0xFFFFFFFF, 0x00020000);
    const int voff = static_cast<int>(threadIdx.x) * 16;
    double v = seed + static_cast<double>(threadIdx.x);
#pragma unroll 1
    for (int c = 0; c < 49; ++c) {
        const int base = c * 37 * 128;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v2d q{v, v + 1.0};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q), r, voff, base + k * 1024, AUX);
            v = Spin<2 * SPACING>(v);
        }
        if (threadIdx.x < 40) {
            const v2d q{v, v + 1.0};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, q), r, voff, base + 4096, AUX);
        }
        v = Spin<2 * SPACING>(v);
    }
}

// the product's present pattern: unit-fastest [entry][node], lane-per-leg, 8-byte stores, 4 x 128 B per instruction
template <int AUX, int SPACING>
__global__ __launch_bounds__(64) void Quad8(double* out, long long count, double seed) {
    OCCUPANCY_ONE
    const int L = (threadIdx.x >> 2) & 3;
    const int nodeInWave = 4 * (threadIdx.x >> 4) + (threadIdx.x & 3);
    const long long i = static_cast<long long>(blockIdx.x) * 16 + nodeInWave;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned je8 = static_cast<unsigned>(count) * 8u;
    const int vLeg = static_cast<int>((static_cast<unsigned>(i) + static_cast<unsigned>(L * 453) * static_cast<unsigned>(count)) * 8u);
    double v = seed + static_cast<double>(threadIdx.x);
#pragma unroll 1
    for (int e = 0; e < 453; ++e) {  // 4 x 453 = 1812 entries
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, vLeg, static_cast<int>(static_cast<unsigned>(e) * je8), AUX);
        v = Spin<SPACING>(v);
    }
}

int main(int argc, char** argv) {
    const long long count = 81920;
    const char* only = argc > 1 ? argv[1] : nullptr;  // run one variant only (counter passes)
    const int reps = argc > 2 ? std::atoi(argv[2]) : 50;
    double* out;
    const size_t bytesAlloc = static_cast<size_t>(kTileDoubles) * (count / 16) * 8 + 4096;
    CK(hipMalloc(&out, bytesAlloc));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = static_cast<double>(kEntries) * count * 8;
    auto time = [&](const char* name, auto launch) {
        if (only && std::strcmp(only, name) != 0) return;
        for (int i = 0; i < (only ? 20 : 400); ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::printf("%-28s %.4f ms  %.2f TB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
    };
    const dim3 g(static_cast<unsigned>(count / 16)), b(64);
    time("quad8_nt", [&] { hipLaunchKernelGGL((Quad8<2, 0>), g, b, 0, 0, out, count, 1.0); });
    time("quad8_nt_s18", [&] { hipLaunchKernelGGL((Quad8<2, 18>), g, b, 0, 0, out, count, 1.0); });
    time("wave8_nt", [&] { hipLaunchKernelGGL((Wave8<2, 0>), g, b, 0, 0, out, 1.0); });
    time("wave8_wb", [&] { hipLaunchKernelGGL((Wave8<0, 0>), g, b, 0, 0, out, 1.0); });
    time("wave8_nt_s18", [&] { hipLaunchKernelGGL((Wave8<2, 18>), g, b, 0, 0, out, 1.0); });
    time("wave16_nt", [&] { hipLaunchKernelGGL((Wave16<2, 0, 0>), g, b, 0, 0, out, 1.0); });
    time("wave16_wb", [&] { hipLaunchKernelGGL((Wave16<0, 0, 0>), g, b, 0, 0, out, 1.0); });
    time("wave16_nt_shift512", [&] { hipLaunchKernelGGL((Wave16<2, 0, 512>), g, b, 0, 0, out, 1.0); });
    time("wave16_nt_shift128", [&] { hipLaunchKernelGGL((Wave16<2, 0, 128>), g, b, 0, 0, out, 1.0); });
    time("wave16_nt_s18", [&] { hipLaunchKernelGGL((Wave16<2, 18, 0>), g, b, 0, 0, out, 1.0); });
    time("wave16_wb_s18", [&] { hipLaunchKernelGGL((Wave16<0, 18, 0>), g, b, 0, 0, out, 1.0); });
    time("wave16swap_nt", [&] { hipLaunchKernelGGL((Wave16Swap<2, 0>), g, b, 0, 0, out, 1.0); });
    time("wave16swap_nt_s18", [&] { hipLaunchKernelGGL((Wave16Swap<2, 18>), g, b, 0, 0, out, 1.0); });
    time("col16_nt", [&] { hipLaunchKernelGGL((Col16<2, 0>), g, b, 0, 0, out, 1.0); });
    time("col16_wb", [&] { hipLaunchKernelGGL((Col16<0, 0>), g, b, 0, 0, out, 1.0); });
    time("col16_nt_s18", [&] { hipLaunchKernelGGL((Col16<2, 18>), g, b, 0, 0, out, 1.0); });
    for (int band : {1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 5120}) {
        char name[64];
        std::snprintf(name, sizeof name, "band16_nt_%d", band);
        time(name, [&] { hipLaunchKernelGGL((Band16<2, 0>), g, b, 0, 0, out, 1.0, band); });
    }
    for (int band : {1, 8, 64, 1024, 5120}) {
        char name[64];
        std::snprintf(name, sizeof name, "band16_nt_s18_%d", band);
        time(name, [&] { hipLaunchKernelGGL((Band16<2, 18>), g, b, 0, 0, out, 1.0, band); });
        std::snprintf(name, sizeof name, "band16_wb_%d", band);
        time(name, [&] { hipLaunchKernelGGL((Band16<0, 0>), g, b, 0, 0, out, 1.0, band); });
        std::snprintf(name, sizeof name, "band8_nt_%d", band);
        time(name, [&] { hipLaunchKernelGGL((Band8<2, 0>), g, b, 0, 0, out, 1.0, band); });
    }
    CK(hipMemsetAsync(out, 0, static_cast<size_t>(bytes), 0));
    time("memset", [&] { hipMemsetAsync(out, 0, static_cast<size_t>(bytes), 0); });
    return 0;
}
