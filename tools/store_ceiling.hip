// What the result stores of the ANYmal node kernels can reach on their own: a store-only kernel that writes the 37 + 37 x 49
// doubles of 81 920 nodes in the unit-fastest layout ([entry][node]) with the same instructions as the node kernels
// (raw_buffer_store_b64, non-temporal or not), for the lane layouts in question:
//   quad : lane = 16 * row + 4 * leg + node % 4, 16 nodes per wavefront, a store = 4 entries x 128 B (the lane-per-leg kernels)
//   node : lane = node, 64 nodes per wavefront, a store = 1 entry x 512 B (what a lane-per-node kernel writes)
//   pair : two entries per lane and instruction (16-byte stores, entries e and e + 1 of a node are NOT adjacent in memory: timing
//          of wider instructions on the node layout -- [entry pair][node][2])
// and for 1 / 2 / 4 wavefronts per SIMD (register budget forced with amdgpu_waves_per_eu).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int kEntries = 37 * 49;

template <bool NT>
__device__ __forceinline__ void Put(__amdgpu_buffer_rsrc_t r, int voff, unsigned soff, double v) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, voff, static_cast<int>(soff), NT ? 2 : 0);
}

// quad layout: every lane writes the entries of its leg's row block (6 rows x 49) + its share of the 13 base rows -> 599 stores like the real kernel
template <bool NT, int WAVES, int SPACING>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void QuadStores(double* out, long long count, double seed) {
    const int L = (threadIdx.x >> 2) & 3;
    const int nodeInWave = 4 * (threadIdx.x >> 4) + (threadIdx.x & 3);
    const long long i = static_cast<long long>(blockIdx.x) * 16 + nodeInWave;
    if (i >= count) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned je8 = static_cast<unsigned>(count) * 8u;
    const int vLeg = static_cast<int>((static_cast<unsigned>(i) + static_cast<unsigned>((7 + 3 * L) * 49) * static_cast<unsigned>(count)) * 8u);
    const int vLeg2 = static_cast<int>((static_cast<unsigned>(i) + static_cast<unsigned>((25 + 3 * L) * 49) * static_cast<unsigned>(count)) * 8u);
    const int vNode = static_cast<int>(static_cast<unsigned>(i) * 8u);
    double v = seed + static_cast<double>(threadIdx.x);
    // leg rows: 3 + 3 rows x 49 columns
#pragma unroll 1
    for (int c = 0; c < 49; ++c) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            Put<NT>(r, vLeg, static_cast<unsigned>(rr * 49 + c) * je8, v);
            Put<NT>(r, vLeg2, static_cast<unsigned>(rr * 49 + c) * je8, v);
            if (SPACING > 0) {
                for (int s = 0; s < SPACING; ++s) v = __builtin_fma(v, 1.0000001, 1e-9);
            }
        }
    }
    // base rows (0..6, 19..24) of the 9 columns this leg owns: per-lane column offset, 117 stores
    const int vOwn = static_cast<int>((static_cast<unsigned>(i) + static_cast<unsigned>(3 * L) * static_cast<unsigned>(count)) * 8u);
#pragma unroll 1
    for (int c = 0; c < 9; ++c) {
        const int col = (c < 3 ? 7 : c < 6 ? 25 - 3 : 37 - 6) + c;
#pragma unroll
        for (int rr = 0; rr < 13; ++rr) {
            const int row = rr < 7 ? rr : 19 + (rr - 7);
            Put<NT>(r, vOwn, static_cast<unsigned>(row * 49 + col) * je8, v);
            if (SPACING > 0) {
                for (int s = 0; s < SPACING; ++s) v = __builtin_fma(v, 1.0000001, 1e-9);
            }
        }
    }
    // base rows of the 13 shared columns (0..6, 19..24): the four lanes of a node write the same value to the same address, 169 stores
#pragma unroll 1
    for (int c = 0; c < 13; ++c) {
        const int col = c < 7 ? c : 19 + (c - 7);
#pragma unroll
        for (int rr = 0; rr < 13; ++rr) {
            const int row = rr < 7 ? rr : 19 + (rr - 7);
            Put<NT>(r, vNode, static_cast<unsigned>(row * 49 + col) * je8, v);
            if (SPACING > 0) {
                for (int s = 0; s < SPACING; ++s) v = __builtin_fma(v, 1.0000001, 1e-9);
            }
        }
    }
}

// node layout: lane = node, all 1813 entries
template <bool NT, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void NodeStores(double* out, long long count, double seed) {
    const long long i = static_cast<long long>(blockIdx.x) * 64 + threadIdx.x;
    if (i >= count) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned je8 = static_cast<unsigned>(count) * 8u;
    const int vNode = static_cast<int>(static_cast<unsigned>(i) * 8u);
    const double v = seed + static_cast<double>(threadIdx.x);
#pragma unroll 1
    for (int e = 0; e < kEntries; e += 7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) Put<NT>(r, vNode, static_cast<unsigned>(e + k) * je8, v);
    }
}

template <int AUX>
__device__ __forceinline__ void Put4(__amdgpu_buffer_rsrc_t r, int voff, unsigned soff, double v0, double v1) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v{v0, v1};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, voff, static_cast<int>(soff), AUX);
}

// node-pair layout: lane = 2 nodes, 16-byte stores, 1 KiB contiguous per instruction
template <int AUX, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void PairStores(double* out, long long count, double seed) {
    const long long i = (static_cast<long long>(blockIdx.x) * 64 + threadIdx.x) * 2;
    if (i >= count) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned je8 = static_cast<unsigned>(count) * 8u;
    const int vNode = static_cast<int>(static_cast<unsigned>(i) * 8u);
    const double v = seed + static_cast<double>(threadIdx.x);
#pragma unroll 1
    for (int e = 0; e < kEntries; e += 7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) Put4<AUX>(r, vNode, static_cast<unsigned>(e + k) * je8, v, v);
    }
}

// quad layout, 16-byte stores after a pairwise exchange: of the 4 adjacent lanes of a leg (nodes n..n+3) the even ones write entry e of
// nodes (n, n+1) / (n+2, n+3), the odd ones entry e' = e + delta of the same node pairs: half the store instructions, same 32-byte runs
template <int AUX, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void QuadPairStores(double* out, long long count, double seed) {
    const int L = (threadIdx.x >> 2) & 3;
    const int nodeInWave = 4 * (threadIdx.x >> 4) + (threadIdx.x & 2);  // first node of the lane's pair
    const int odd = threadIdx.x & 1;
    const long long i = static_cast<long long>(blockIdx.x) * 16 + nodeInWave;
    if (i >= count) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned je8 = static_cast<unsigned>(count) * 8u;
    // odd lanes write the NEXT entry (e + 1): part of the lane offset
    const unsigned oddOff = odd ? static_cast<unsigned>(count) : 0u;
    const int vLeg = static_cast<int>((static_cast<unsigned>(i) + oddOff + static_cast<unsigned>((7 + 3 * L) * 49) * static_cast<unsigned>(count)) * 8u);
    const int vLeg2 = static_cast<int>((static_cast<unsigned>(i) + oddOff + static_cast<unsigned>((25 + 3 * L) * 49) * static_cast<unsigned>(count)) * 8u);
    const int vNode = static_cast<int>((static_cast<unsigned>(i) + oddOff) * 8u);
    const int vOwn = static_cast<int>((static_cast<unsigned>(i) + oddOff + static_cast<unsigned>(3 * L) * static_cast<unsigned>(count)) * 8u);
    const double v = seed + static_cast<double>(threadIdx.x);
    // leg rows: 6 x 49 entries = 147 pairs
#pragma unroll 1
    for (int e = 0; e < 3 * 49 - 1; e += 2) {
        Put4<AUX>(r, vLeg, static_cast<unsigned>(e) * je8, v, v);
        Put4<AUX>(r, vLeg2, static_cast<unsigned>(e) * je8, v, v);
    }
    Put4<AUX>(r, vLeg, static_cast<unsigned>(3 * 49 - 2) * je8, v, v);
    Put4<AUX>(r, vLeg2, static_cast<unsigned>(3 * 49 - 2) * je8, v, v);
    // own base rows: 117 entries -> 59 pairs;  shared base rows: 169 entries, all four lanes the same address: 85 stores (timing model)
#pragma unroll 1
    for (int e = 0; e < 59; ++e) Put4<AUX>(r, vOwn, static_cast<unsigned>((e % 13 < 7 ? e % 13 : 12 + e % 13) * 49 + 7 + 2 * (e / 13)) * je8, v, v);
#pragma unroll 1
    for (int e = 0; e < 85; ++e) Put4<AUX>(r, vNode, static_cast<unsigned>((e % 13 < 7 ? e % 13 : 12 + e % 13) * 49 + (e / 13 < 3 ? 2 * (e / 13) : 19 + 2 * (e / 13 - 3))) * je8, v, v);
}

// tiled layout [tile of 16 nodes][entry][16 nodes]: a wavefront's whole block (232 KB) is contiguous
template <int AUX, int WAVES, bool PAIRED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void TiledStores(double* out, long long count, double seed) {
    const int L = (threadIdx.x >> 2) & 3;
    const int row = threadIdx.x >> 4, j = threadIdx.x & 3;
    const int nodeInWave = PAIRED ? 8 * (row >> 1) + 2 * j + (row & 1) : 4 * row + j;
    const int odd = PAIRED ? (row & 1) : 0;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + static_cast<long long>(blockIdx.x) * kEntries * 16, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned je8 = 16u * 8u;
    auto off = [&](int entry, int deltaEntries) { return static_cast<int>((static_cast<unsigned>(entry + (odd ? deltaEntries : 0)) * 16u + static_cast<unsigned>(nodeInWave - odd)) * 8u); };
    const int vLeg = off((7 + 3 * L) * 49, 18 * 49), vNode = off(0, 49), vOwn = off(3 * L, 49);
    const double v = seed + static_cast<double>(threadIdx.x);
    typedef int v2i __attribute__((ext_vector_type(2)));
    if constexpr (PAIRED) {
#pragma unroll 1
        for (int e = 0; e < 3 * 49; ++e) Put4<AUX>(r, vLeg, static_cast<unsigned>(e) * je8, v, v);   // rows 7+k and 25+k together
#pragma unroll 1
        for (int c = 0; c < 9; ++c)
            for (int rr = 0; rr < 7; ++rr) Put4<AUX>(r, vOwn, static_cast<unsigned>((rr < 4 ? 2 * rr : 19 + 2 * (rr - 4)) * 49 + (c < 3 ? 7 : c < 6 ? 22 : 31) + c) * je8, v, v);
#pragma unroll 1
        for (int c = 0; c < 13; ++c)
            for (int rr = 0; rr < 7; ++rr) Put4<AUX>(r, vNode, static_cast<unsigned>((rr < 4 ? 2 * rr : 19 + 2 * (rr - 4)) * 49 + (c < 7 ? c : 12 + c)) * je8, v, v);
    } else {
#pragma unroll 1
        for (int e = 0; e < 3 * 49; ++e) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, vLeg, static_cast<int>(static_cast<unsigned>(e) * je8), AUX);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, vLeg, static_cast<int>(static_cast<unsigned>(e + 18 * 49) * je8), AUX);
        }
#pragma unroll 1
        for (int c = 0; c < 9; ++c)
            for (int rr = 0; rr < 13; ++rr)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, vOwn, static_cast<int>(static_cast<unsigned>((rr < 7 ? rr : 12 + rr) * 49 + (c < 3 ? 7 : c < 6 ? 22 : 31) + c) * je8), AUX);
#pragma unroll 1
        for (int c = 0; c < 13; ++c)
            for (int rr = 0; rr < 13; ++rr)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, vNode, static_cast<int>(static_cast<unsigned>((rr < 7 ? rr : 12 + rr) * 49 + (c < 7 ? c : 12 + c)) * je8), AUX);
    }
}

// plain fill: grid-stride 16-byte stores over the whole buffer
template <int AUX>
__global__ __launch_bounds__(256) void Fill(double* out, long long doubles, double seed) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xFFFFFFFF, 0x00020000);
    const long long stride = static_cast<long long>(gridDim.x) * 256 * 2;
    for (long long i = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 2; i < doubles; i += stride)
        Put4<AUX>(r, static_cast<int>(static_cast<unsigned>(i) * 8u), 0u, seed, seed);
}

int main() {
    const long long count = 81920;
    double* out;
    CK(hipMalloc(&out, static_cast<size_t>(kEntries) * count * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 200; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::printf("%-44s %.4f ms  %.2f TB/s\n", name, ms / 50, bytes / (ms / 50 * 1e-3) / 1e12);
    };
    const double bytes = static_cast<double>(kEntries) * count * 8;
    const dim3 gq(static_cast<unsigned>(count / 16)), gn(static_cast<unsigned>(count / 64));
    time("quad nt  1 wave/SIMD", [&] { hipLaunchKernelGGL((QuadStores<true, 1, 0>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  2 waves/SIMD", [&] { hipLaunchKernelGGL((QuadStores<true, 2, 0>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  4 waves/SIMD", [&] { hipLaunchKernelGGL((QuadStores<true, 4, 0>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  8 waves/SIMD", [&] { hipLaunchKernelGGL((QuadStores<true, 8, 0>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad wb  1 wave/SIMD", [&] { hipLaunchKernelGGL((QuadStores<false, 1, 0>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad wb  8 waves/SIMD", [&] { hipLaunchKernelGGL((QuadStores<false, 8, 0>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  1 wave/SIMD, 8 fma between stores", [&] { hipLaunchKernelGGL((QuadStores<true, 1, 8>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  2 waves/SIMD, 8 fma between stores", [&] { hipLaunchKernelGGL((QuadStores<true, 2, 8>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  1 wave/SIMD, 14 fma between stores", [&] { hipLaunchKernelGGL((QuadStores<true, 1, 14>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad nt  2 waves/SIMD, 14 fma between stores", [&] { hipLaunchKernelGGL((QuadStores<true, 2, 14>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("node nt  1 wave/SIMD", [&] { hipLaunchKernelGGL((NodeStores<true, 1>), gn, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("node nt  8 waves/SIMD", [&] { hipLaunchKernelGGL((NodeStores<true, 8>), gn, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("node wb  8 waves/SIMD", [&] { hipLaunchKernelGGL((NodeStores<false, 8>), gn, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    const dim3 gp(static_cast<unsigned>(count / 128));
    time("pair wb  b128 8 waves/SIMD", [&] { hipLaunchKernelGGL((PairStores<0, 8>), gp, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("pair nt  b128 8 waves/SIMD", [&] { hipLaunchKernelGGL((PairStores<2, 8>), gp, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("pair nt  b128 1 wave/SIMD", [&] { hipLaunchKernelGGL((PairStores<2, 1>), gp, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad-pair wb b128 1 wave/SIMD", [&] { hipLaunchKernelGGL((QuadPairStores<0, 1>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad-pair nt b128 1 wave/SIMD", [&] { hipLaunchKernelGGL((QuadPairStores<2, 1>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad-pair nt b128 2 waves/SIMD", [&] { hipLaunchKernelGGL((QuadPairStores<2, 2>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad-pair nt+sc1 b128 2 waves/SIMD", [&] { hipLaunchKernelGGL((QuadPairStores<18, 2>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("quad-pair sc0sc1 b128 2 waves/SIMD", [&] { hipLaunchKernelGGL((QuadPairStores<17, 2>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("tiled quad  8-byte nt 1 wave/SIMD", [&] { hipLaunchKernelGGL((TiledStores<2, 1, false>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("tiled quad  8-byte wb 1 wave/SIMD", [&] { hipLaunchKernelGGL((TiledStores<0, 1, false>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("tiled pair 16-byte nt 1 wave/SIMD", [&] { hipLaunchKernelGGL((TiledStores<2, 1, true>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("tiled pair 16-byte nt 2 waves/SIMD", [&] { hipLaunchKernelGGL((TiledStores<2, 2, true>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("tiled pair 16-byte wb 1 wave/SIMD", [&] { hipLaunchKernelGGL((TiledStores<0, 1, true>), gq, dim3(64), 0, 0, out, count, 1.0); }, bytes);
    time("fill wb b128 grid 4096", [&] { hipLaunchKernelGGL((Fill<0>), dim3(4096), dim3(256), 0, 0, out, static_cast<long long>(kEntries) * count, 1.0); }, bytes);
    time("fill nt b128 grid 4096", [&] { hipLaunchKernelGGL((Fill<2>), dim3(4096), dim3(256), 0, 0, out, static_cast<long long>(kEntries) * count, 1.0); }, bytes);
    time("fill wb b128 grid 65536", [&] { hipLaunchKernelGGL((Fill<0>), dim3(65536), dim3(256), 0, 0, out, static_cast<long long>(kEntries) * count, 1.0); }, bytes);
    CK(hipMemset(out, 0, static_cast<size_t>(kEntries) * count * 8));
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipMemsetAsync(out, 0, static_cast<size_t>(kEntries) * count * 8, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-44s %.4f ms  %.2f TB/s\n", "hipMemsetAsync", ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
    return 0;
}
