// What ONE result-store instruction costs the wavefront that issues it (round 6): a loop of FP64 work (4 independent FMA chains, 32 FMAs per iteration) with one
// store form per iteration, timed with s_memtime inside the kernel; the cost of the store form = cycles per iteration - cycles of the FMA-only loop.
// Grids: 256 workgroups of one wavefront (one per CU: nothing shared), 1024 (one per SIMD: the four wavefronts of a CU share its vector-memory path, as in the
// node kernel).  Every kernel holds 40 KiB of LDS (one wavefront per SIMD at most).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// FORM 0: no store; 1: 16-byte buffer store (1 KiB / instruction); 2: the same behind two v_permlane16_swap + s_nop (the tile kernel's t_put2);
// 3: two 8-byte buffer stores (512 B each); 4: 16-byte store, write-back policy; 5: four 4-byte stores (256 B each); 6: swaps only (no store)
template <int FORM, int FMAS>
__global__ __launch_bounds__(64) void Loop(double* out, long long* cycles, int iters, double seed) {
    __shared__ double ldsHome[40 * 1024 / 8];
    if (seed == -1.0) ldsHome[threadIdx.x] = seed;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + static_cast<long long>(blockIdx.x) * (1 << 17), 0, 0xFFFFFFFF, 0x00020000);
    const int voff = static_cast<int>(threadIdx.x) * 16;
    double a = seed + threadIdx.x, b = seed * 2 + threadIdx.x, c = seed * 3 + threadIdx.x, d = seed * 4 + threadIdx.x;
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int soff = (it & 127) * 1024;
#pragma unroll
        for (int k = 0; k < FMAS / 4; ++k) {
            a = __builtin_fma(a, 1.0000001, 1e-9);
            b = __builtin_fma(b, 1.0000001, 1e-9);
            c = __builtin_fma(c, 1.0000001, 1e-9);
            d = __builtin_fma(d, 1.0000001, 1e-9);
        }
        if constexpr (FORM == 1 || FORM == 4) {
            const v4i q{__double2loint(a), __double2hiint(a), __double2loint(b), __double2hiint(b)};
            __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, FORM == 4 ? 0 : 2);
        } else if constexpr (FORM == 2 || FORM == 6) {
            const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(a)), static_cast<unsigned>(__double2loint(b)), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(a)), static_cast<unsigned>(__double2hiint(b)), false, false);
            const v4i q{static_cast<int>(lo[0]), static_cast<int>(hi[0]), static_cast<int>(lo[1]), static_cast<int>(hi[1])};
            if constexpr (FORM == 2) {
                __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, 2);
                asm volatile("s_nop 1" ::"v"(q) : "memory");
            } else {
                asm volatile("" ::"v"(q));
            }
        } else if constexpr (FORM == 3) {
            __builtin_amdgcn_raw_buffer_store_b64(v2i{__double2loint(a), __double2hiint(a)}, r, voff / 2, soff, 2);
            __builtin_amdgcn_raw_buffer_store_b64(v2i{__double2loint(b), __double2hiint(b)}, r, voff / 2, soff + 512, 2);
        } else if constexpr (FORM == 5) {
            __builtin_amdgcn_raw_buffer_store_b32(__double2loint(a), r, voff / 4, soff, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__double2hiint(a), r, voff / 4, soff + 256, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__double2loint(b), r, voff / 4, soff + 512, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__double2hiint(b), r, voff / 4, soff + 768, 2);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 12345.678) out[0] = a;
}

int main() {
    double* out;
    long long* cyc;
    const int maxGrid = 1024, iters = 2000;
    CK(hipMalloc(&out, static_cast<size_t>(maxGrid) * (1 << 17) * 8));
    CK(hipMalloc(&cyc, maxGrid * 8));
    std::vector<long long> h(maxGrid);
    auto run = [&](const char* name, auto kernel, int grid, double base) -> double {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.0);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double acc = 0;
        for (int i = 0; i < grid; ++i) acc += static_cast<double>(h[i]);
        const double per = acc / grid / iters;
        std::printf("%-44s grid %4d: %7.1f clock64 ticks per iteration%s\n", name, grid, per, base > 0 ? "" : "");
        if (base > 0) std::printf("%-44s            -> %6.1f ticks per KiB stored\n", "", per - base);
        return per;
    };
    for (int grid : {256, 1024}) {
        const double b32 = run("32 FMAs", Loop<0, 32>, grid, 0);
        run("32 FMAs + b128 nt", Loop<1, 32>, grid, b32);
        run("32 FMAs + swaps + b128 nt + s_nop", Loop<2, 32>, grid, b32);
        run("32 FMAs + swaps only", Loop<6, 32>, grid, b32);
        run("32 FMAs + 2 x b64 nt", Loop<3, 32>, grid, b32);
        run("32 FMAs + b128 write-back", Loop<4, 32>, grid, b32);
        run("32 FMAs + 4 x b32 nt", Loop<5, 32>, grid, b32);
        const double b64 = run("64 FMAs", Loop<0, 64>, grid, 0);
        run("64 FMAs + b128 nt", Loop<1, 64>, grid, b64);
        run("64 FMAs + 2 x b64 nt", Loop<3, 64>, grid, b64);
        const double b128 = run("128 FMAs", Loop<0, 128>, grid, 0);
        run("128 FMAs + b128 nt", Loop<1, 128>, grid, b128);
        run("128 FMAs + swaps + b128 nt + s_nop", Loop<2, 128>, grid, b128);
    }
    return 0;
}
