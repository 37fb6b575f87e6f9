// Round 6: the headline kernel with wave-tile result stores (csrc/kernels/quad_tile_kernel.hpp) against the product's unit-fastest kernel with paired
// 16-byte stores, on the same 81 920 nodes: every entry of the tile operand compared with the unit-fastest block through the generated slot table,
// then timings (steady-state clocks) of both, of the tile kernel with its stores behind a wave-uniform flag (flag = 0: the compute floor of the
// program; flag = 1: what the guard itself costs).
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++20 -I <gen dir> -o build/variants/<name> tools/quad_tile_bench.hip
//   run  :  <name> [label] [count] [mode]     mode "pmc": a few launches of each kernel only (counter passes)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "anymal_quad_gen.hpp"
#include "anymal_tiles_gen.hpp"
#include "../ungar_amd/csrc/kernels/quad_tile_kernel.hpp"

using namespace ungar_amd::kernels;
namespace Q = ungar_amd::gen::anymal_quad;
namespace T = ungar_amd::gen::anymal_tiles;

struct Body {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { Q::ValueJacobianQuad<double>(io); }
};
struct TileBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { T::ValueJacobianQuadTiles<double>(io); }
};

// per-phase timestamps (s_memtime) of one lane of every 64th wavefront: where does the time go with the stores on / off?
template <bool GUARD>
struct TimedTileIO : QuadTileIO<true, GUARD> {
    long long* ts;
    mutable int k = 0;
    __device__ __forceinline__ void phase() const {
        __builtin_amdgcn_sched_barrier(0);
        if (ts) ts[k++] = clock64();
        __builtin_amdgcn_sched_barrier(0);
    }
};
template <int LDS_SLOTS, int LDS_USLOTS, bool GUARD>
__global__ __launch_bounds__(64) void TimedTileKernel(const NodeLaunch a, const double (*ctab)[4], double* tiles, int images, int storeFlag, long long* stamps) {
    __shared__ double lds[LDS_SLOTS * 64 + LDS_USLOTS * 16];
    const int lane = static_cast<int>(threadIdx.x);
    const int L = (lane >> 2) & 3;
    const int nodeInWave = QuadNodeInWave<false>(lane);
    long long i = static_cast<long long>(blockIdx.x) * kTileNodes + nodeInWave;
    if (i >= a.count) i = a.count - 1;
    double* const fb = a.f.base + i;
    TimedTileIO<GUARD> io{{{a.x.base + i, a.u.base + i, a.p.base, fb, nullptr, a.x.es, a.u.es, a.f.es, 0u, L, nullptr, {nullptr, nullptr, nullptr, nullptr}, nullptr,
                            fb + 3LL * L * a.f.es, ctab, {}, lds + threadIdx.x, lds + LDS_SLOTS * 64 + nodeInWave, {}}},
                          (stamps && threadIdx.x == 0 && blockIdx.x % 64 == 0) ? stamps + (blockIdx.x / 64) * 32 : nullptr};
#if defined(__HIP_DEVICE_COMPILE__)
    const long long t = blockIdx.x, g = t / kTileBandTiles, r = t % kTileBandTiles;
    char* const first = reinterpret_cast<char*>(tiles) + ((g * (images / 2)) * kTileBandTiles + r) * static_cast<long long>(kTileUnitBytes);
    io.tr = __builtin_amdgcn_make_buffer_rsrc(first, 0, 0xFFFFFFFF, 0x00020000);
    io.tv = lane * 16;
#endif
    io.storeFlag = storeFlag != 0;
    if (io.ts) io.ts[io.k++] = clock64();
    T::ValueJacobianQuadTiles<double>(io);
    if (io.ts) io.ts[io.k++] = clock64();
}

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char** argv) {
    const char* label = argc > 1 ? argv[1] : "";
    const long long count = argc > 2 ? std::atoll(argv[2]) : 81920;
    const bool pmc = argc > 3 && std::strcmp(argv[3], "pmc") == 0;
    const bool noCheck = argc > 3 && std::strcmp(argv[3], "nocheck") == 0;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> x(37 * count), u(12 * count), p(1, 0.01);
    for (long long i = 0; i < count; ++i) {
        double q[4], n = 0;
        for (double& v : q) {
            v = U(rng);
            n += v * v;
        }
        for (int e = 0; e < 37; ++e) x[e * count + i] = U(rng);
        for (int k = 0; k < 4; ++k) x[(3 + k) * count + i] = q[k] / std::sqrt(n);
        for (int e = 0; e < 12; ++e) u[e * count + i] = 20 * U(rng);
    }
    const long long tileDoubles = TileOperandDoubles(count, T::kImages);
    double *dx, *du, *dp, *df, *dj, *df2, *dt;
    CK(hipMalloc(&dx, x.size() * 8));
    CK(hipMalloc(&du, u.size() * 8));
    CK(hipMalloc(&dp, 8));
    CK(hipMalloc(&df, 37 * count * 8));
    CK(hipMalloc(&df2, 37 * count * 8));
    CK(hipMalloc(&dj, 1813 * count * 8));
    CK(hipMalloc(&dt, tileDoubles * 8));
    CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, p.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemset(dj, 0xFF, 1813 * count * 8));
    CK(hipMemset(dt, 0xFF, tileDoubles * 8));
    NodeLaunch a{};
    a.count = count;
    a.knots = 1;
    a.x = {dx, 1, 0, count};
    a.u = {du, 1, 0, count};
    a.p = {dp, 0, 0, 1};
    a.f = {df, 1, 0, count};
    a.jac = {dj, 1, 0, count};
    NodeLaunch at = a;
    at.f = {df2, 1, 0, count};
    at.jac = {nullptr, 0, 0, 0};
    void* sym = nullptr;
    CK(hipGetSymbolAddress(&sym, HIP_SYMBOL(Q::kLegConstantsDev)));
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((count + 15) / 16)), block(64);
    auto product = [&] {
        hipLaunchKernelGGL((QuadNodeKernel<64, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, Body, NoSparsePlan, unsigned, true, true>), grid, block, 0, 0, a, ctab, Body{});
    };
    auto tiled = [&] { hipLaunchKernelGGL((QuadTileKernel<T::kLdsSlots, T::kLdsUniformSlots, true, TileBody>), grid, block, 0, 0, at, ctab, dt, T::kImages, 1, TileBody{}); };
    auto guarded = [&](int flag) {
        hipLaunchKernelGGL((QuadTileKernel<T::kLdsSlots, T::kLdsUniformSlots, true, TileBody, true>), grid, block, 0, 0, at, ctab, dt, T::kImages, flag, TileBody{});
    };
    if (pmc) {
        for (int i = 0; i < 5; ++i) product();
        for (int i = 0; i < 5; ++i) tiled();
        CK(hipDeviceSynchronize());
        return 0;
    }
    product();
    tiled();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    if (!noCheck) {
        std::vector<double> hj(1813 * count), ht(static_cast<std::size_t>(tileDoubles)), hf(37 * count), hf2(37 * count);
        CK(hipMemcpy(hj.data(), dj, hj.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ht.data(), dt, ht.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf.data(), df, hf.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf2.data(), df2, hf2.size() * 8, hipMemcpyDeviceToHost));
        long long differ = 0, unwritten = 0, seen = 0;
        double worst = 0, scale = 0;
        for (int slot = 0; slot < 4 * T::kImages; ++slot) {
            const int e = T::kEntryOfSlot[slot];
            if (e < 0) continue;
            ++seen;
            for (long long i = 0; i < count; ++i) {
                const double tv = ht[static_cast<std::size_t>(TileSlotOffset(i, slot, T::kImages))], uv = hj[static_cast<std::size_t>(e) * count + i];
                if (std::memcmp(&tv, &uv, 8) != 0) {
                    ++differ;
                    if (tv != tv) ++unwritten;
                    else worst = std::fmax(worst, std::fabs(tv - uv));
                }
                scale = std::fmax(scale, std::fabs(uv));
            }
        }
        long long fdiffer = 0;
        for (std::size_t i = 0; i < hf.size(); ++i) fdiffer += std::memcmp(&hf[i], &hf2[i], 8) != 0;
        std::printf("%s check: %lld entries per node compared over %lld nodes: %lld differ in bits (%lld unwritten), max |dJ| %.3e at |J| <= %.3g; f: %lld of %zu differ\n", label,
                    seen, count, differ, unwritten, worst, scale, fdiffer, hf.size());
    }
    if (argc > 3 && std::strcmp(argv[3], "phases") == 0) {
        const int nw = static_cast<int>(grid.x / 64);
        long long* dts = nullptr;
        CK(hipMalloc(&dts, nw * 32 * 8));
        std::vector<long long> ts(nw * 32);
        for (int flag : {1, 0}) {
            for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((TimedTileKernel<T::kLdsSlots, T::kLdsUniformSlots, true>), grid, block, 0, 0, at, ctab, dt, T::kImages, flag, static_cast<long long*>(nullptr));
            CK(hipMemset(dts, 0, nw * 32 * 8));
            hipLaunchKernelGGL((TimedTileKernel<T::kLdsSlots, T::kLdsUniformSlots, true>), grid, block, 0, 0, at, ctab, dt, T::kImages, flag, dts);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ts.data(), dts, ts.size() * 8, hipMemcpyDeviceToHost));
            std::printf("%s stores %s: cycles per phase, mean over %d wavefronts (first generation | later generations):\n", label, flag ? "ON " : "OFF", nw);
            for (int gen = 0; gen < 2; ++gen) {
                double total = 0;
                std::printf("  %s", gen ? "later:" : "first:");
                for (int k = 0; k + 1 < 32; ++k) {
                    double acc = 0;
                    int cnt = 0;
                    for (int w = 0; w < nw; ++w) {
                        const bool first = w * 64 < 1024;
                        if (first != (gen == 0)) continue;
                        if (ts[w * 32 + k + 1] && ts[w * 32 + k]) acc += static_cast<double>(ts[w * 32 + k + 1] - ts[w * 32 + k]), ++cnt;
                    }
                    if (cnt) {
                        std::printf(" %.0f", acc / cnt);
                        total += acc / cnt;
                    }
                }
                std::printf("  | total %.0f\n", total);
            }
        }
        return 0;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch, int reps) {
        for (int i = 0; i < 1200; ++i) {  // steady-state clocks (bench.py pre-warms as well)
            launch();
            if (i % 100 == 99) hipDeviceSynchronize();
        }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::printf("%s %-34s kernel_ms=%.4f frac=%.4f\n", label, name, ms / reps, count * 15192.0 / (ms / reps * 1e-3) / 8e12);
    };
    for (int round = 0; round < 2; ++round) {
        time("unit-fastest, paired stores", product, 100);
        time("wave tiles", tiled, 100);
        time("wave tiles, guarded stores ON", [&] { guarded(1); }, 100);
        time("wave tiles, guarded stores OFF", [&] { guarded(0); }, 100);
    }
    return 0;
}
