// Standalone harness for the SPLIT lane-per-leg ANYmal program (quad_split_kernel.hpp): runs it next to the fused kernel on the
// same 81 920 nodes, compares every value and every entry of the dense block, and times both at steady-state clocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I <gen dir> -o build/variants/<name> tools/quad_split_bench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "anymal_quad_gen.hpp"
#include "anymal_split_gen.hpp"
#include "../ungar_amd/csrc/kernels/quad_split_kernel.hpp"

using namespace ungar_amd::kernels;
namespace Q = ungar_amd::gen::anymal_quad;
namespace S = ungar_amd::gen::anymal_split;

#ifndef SPLIT_RING
#define SPLIT_RING 3
#endif

struct Body {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { Q::ValueJacobianQuad<double>(io); }
};
struct ProducerBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const {
#ifndef SPLIT_NO_PRODUCER  // resource-usage experiments: one half alone
        S::ProducerQuad<double>(io);
#endif
    }
};
struct ConsumerBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const {
#ifndef SPLIT_NO_CONSUMER
        S::ConsumerQuad<double>(io);
#endif
    }
};

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char** argv) {
    const long long count = argc > 2 ? std::atoll(argv[2]) : 81920;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> x(37 * count), u(12 * count), p(1, 0.05);
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
    double *dx, *du, *dp, *df, *dj, *df2, *dj2;
    CK(hipMalloc(&dx, x.size() * 8));
    CK(hipMalloc(&du, u.size() * 8));
    CK(hipMalloc(&dp, 8));
    CK(hipMalloc(&df, 37 * count * 8));
    CK(hipMalloc(&dj, 1813 * count * 8));
    CK(hipMalloc(&df2, 37 * count * 8));
    CK(hipMalloc(&dj2, 1813 * count * 8));
    CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, p.data(), 8, hipMemcpyHostToDevice));
    CK(hipMemset(dj2, 0xFF, 1813 * count * 8));  // NaN: every entry must be written
    CK(hipMemset(df2, 0xFF, 37 * count * 8));
    NodeLaunch a{};
    a.count = count;
    a.knots = 1;
    a.x = {dx, 1, 0, count};
    a.u = {du, 1, 0, count};
    a.p = {dp, 0, 0, 1};
    a.f = {df, 1, 0, count};
    a.jac = {dj, 1, 0, count};
    NodeLaunch a2 = a;
    a2.f.base = df2;
    a2.jac.base = dj2;
    void* sym = nullptr;
    CK(hipGetSymbolAddress(&sym, HIP_SYMBOL(Q::kLegConstantsDev)));
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((count + 15) / 16));
    constexpr int PS = S::kProducerLdsSlots, PU = S::kProducerLdsUniformSlots, CS = S::kConsumerLdsSlots, CU = S::kConsumerLdsUniformSlots;
    struct Variant {
        const char* name;
        void (*launch)(dim3, const NodeLaunch&, const double (*)[4]);
    };
    // reference: the fused kernel as the library launches it today (8-byte non-temporal buffer stores)
    const Variant reference{"fused  8-byte nt", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
                                hipLaunchKernelGGL((QuadNodeKernel<64, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, Body, NoSparsePlan, unsigned, true>), g, dim3(64), 0, 0, a, ctab, Body{});
                            }};
    const Variant variants[] = {
        {"fused 16-byte wb", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
             hipLaunchKernelGGL((QuadNodeKernel<64, Q::kLdsSlots, Q::kLdsUniformSlots, false, false, Body, NoSparsePlan, unsigned, true, true>), g, dim3(64), 0, 0, a, ctab, Body{});
         }},
        {"fused 16-byte nt", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
             hipLaunchKernelGGL((QuadNodeKernel<64, Q::kLdsSlots, Q::kLdsUniformSlots, false, true, Body, NoSparsePlan, unsigned, true, true>), g, dim3(64), 0, 0, a, ctab, Body{});
         }},
        {"fused  8-byte wb", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
             hipLaunchKernelGGL((QuadNodeKernel<64, Q::kLdsSlots, Q::kLdsUniformSlots, false, false, Body, NoSparsePlan, unsigned, true>), g, dim3(64), 0, 0, a, ctab, Body{});
         }},
        {"split  8-byte nt", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
             hipLaunchKernelGGL((QuadSplitKernel<PS, PU, CS, CU, SPLIT_RING, true, ProducerBody, ConsumerBody, unsigned, true>), g, dim3(128), 0, 0, a, ctab, ProducerBody{}, ConsumerBody{});
         }},
        {"split 16-byte nt", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
             hipLaunchKernelGGL((QuadSplitKernel<PS, PU, CS, CU, SPLIT_RING, true, ProducerBody, ConsumerBody, unsigned, true, true>), g, dim3(128), 0, 0, a, ctab, ProducerBody{},
                                ConsumerBody{});
         }},
        {"split 16-byte wb", [](dim3 g, const NodeLaunch& a, const double(*ctab)[4]) {
             hipLaunchKernelGGL((QuadSplitKernel<PS, PU, CS, CU, SPLIT_RING, false, ProducerBody, ConsumerBody, unsigned, true, true>), g, dim3(128), 0, 0, a, ctab, ProducerBody{},
                                ConsumerBody{});
         }},
    };
    reference.launch(grid, a, ctab);
    CK(hipDeviceSynchronize());
    std::vector<double> j1(1813 * count), j2(1813 * count), f1(37 * count), f2(37 * count);
    CK(hipMemcpy(j1.data(), dj, j1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(f1.data(), df, f1.size() * 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeIt = [&](const Variant& v, const NodeLaunch& args) -> int {
        const bool quick = std::getenv("QUAD_BENCH_QUICK") != nullptr;  // profiler runs: a handful of launches per variant
        for (int i = 0; i < (quick ? 10 : 1800); ++i) {  // steady-state clocks first (bench.py does the same)
            v.launch(grid, args, ctab);
            if (i % 100 == 99) CK(hipDeviceSynchronize());
        }
        CK(hipDeviceSynchronize());
        const int reps = quick ? 5 : 100;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) v.launch(grid, args, ctab);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%s %-18s kernel_ms=%.4f frac=%.4f\n", argc > 1 ? argv[1] : "", v.name, ms / reps, count * 15192.0 / (ms / reps * 1e-3) / 8e12);
        return 0;
    };
    if (timeIt(reference, a)) return 1;
    const char* only = argc > 3 ? argv[3] : nullptr;  // substring of the variant names to run
    for (const Variant& v : variants) {
        if (only && !std::strstr(v.name, only)) continue;
        CK(hipMemset(dj2, 0xFF, 1813 * count * 8));  // NaN: every entry must be written
        CK(hipMemset(df2, 0xFF, 37 * count * 8));
        v.launch(grid, a2, ctab);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(j2.data(), dj2, j2.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(f2.data(), df2, f2.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0, worstF = 0;
        long long nans = 0, differ = 0;
        int shown = 0;
        std::vector<int> byEntry(1813, 0), byLane(16, 0);
        for (std::size_t i = 0; i < j1.size(); ++i) {
            if (std::isnan(j2[i])) ++nans;
            const double d = std::fabs(j1[i] - j2[i]);
            if (d > worst) worst = d;
            if (d != 0) ++differ;
            if (d > 1e-9 * (1.0 + std::fabs(j1[i]))) {
                ++shown;
                ++byEntry[i / count];
                ++byLane[(i % count) % 16];
            }
            scale = std::fmax(scale, std::fabs(j1[i]));
        }
        for (std::size_t i = 0; i < f1.size(); ++i) {
            if (std::isnan(f2[i])) ++nans;
            worstF = std::fmax(worstF, std::fabs(f1[i] - f2[i]));
        }
        if (shown) {
            std::printf("    %d entries off by more than 1e-9: by node %% 16:", shown);
            for (int l = 0; l < 16; ++l) std::printf(" %d", byLane[l]);
            std::printf("\n    by (row, col):");
            int printed = 0;
            for (int e = 0; e < 1813 && printed < 40; ++e)
                if (byEntry[e]) std::printf(" (%d,%d):%d", e / 49, e % 49, byEntry[e]), ++printed;
            std::printf("\n");
        }
        std::printf("%s %-18s vs reference: max |dJ| %.3e (max |J| %.3e), max |df| %.3e, unwritten %lld, entries that differ %lld of %zu\n", argc > 1 ? argv[1] : "", v.name, worst,
                    scale, worstF, nans, differ, j1.size());
        if (timeIt(v, a2)) return 1;
    }
    return 0;
}
