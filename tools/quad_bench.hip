// Standalone timing harness for variants of the generated lane-per-leg ANYmal program: compiled once per
// variant against a different anymal_quad_gen.hpp (-I <dir>), run on the GPU box by tools/sweep_variants.sh.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <random>
#include <vector>

#include "anymal_quad_gen.hpp"
#include "../ungar_amd/csrc/kernels/quad_kernel.hpp"

using namespace ungar_amd::kernels;
namespace Q = ungar_amd::gen::anymal_quad;

#ifndef BENCH_BUF
#define BENCH_BUF true  // MUBUF result stores: what the launcher picks when the operands span < 4 GiB
#endif
#ifndef BENCH_STREAM
#define BENCH_STREAM true  // non-temporal stores: what the launcher picks for this 1.2 GB unit-fastest output
#endif
struct Body {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { Q::ValueJacobianQuad<double>(io); }
};

// diagnostics: per-phase timestamps of one lane per wavefront (s_memtime), and an output window that wraps
// so that the Jacobian stays cache-resident (separates compute/LDS stalls from HBM write back-pressure)
template <bool TIMED>
struct TimedIOT : QuadIO<false> {
    long long* ts;
    mutable int k = 0;
    __device__ __forceinline__ void phase() const {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TIMED) {
            if (ts) ts[k++] = clock64();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// store-path experiments: SPLIT = every 8-byte store issued as two 4-byte stores (same bytes, twice the
// instructions); WIDE = every store issued as one 16-byte store into a buffer with doubled element size
// (same instructions, twice the bytes)
template <int VARIANT, bool TIMED>
struct StoreIO : TimedIOT<TIMED> {
    using QuadIO<false>::fb;
    using QuadIO<false>::fe;
    using QuadIO<false>::fLeg;
    using QuadIO<false>::jLegCol;
    using QuadIO<false>::jLeg;
    using QuadIO<false>::jOwnCol;
    using QuadIO<false>::jb;
    using QuadIO<false>::je;
    __device__ __forceinline__ void put(double* p, double v) const {
        if constexpr (VARIANT == 1) {
            int* q = reinterpret_cast<int*>(p);
            q[0] = __double2loint(v);
            __builtin_amdgcn_sched_barrier(0);
            q[1] = __double2hiint(v);
        } else if constexpr (VARIANT == 2) {
            *reinterpret_cast<double2*>(p) = make_double2(v, v);
        } else {
            *p = v;
        }
    }
    static constexpr unsigned kMul = VARIANT == 2 ? 2 : 1;
    __device__ __forceinline__ void f_base(int row, double v) const { put(fb + row * fe * kMul, v); }
    __device__ __forceinline__ void f_leg(int rowBase, double v) const { put(fLeg + rowBase * fe * kMul, v); }
    __device__ __forceinline__ void j_leg(int rowBase, int colBase, int legMul, int rot, int, int, int, int, double v) const {
        put((legMul ? jLegCol[rot] : jLeg) + static_cast<unsigned>(rowBase * 49 + colBase) * je * kMul, v);
    }
    __device__ __forceinline__ void j_base_own(int row, int colBase, int, int, int, int, int, int, double v) const {
        put(jOwnCol + static_cast<unsigned>(row * 49 + colBase) * je * kMul, v);
    }
    __device__ __forceinline__ void j_base_shared(int row, int colBase, int, int, int, int, int, int, double v) const {
        put(jb + static_cast<unsigned>(row * 49 + colBase) * je * kMul, v);
    }
    __device__ __forceinline__ void j_leg2(int row, int row2, int colBase, int legMul, int rot, int, int, int, int, int, int, int, int, double v, double v2) const {
        j_leg(row, colBase, legMul, rot, 0, 0, 0, 0, v);
        j_leg(row2, colBase, legMul, rot, 0, 0, 0, 0, v2);
    }
    __device__ __forceinline__ void j_base_own2(int row, int row2, int colBase, int, int, int, int, int, int, int, int, int, int, double v, double v2) const {
        j_base_own(row, colBase, 0, 0, 0, 0, 0, 0, v);
        j_base_own(row2, colBase, 0, 0, 0, 0, 0, 0, v2);
    }
    __device__ __forceinline__ void j_base_shared2(int row, int row2, int colBase, int, int, int, int, int, int, int, int, int, int, double v, double v2) const {
        j_base_shared(row, colBase, 0, 0, 0, 0, 0, 0, v);
        j_base_shared(row2, colBase, 0, 0, 0, 0, 0, 0, v2);
    }
};

// REMAP: output ADDRESSES as if the lanes of a leg were contiguous (lane = 16 * leg + node): timing only,
// the values land in the wrong places -- measures what a row-per-leg lane layout would buy the store path
template <int LDS_SLOTS, int LDS_USLOTS, int VARIANT, int REMAP = 0, bool TIMED = false>
__global__ __launch_bounds__(64) __attribute__((flatten)) void DiagKernel(const NodeLaunch a, const double (*ctab)[4], long long wrap, long long* stamps) {
    __shared__ double lds[LDS_SLOTS * 64 + LDS_USLOTS * 16];
    const int Lreal = (threadIdx.x >> 2) & 3;  // lane layout of quad_kernel.hpp
    const int nodeInWave = 4 * (threadIdx.x >> 4) + (threadIdx.x & 3);
    const long long i = static_cast<long long>(blockIdx.x) * 16 + nodeInWave;
    if (i >= a.count) return;
    // REMAP 1: lane = 16 leg + node (a leg per 16-lane row); REMAP 2: lane = 4 node + leg (legs of a node adjacent)
    const int L = REMAP == 1 ? threadIdx.x >> 4 : REMAP == 2 ? threadIdx.x & 3 : Lreal;
    const long long oi = REMAP == 1   ? static_cast<long long>(blockIdx.x) * 16 + (threadIdx.x & 15)
                         : REMAP == 2 ? static_cast<long long>(blockIdx.x) * 16 + (threadIdx.x >> 2)
                                      : i;
    const long long o = wrap ? oi % wrap : oi;
    constexpr long long kMulA = VARIANT == 2 ? 2 : 1;
    // REMAP 3: tiled output (array of [element][16 nodes] tiles): one wavefront's block is contiguous in HBM
    double* const fb = REMAP == 3 ? a.f.base + (o / 16) * (37 * 16) + (o % 16) : a.f.base + o * kMulA;
    double* const jb = REMAP == 3 ? a.jac.base + (o / 16) * (1813 * 16) + (o % 16) : a.jac.base + o * kMulA;
    const long long je = REMAP == 3 ? 16 : a.jac.es;
    const long long fe = REMAP == 3 ? 16 : a.f.es;
    double* const jLeg = jb + 3LL * L * 49 * je * kMulA;
    constexpr long long kMul = VARIANT == 2 ? 2 : 1;
    StoreIO<VARIANT, TIMED> io{{{a.x.base + i, a.u.base + i, a.p.base, fb, jb, a.x.es, a.u.es, fe, static_cast<unsigned>(je), Lreal, jLeg,
                {jLeg + 3LL * L * je * kMul, jLeg + 3LL * ((L + 1) & 3) * je * kMul, jLeg + 3LL * ((L + 2) & 3) * je * kMul, jLeg + 3LL * ((L + 3) & 3) * je * kMul},
                jb + 3LL * L * je * kMul, fb + 3LL * L * fe * kMul, ctab, lds + threadIdx.x, lds + LDS_SLOTS * 64 + nodeInWave},
               (stamps && threadIdx.x == 0 && blockIdx.x % 64 == 0) ? stamps + (blockIdx.x / 64) * 32 : nullptr}};
    if constexpr (TIMED) {
        if (io.ts) io.ts[io.k++] = clock64();
    }
    Q::ValueJacobianQuad<double>(io);
    if constexpr (TIMED) {
        if (io.ts) io.ts[io.k++] = clock64();
    }
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    const long long count = 81920;
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
    double *dx, *du, *dp, *df, *dj;
    CK(hipMalloc(&dx, x.size() * 8));
    CK(hipMalloc(&du, u.size() * 8));
    CK(hipMalloc(&dp, 8));
    CK(hipMalloc(&df, 37 * count * 8));
    CK(hipMalloc(&dj, 1813 * count * 8));
    CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, p.data(), 8, hipMemcpyHostToDevice));
    NodeLaunch a{};
    a.count = count;
    a.knots = 1;
    a.x = {dx, 1, 0, count};
    a.u = {du, 1, 0, count};
    a.p = {dp, 0, 0, 1};
    a.f = {df, 1, 0, count};
    a.jac = {dj, 1, 0, count};
    void* sym = nullptr;
    CK(hipGetSymbolAddress(&sym, HIP_SYMBOL(Q::kLegConstantsDev)));
    const dim3 grid(static_cast<unsigned>((count + 15) / 16)), block(64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&] {
        hipLaunchKernelGGL((QuadNodeKernel<64, Q::kLdsSlots, Q::kLdsUniformSlots, false, BENCH_STREAM, Body, NoSparsePlan, unsigned, BENCH_BUF>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), Body{});
    };
    const int mode = argc > 2 ? std::atoi(argv[2]) : 0;
    if (mode) {
        long long* dts = nullptr;
        const int nw = static_cast<int>(grid.x / 64);
        CK(hipMalloc(&dts, nw * 32 * 8));
        CK(hipMemset(dts, 0, nw * 32 * 8));
        // the WIDE variant needs output buffers of twice the size
        CK(hipFree(dj));
        CK(hipFree(df));
        CK(hipMalloc(&df, 2 * 37 * count * 8));
        CK(hipMalloc(&dj, 2 * 1813 * count * 8));
        a.f.base = df;
        a.jac.base = dj;
        auto diag = [&](long long wrap, long long* st) {
            if (mode == 1) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 0>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 2) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 1>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 3) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 2>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 4) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 0, 1>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 6) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 0, 2>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 7) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 1, 2>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 8) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 0, 3>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
            if (mode == 5) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 1, 1>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), wrap, st);
        };
        for (long long wrap : {0LL, 1024LL}) {
            for (int i = 0; i < 3; ++i) diag(wrap, nullptr);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) diag(wrap, nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("%s diag mode=%d wrap=%lld kernel_ms=%.4f\n", argc > 1 ? argv[1] : "", mode, wrap, ms / 20);
        }
        if (mode == 1) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 0, 0, true>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), 0LL, dts);
        if (mode == 2) hipLaunchKernelGGL((DiagKernel<Q::kLdsSlots, Q::kLdsUniformSlots, 1, 0, true>), grid, block, 0, 0, a, static_cast<const double(*)[4]>(sym), 0LL, dts);
        if (mode > 2) return 0;
        CK(hipDeviceSynchronize());
        std::vector<long long> ts(nw * 32);
        CK(hipMemcpy(ts.data(), dts, ts.size() * 8, hipMemcpyDeviceToHost));
        // mean cycles per phase over the sampled wavefronts
        std::printf("phase cycles (mean over %d waves):", nw);
        for (int k = 0; k + 1 < 32; ++k) {
            double acc = 0;
            int cnt = 0;
            for (int w = 0; w < nw; ++w)
                if (ts[w * 32 + k + 1] && ts[w * 32 + k]) acc += static_cast<double>(ts[w * 32 + k + 1] - ts[w * 32 + k]), ++cnt;
            if (cnt) std::printf(" %d:%.0f", k, acc / cnt);
        }
        std::printf("\n");
        return 0;
    }
    // steady-state clocks first (bench.py does the same): ~0.5 s of untimed launches
    for (int i = 0; i < 1800; ++i) {
        launch();
        if (i % 100 == 99) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    const int reps = 100;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> jh(64);
    CK(hipMemcpy(jh.data(), dj + 500 * count, 64 * 8, hipMemcpyDeviceToHost));
    double chk = 0;
    for (double v : jh) chk += v;
    std::printf("%s kernel_ms=%.4f frac=%.4f checksum=%.12g\n", argc > 1 ? argv[1] : "", ms / reps, count * 15192.0 / (ms / reps * 1e-3) / 8e12, chk);
    return 0;
}
