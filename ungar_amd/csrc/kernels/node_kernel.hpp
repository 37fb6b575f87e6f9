// ungar_amd :: hand-written HIP kernel skeletons for batched shooting-node evaluation (gfx950).
//
// The hot loop of the reference is one generated C function per model, run single-threaded on one
// problem instance per call (include/ungar/autodiff/function.hpp:186-257).  Here a
// launch evaluates `count` independent nodes; the straight-line model bodies come from the tape
// (csrc/gen/<model>_gen.hpp) and are generic over the I/O policy defined in this file, which is
// where the mapping of lanes onto nodes and of operands onto HBM is decided.
//
// Mapping "lane per node" (this file): lane l of a wavefront owns node i = block*BLOCK + l and runs
// the whole tape in registers.  With the unit-fastest layout (element_stride = count) every
// load/store instruction of a wavefront touches 64 consecutive doubles = 512 contiguous bytes, so
// HBM traffic equals the algorithmic bytes; other layouts go through the same code with their
// strides (correct, but only coalesced when nodes are the fastest axis).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>

namespace ungar_amd::kernels {

struct OperandView {
    double* base;
    long long bs, ks, es;  // instance, knot, element strides (doubles)
};

struct NodeLaunch {
    long long count, knots;
    OperandView x, u, w, p, f, jac;
    OperandView hes{};  // Hessian values of scalar (cost) node models; unused by the dynamics models
};

enum : int { kModeValue = 0, kModeSparseJacobian = 1, kModeDenseJacobian = 2, kModeHessian = 3 };

/// Output store of the node kernels.  STREAM = true: non-temporal (the results are written once and not
/// re-read by the kernel).  Measured on MI355X: +5..25 % for unit-fastest operands whose output exceeds
/// the 256 MB last-level cache (ANYmal 0.322 -> 0.302 ms, rc_car 64 -> 82 % of the HBM spec), but 10x
/// SLOWER for instance-major operands (element stride 1: a lane fills a line over several instructions,
/// which only the write-back cache can merge) and slower for outputs that fit the cache.  The launchers
/// pick per call with UseStreamingStores.
template <bool STREAM>
__device__ __forceinline__ void StoreResult(double* p, double v) {
    if constexpr (STREAM) __builtin_nontemporal_store(v, p);
    else *p = v;
}

/// Streaming (non-temporal) stores pay off when every store instruction covers whole cache lines
/// (unit-fastest layout: consecutive nodes are consecutive in memory) and the output does not fit the
/// last-level cache anyway.
inline bool UseStreamingStores(const NodeLaunch& a, int mode, long long jacobianEntries, long long valueEntries) {
    constexpr long long kLastLevelCacheBytes = 256LL << 20;
    const OperandView& out = mode == kModeValue ? a.f : a.jac;
    const bool unitFastest = (a.knots > 1 ? out.ks == 1 || out.bs == 1 : out.bs == 1) && out.es != 1;
    const long long bytes = a.count * 8 * (mode == kModeValue ? valueEntries : jacobianEntries);
    return unitFastest && bytes > kLastLevelCacheBytes;
}

/// I/O policy: every lane addresses its own node through (base, element stride).
template <int NCOLS, bool DENSE, bool STREAM = false>
struct StridedIO {
    const double* __restrict__ xb;
    const double* __restrict__ ub;
    const double* __restrict__ wb;
    const double* __restrict__ pb;
    double* __restrict__ fb;
    double* __restrict__ jb;
    long long xe, ue, we, pe, fe, je;

    __device__ __forceinline__ double x(int i) const { return xb[i * xe]; }
    __device__ __forceinline__ double u(int i) const { return ub[i * ue]; }
    __device__ __forceinline__ double w(int i) const { return wb[i * we]; }
    __device__ __forceinline__ double p(int i) const { return pb[i * pe]; }
    __device__ __forceinline__ void f(int i, double v) const {
        if (fb) StoreResult<STREAM>(fb + i * fe, v);
    }
    __device__ __forceinline__ void j(int k, int r, int c, double v) const {
        StoreResult<STREAM>(jb + (DENSE ? r * NCOLS + c : k) * je, v);
    }
};

namespace detail {

/// Dense offset (r * cols + c) of the z-th structural ZERO of the model's Jacobian pattern.
template <class M>
constexpr int ZeroOffset(int z) {
    int seen = 0, k = 0;
    for (int e = 0; e < M::kJacRows * M::kJacCols; ++e) {
        const int r = e / M::kJacCols, c = e % M::kJacCols;
        // pattern is sorted row-major: advance k to the first entry >= (r, c)
        while (k < M::kJacNnz && (M::JacRow(k) < r || (M::JacRow(k) == r && M::JacCol(k) < c))) ++k;
        const bool nz = k < M::kJacNnz && M::JacRow(k) == r && M::JacCol(k) == c;
        if (!nz) {
            if (seen == z) return e;
            ++seen;
        }
    }
    return -1;
}

template <class M, bool STREAM, std::size_t... Z>
__device__ __forceinline__ void StoreZeros(double* __restrict__ jb, long long je, std::index_sequence<Z...>) {
    // offsets are compile-time constants; one store per structural zero
    (StoreResult<STREAM>(jb + static_cast<long long>(std::integral_constant<int, ZeroOffset<M>(static_cast<int>(Z))>::value) * je, 0.0), ...);
}

/// All structural zeros of the pattern in one pass (for blocks with thousands of them, e.g. d M(q) / d q: 5240 of 6156).
template <class M>
struct ZeroTable {
    static constexpr int kCount = M::kJacRows * M::kJacCols - M::kJacNnz;
    int offset[kCount > 0 ? kCount : 1];
    constexpr ZeroTable() : offset{} {
        int z = 0, k = 0;
        for (int e = 0; e < M::kJacRows * M::kJacCols; ++e) {
            const int r = e / M::kJacCols, c = e % M::kJacCols;
            while (k < M::kJacNnz && (M::JacRow(k) < r || (M::JacRow(k) == r && M::JacCol(k) < c))) ++k;
            if (!(k < M::kJacNnz && M::JacRow(k) == r && M::JacCol(k) == c)) offset[z++] = e;
        }
    }
};
template <class M>
__device__ __constant__ const ZeroTable<M> kZeroTable{};

template <class M, bool STREAM>
__device__ __forceinline__ void StoreZerosFromTable(double* __restrict__ jb, long long je) {
    for (int z = 0; z < ZeroTable<M>::kCount; ++z) StoreResult<STREAM>(jb + static_cast<long long>(kZeroTable<M>.offset[z]) * je, 0.0);
}

inline constexpr int kMaxUnrolledZeros = 1024;  // beyond this the fold expression exceeds the compiler's nesting limit

}  // namespace detail

/// One lane per shooting node.
template <class M, int MODE, int BLOCK, bool STREAM>
__global__ __launch_bounds__(BLOCK) void NodeKernel(const NodeLaunch a) {
    const long long i = static_cast<long long>(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= a.count) return;
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    StridedIO<M::kJacCols, MODE == kModeDenseJacobian, STREAM> io{
        a.x.base + b * a.x.bs + k * a.x.ks,
        a.u.base + b * a.u.bs + k * a.u.ks,
        a.w.base ? a.w.base + b * a.w.bs + k * a.w.ks : nullptr,
        a.p.base + b * a.p.bs + k * a.p.ks,
        a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr,
        MODE == kModeValue ? nullptr : a.jac.base + b * a.jac.bs + k * a.jac.ks,
        a.x.es, a.u.es, a.w.es, a.p.es, a.f.es, a.jac.es};
    if constexpr (MODE == kModeValue) {
        M::Value(io);
    } else {
        if constexpr (MODE == kModeDenseJacobian) {
            if constexpr (M::kJacRows * M::kJacCols - M::kJacNnz <= detail::kMaxUnrolledZeros)
                detail::StoreZeros<M, STREAM>(io.jb, io.je, std::make_index_sequence<M::kJacRows * M::kJacCols - M::kJacNnz>{});
            else
                detail::StoreZerosFromTable<M, STREAM>(io.jb, io.je);
        }
        M::ValueJacobian(io);
    }
}

/// I/O policy for INSTANCE-MAJOR dense Jacobians (element stride 1: what a VariableMap-style buffer per
/// instance looks like).  A lane storing its own node's entries one by one touches a different cache line
/// in every lane of the wavefront (8 bytes per line per instruction: 3-4x slower than the unit-fastest
/// layout).  Here every wavefront collects one Jacobian ROW of its 64 nodes in LDS and writes it out
/// cooperatively: consecutive lanes store consecutive doubles of the node-major array, i.e. runs of NCOLS
/// contiguous doubles per node.  The generated body emits the non-zeros row by row with literal (row, col)
/// arguments, so `row != current` folds at compile time; rows are zero-filled in LDS, which also replaces the
/// separate structural-zero stores.  No workgroup barrier: the buffer is per wavefront and LDS is in-order.
template <int NROWS, int NCOLS, int NX, int NU>
struct RowBufferedIO {
    static constexpr int kPitch = NCOLS | 1;  // odd pitches: lane-strided LDS accesses hit distinct banks
    static constexpr int kPitchX = NX | 1, kPitchU = NU | 1, kPitchF = NROWS | 1;
    static constexpr int kDoublesPerWave = 64 * (kPitch + kPitchX + kPitchU + kPitchF);
    const double* __restrict__ wb;
    const double* __restrict__ pb;
    long long we, pe;
    double* rowbuf;  // this wavefront's [64][kPitch] Jacobian row
    double* xl;      // [64][kPitchX] staged states
    double* ul;      // [64][kPitchU] staged inputs
    double* fl;      // [64][kPitchF] values, written out at the end
    double* __restrict__ jw;  // Jacobian / value of the wavefront's first node
    double* __restrict__ fw;
    long long jacNodeStride, fNodeStride;  // doubles between consecutive nodes
    int lane, nodesInWave;                 // nodes of this wavefront that exist (tail)
    int current = 0;

    __device__ __forceinline__ double x(int i) const { return xl[lane * kPitchX + i]; }
    __device__ __forceinline__ double u(int i) const { return ul[lane * kPitchU + i]; }
    __device__ __forceinline__ double w(int i) const { return wb[i * we]; }
    __device__ __forceinline__ double p(int i) const { return pb[i * pe]; }
    __device__ __forceinline__ void f(int i, double v) const { fl[lane * kPitchF + i] = v; }

    /// Cooperative copy between a node-major global array (n doubles per node, `stride` apart) and an LDS
    /// tile [64][pitch]: lane l takes flat index q = t * 64 + l over [node][element].
    template <int N, bool TO_LDS>
    __device__ __forceinline__ void copy(double* tile, int pitch, double* __restrict__ global, long long stride) const {
        for (int q = lane; q < nodesInWave * N; q += 64) {
            const int node = q / N, e = q - node * N;
            if constexpr (TO_LDS) tile[node * pitch + e] = global[node * stride + e];
            else global[node * stride + e] = tile[node * pitch + e];
        }
    }
    __device__ __forceinline__ void clear() const {
#pragma unroll
        for (int c = 0; c < NCOLS; ++c) rowbuf[lane * kPitch + c] = 0.0;
    }
    __device__ __forceinline__ void flush(int row) const {
        copy<NCOLS, false>(rowbuf, kPitch, jw + row * NCOLS, jacNodeStride);
    }
    __device__ __forceinline__ void j(int /*k*/, int r, int c, double v) {
        while (current < r) {  // literal rows: unrolled and folded at compile time
            flush(current);
            clear();
            ++current;
        }
        rowbuf[lane * kPitch + c] = v;
    }
    __device__ __forceinline__ void finish() {
        while (current < NROWS) {
            flush(current);
            if (current + 1 < NROWS) clear();
            ++current;
        }
        if (fw) copy<NROWS, false>(fl, kPitchF, fw, fNodeStride);
    }
};

/// Dense Jacobians for node-major (element stride 1) x, u, f and jac operands whose nodes are equally spaced
/// (knots == 1, or contiguous trajectories): states and inputs are staged, values and Jacobian rows written
/// out, through LDS with consecutive lanes on consecutive doubles (RowBufferedIO).
template <class M, int BLOCK>
__global__ __launch_bounds__(BLOCK) void NodeKernelAosDense(const NodeLaunch a, long long xStride, long long uStride, long long fStride, long long jacStride) {
    using IO = RowBufferedIO<M::kJacRows, M::kJacCols, M::kNx, M::kNu>;
    __shared__ double lds[(BLOCK / 64) * IO::kDoublesPerWave];
    const long long first = static_cast<long long>(blockIdx.x) * BLOCK + (threadIdx.x & ~63);  // first node of the wavefront
    if (first >= a.count) return;  // whole wavefront
    const long long i = min(static_cast<long long>(blockIdx.x) * BLOCK + threadIdx.x, a.count - 1);  // tail lanes recompute the last node
    long long b = i, k = 0, fb = first, fk = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
        fb = first / a.knots;
        fk = first - fb * a.knots;
    }
    double* const tile = lds + (threadIdx.x >> 6) * IO::kDoublesPerWave;
    IO io{a.w.base ? a.w.base + b * a.w.bs + k * a.w.ks : nullptr,
          a.p.base + b * a.p.bs + k * a.p.ks,
          a.w.es,
          a.p.es,
          tile,
          tile + 64 * IO::kPitch,
          tile + 64 * (IO::kPitch + IO::kPitchX),
          tile + 64 * (IO::kPitch + IO::kPitchX + IO::kPitchU),
          a.jac.base + fb * a.jac.bs + fk * a.jac.ks,
          a.f.base ? a.f.base + fb * a.f.bs + fk * a.f.ks : nullptr,
          jacStride,
          fStride,
          static_cast<int>(threadIdx.x & 63),
          static_cast<int>(min(64LL, a.count - first))};
    io.template copy<M::kNx, true>(io.xl, IO::kPitchX, const_cast<double*>(a.x.base + fb * a.x.bs + fk * a.x.ks), xStride);
    if constexpr (M::kNu > 0) io.template copy<M::kNu, true>(io.ul, IO::kPitchU, const_cast<double*>(a.u.base + fb * a.u.bs + fk * a.u.ks), uStride);
    // tail lanes (no node of their own) read the staged state of the last existing node
    if (io.lane >= io.nodesInWave) {
        for (int e = 0; e < M::kNx; ++e) io.xl[io.lane * IO::kPitchX + e] = io.xl[(io.nodesInWave - 1) * IO::kPitchX + e];
        for (int e = 0; e < M::kNu; ++e) io.ul[io.lane * IO::kPitchU + e] = io.ul[(io.nodesInWave - 1) * IO::kPitchU + e];
    }
    io.clear();
    M::ValueJacobian(io);
    io.finish();
}

/// I/O policy of the phased bodies: StridedIO plus a per-lane LDS home for values that live across
/// phases (slot s of lane l at lds[s * BLOCK + l]: a wavefront access is 64 consecutive doubles,
/// conflict-free for ds_read_b64 / ds_write_b64) and a scheduling barrier between phases so that
/// the compiler cannot stretch live ranges across them.
template <int NCOLS, bool DENSE, int BLOCK>
struct PhasedIO : StridedIO<NCOLS, DENSE> {
    double* lds;  // already offset by the lane
    __device__ __forceinline__ double ld(int slot) const { return lds[slot * BLOCK]; }
    __device__ __forceinline__ void st(int slot, double v) const { lds[slot * BLOCK] = v; }
    __device__ __forceinline__ void phase() const { __builtin_amdgcn_sched_barrier(0); }
};

/// One lane per shooting node, long-lived state in LDS (one workgroup of BLOCK lanes owns
/// kLdsSlots * BLOCK doubles; with 320 slots and BLOCK = 64 that is the whole 160 KiB of a CU).
template <class M, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void NodeKernelPhased(const NodeLaunch a) {
    __shared__ double lds[(M::kLdsSlots > 0 ? M::kLdsSlots : 1) * BLOCK];
    const long long i = static_cast<long long>(blockIdx.x) * BLOCK + threadIdx.x;
    if (i >= a.count) return;
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    PhasedIO<M::kJacCols, MODE == kModeDenseJacobian, BLOCK> io{
        {a.x.base + b * a.x.bs + k * a.x.ks, a.u.base + b * a.u.bs + k * a.u.ks, a.w.base ? a.w.base + b * a.w.bs + k * a.w.ks : nullptr,
         a.p.base + b * a.p.bs + k * a.p.ks, a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr,
         a.jac.base + b * a.jac.bs + k * a.jac.ks, a.x.es, a.u.es, a.w.es, a.p.es, a.f.es, a.jac.es},
        lds + threadIdx.x};
    if constexpr (MODE == kModeDenseJacobian) {
        if constexpr (M::kJacRows * M::kJacCols - M::kJacNnz <= detail::kMaxUnrolledZeros)
            detail::StoreZeros<M, false>(io.jb, io.je, std::make_index_sequence<M::kJacRows * M::kJacCols - M::kJacNnz>{});
        else
            detail::StoreZerosFromTable<M, false>(io.jb, io.je);
    }
    M::ValueJacobianPhased(io);
}

/// Node-major operand (element stride 1) whose nodes are equally spaced in memory.
inline bool NodeMajorEquallySpaced(const NodeLaunch& a, const OperandView& o) {
    return o.base && o.es == 1 && (a.knots == 1 || o.bs == a.knots * o.ks);
}
inline long long NodeStride(const NodeLaunch& a, const OperandView& o) {
    return a.knots == 1 ? o.bs : o.ks;
}

/// The row-buffered node-major kernel needs a row-major body, equally spaced node-major x, u, f, jac and
/// its LDS tiles to fit a workgroup's static allocation (the taped-ABA ANYmal comparison kernel does not).
template <class M, int BLOCK>
inline constexpr bool kAosDenseFits =
    M::kRowMajorEmission && M::kLdsSlots == 0 &&
    sizeof(double) * (BLOCK / 64) * RowBufferedIO<M::kJacRows, M::kJacCols, M::kNx, M::kNu>::kDoublesPerWave <= 64 * 1024;
template <class M, int BLOCK>
inline bool AosDenseApplies(const NodeLaunch& a) {
    if constexpr (!kAosDenseFits<M, BLOCK>) return false;
    else
        return NodeMajorEquallySpaced(a, a.jac) && NodeMajorEquallySpaced(a, a.x) && (M::kNu == 0 || NodeMajorEquallySpaced(a, a.u)) &&
               (!a.f.base || NodeMajorEquallySpaced(a, a.f));
}
template <class M, int BLOCK>
inline void LaunchAosDense(const NodeLaunch& a, dim3 grid, dim3 block, hipStream_t stream) {
    if constexpr (kAosDenseFits<M, BLOCK>)
        hipLaunchKernelGGL((NodeKernelAosDense<M, BLOCK>), grid, block, 0, stream, a, NodeStride(a, a.x), NodeStride(a, a.u), NodeStride(a, a.f), NodeStride(a, a.jac));
}

template <class M, int BLOCK>
inline hipError_t LaunchNodeModel(int mode, const NodeLaunch& a, hipStream_t stream) {
    if (a.count <= 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((a.count + BLOCK - 1) / BLOCK)), block(BLOCK);
    const bool streaming =
        UseStreamingStores(a, mode, mode == kModeDenseJacobian ? static_cast<long long>(M::kJacRows) * M::kJacCols : M::kJacNnz, M::kJacRows);
    switch (mode) {
        case kModeValue:
            if (streaming) hipLaunchKernelGGL((NodeKernel<M, kModeValue, BLOCK, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((NodeKernel<M, kModeValue, BLOCK, false>), grid, block, 0, stream, a);
            break;
        case kModeSparseJacobian:
            if constexpr (M::kLdsSlots > 0) hipLaunchKernelGGL((NodeKernelPhased<M, kModeSparseJacobian, BLOCK>), grid, block, 0, stream, a);
            else if (streaming) hipLaunchKernelGGL((NodeKernel<M, kModeSparseJacobian, BLOCK, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((NodeKernel<M, kModeSparseJacobian, BLOCK, false>), grid, block, 0, stream, a);
            break;
        case kModeDenseJacobian:
            if constexpr (M::kLdsSlots > 0) hipLaunchKernelGGL((NodeKernelPhased<M, kModeDenseJacobian, BLOCK>), grid, block, 0, stream, a);
            else if (AosDenseApplies<M, BLOCK>(a)) LaunchAosDense<M, BLOCK>(a, grid, block, stream);
            else if (streaming) hipLaunchKernelGGL((NodeKernel<M, kModeDenseJacobian, BLOCK, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((NodeKernel<M, kModeDenseJacobian, BLOCK, false>), grid, block, 0, stream, a);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ungar_amd::kernels

/// Binds a generated model namespace to the traits the skeletons expect (no launcher).
#define UNGAR_AMD_DEFINE_NODE_TRAITS(ns)                                                                         \
    namespace ungar_amd::kernels {                                                                               \
    struct Model_##ns {                                                                                          \
        static constexpr int kNx = gen::ns::kNx, kNu = gen::ns::kNu, kNw = gen::ns::kNw, kNp = gen::ns::kNp;     \
        static constexpr int kJacRows = gen::ns::kJacRows, kJacCols = gen::ns::kJacCols, kJacNnz = gen::ns::kJacNnz; \
        static constexpr int kLdsSlots = gen::ns::kLdsSlots;                                                     \
        static constexpr bool kRowMajorEmission = gen::ns::kJacMode < 10; /* structured bodies emit column by column */ \
        static constexpr int JacRow(int k) { return gen::ns::kJacRow[k]; }                                       \
        static constexpr int JacCol(int k) { return gen::ns::kJacCol[k]; }                                       \
        template <class IO>                                                                                      \
        __device__ __forceinline__ static void Value(IO& io) { gen::ns::Value(io); }                             \
        template <class IO>                                                                                      \
        __device__ __forceinline__ static void ValueJacobian(IO& io) { gen::ns::ValueJacobian(io); }             \
        template <class IO>                                                                                      \
        __device__ __forceinline__ static void ValueJacobianPhased(IO& io) {                                     \
            if constexpr (gen::ns::kLdsSlots > 0) gen::ns::ValueJacobianPhased(io);                              \
        }                                                                                                        \
    };                                                                                                           \
    }                                                                                                            \
    extern "C" const int* ungar_amd_pattern_##ns(int which, int* nnz) {                                          \
        *nnz = ungar_amd::gen::ns::kJacNnz;                                                                      \
        return which == 0 ? ungar_amd::gen::ns::kJacRow : ungar_amd::gen::ns::kJacCol;                           \
    }                                                                                                            \
    extern "C" void ungar_amd_dims_##ns(int* d) {                                                                \
        d[0] = ungar_amd::gen::ns::kNx;                                                                          \
        d[1] = ungar_amd::gen::ns::kNu;                                                                          \
        d[2] = ungar_amd::gen::ns::kNw;                                                                          \
        d[3] = ungar_amd::gen::ns::kNp;                                                                          \
        d[4] = ungar_amd::gen::ns::kJacRows; /* outputs per node */                                              \
    }

/// Binds a generated model namespace to the traits the skeletons expect and defines its launcher.
#define UNGAR_AMD_DEFINE_NODE_MODEL(ns, BLOCK)                                                                   \
    namespace ungar_amd::kernels {                                                                               \
    struct Model_##ns {                                                                                          \
        static constexpr int kNx = gen::ns::kNx, kNu = gen::ns::kNu, kNw = gen::ns::kNw, kNp = gen::ns::kNp;     \
        static constexpr int kJacRows = gen::ns::kJacRows, kJacCols = gen::ns::kJacCols, kJacNnz = gen::ns::kJacNnz; \
        static constexpr int kLdsSlots = gen::ns::kLdsSlots;                                                     \
        static constexpr bool kRowMajorEmission = gen::ns::kJacMode < 10; /* structured bodies emit column by column */ \
        static constexpr int JacRow(int k) { return gen::ns::kJacRow[k]; }                                       \
        static constexpr int JacCol(int k) { return gen::ns::kJacCol[k]; }                                       \
        template <class IO>                                                                                      \
        __device__ __forceinline__ static void Value(IO& io) { gen::ns::Value(io); }                             \
        template <class IO>                                                                                      \
        __device__ __forceinline__ static void ValueJacobian(IO& io) { gen::ns::ValueJacobian(io); }             \
        template <class IO>                                                                                      \
        __device__ __forceinline__ static void ValueJacobianPhased(IO& io) {                                     \
            if constexpr (gen::ns::kLdsSlots > 0) gen::ns::ValueJacobianPhased(io);                              \
        }                                                                                                        \
    };                                                                                                           \
    }                                                                                                            \
    extern "C" int ungar_amd_launch_##ns(int mode, const ungar_amd::kernels::NodeLaunch* a, void* stream) {      \
        return static_cast<int>(ungar_amd::kernels::LaunchNodeModel<ungar_amd::kernels::Model_##ns, BLOCK>(      \
            mode, *a, static_cast<hipStream_t>(stream)));                                                        \
    }                                                                                                            \
    extern "C" const int* ungar_amd_pattern_##ns(int which, int* nnz) {                                          \
        *nnz = ungar_amd::gen::ns::kJacNnz;                                                                      \
        return which == 0 ? ungar_amd::gen::ns::kJacRow : ungar_amd::gen::ns::kJacCol;                           \
    }                                                                                                            \
    extern "C" void ungar_amd_dims_##ns(int* d) {                                                                \
        d[0] = ungar_amd::gen::ns::kNx;                                                                          \
        d[1] = ungar_amd::gen::ns::kNu;                                                                          \
        d[2] = ungar_amd::gen::ns::kNw;                                                                          \
        d[3] = ungar_amd::gen::ns::kNp;                                                                          \
        d[4] = ungar_amd::gen::ns::kJacRows; /* outputs per node */                                              \
    }
