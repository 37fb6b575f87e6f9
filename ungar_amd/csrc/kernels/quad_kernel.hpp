// ungar_amd :: kernel skeleton of the lane-per-leg SPMD node program.
//
// Four lanes own one shooting node of a floating-base quadruped, one lane per leg.  A 64-lane wavefront
// therefore evaluates 16 nodes; four wavefronts run per CU (one per SIMD) -- against ONE for the
// lane-per-node body whose state had to be parked in 160 KiB of LDS.
// Lane layout inside each 16-lane DPP row:  lane = 4 * leg + (node % 4),  row = node / 4.
//   * the four legs of a node sit 4 lanes apart in one row, so they meet through DPP row rotations
//     (v_mov_b32 ... row_ror:4/8/12: register-to-register, no LDS);
//   * 4 ADJACENT lanes hold the same leg of 4 consecutive nodes, so in the unit-fastest layout every
//     store instruction writes 32-byte contiguous runs (4 per 128-byte line).  With the legs of a node
//     in adjacent lanes instead (quad_perm), adjacent lanes hit four different lines and the kernel
//     was bound by the address path: 0.416 ms vs 0.335 ms measured with nothing else changed.
#pragma once

#include <hip/hip_runtime.h>

#include "node_kernel.hpp"

namespace ungar_amd::kernels {

template <int CTRL>
__device__ __forceinline__ double QuadPerm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

/// I/O policy of gen::anymal_quad::ValueJacobianQuad<double>: node-level operands through strides,
/// leg-dependent rows/columns through per-lane offsets computed once.  Every Jacobian sink of the
/// generated program carries both addressings -- (row, column) of the dense block and, per lane, the
/// index k of the entry in the CSR value array (-1 = structural zero); SPARSE picks the second.
/// Default sparse-addressing plan: no precomputed patterns (every sparse store selects its index per lane).
struct NoSparsePlan {
    static constexpr int kCount = 0;
    static constexpr int kDeltas[1][4] = {{0, 0, 0, 0}};
};

/// PLAN (generated: gen::anymal_quad::SparsePlan) lists the distinct per-leg index patterns k_L - k_0 of the
/// sparse sinks; the kernel keeps one per-lane base pointer per pattern (jS[p] = jb + kDeltas[p][leg] * je),
/// so a sparse store is (lane pointer) + (wave-uniform offset k_0 * je) like a dense one.
/// OFF is the type of the wave-uniform element offsets: 32-bit when 1813 * (element stride) < 2^32 (checked by the
/// launcher), 64-bit for larger unit-fastest operands (> 2.37 M nodes per launch: one more scalar multiply per store).
///
/// BUF (dense output only): the Jacobian is stored with MUBUF instructions, buffer_store_dwordx2 v_data, v_lane_offset,
/// s[resource], s_entry_offset -- the lane-dependent part of the address is a 32-bit VGPR byte offset computed once, the
/// entry-dependent part a scalar register, so a store costs one s_mul_i32 and NO vector-ALU address arithmetic (the
/// pointer form needs a 64-bit v_lshl_add_u64 per store: 635 of the 14.2 k instructions a lane issues).  Requires the whole
/// operand to span less than 4 GiB from the first node of the launch (checked by the launcher: 1813 * stride * 8 < 2^32).
///
/// PAIR (with BUF): two entries of a column leave in ONE 16-byte store.  The lanes of the partner nodes n, n + 1 (same lane of two
/// neighbouring 16-lane rows: same leg) swap one value each; the lane of the even node then writes entry e of both nodes, the lane of the
/// odd node entry e' = e + delta of both nodes (delta = 18 rows for the leg rows 7 + k / 25 + k, one row for the base rows): half the
/// store instructions, and -- what matters -- 16-byte stores reach 6.1-6.4 TB/s where the 8-byte ones stop at 5.0-5.3
/// (tools/store_ceiling.hip, profiles/r05a_store_ceiling.log).  Needs consecutive nodes at consecutive addresses and an even node
/// count (checked by the launcher).
template <bool SPARSE, bool STREAM = false, class PLAN = NoSparsePlan, class OFF = unsigned, bool BUF = false, bool PAIR = false>
struct QuadIO {
    static_assert(!PAIR || (BUF && !SPARSE), "paired stores are a variant of the buffer-store path of the dense block");
    const double* __restrict__ xb;  // node's x (element stride xe)
    const double* __restrict__ ub;
    const double* __restrict__ pb;
    double* __restrict__ fb;
    double* __restrict__ jb;
    long long xe, ue, fe;
    OFF je;  // element stride of the Jacobian operand (OFF = unsigned: 1813 * je < 2^32 is checked by the launcher)
    int L;                       // this lane's leg
    // per-lane base pointers, so that every store address is (lane pointer) + (wave-uniform offset):
    double* __restrict__ jLeg;        // jb + 3 L * 49 * je            : this leg's row block, leg-independent column
    double* __restrict__ jLegCol[4];  // jLeg + 3 ((L + rot) & 3) * je : ... column owned by the leg `rot` lanes away
    double* __restrict__ jOwnCol;     // jb + 3 L * je                  : base rows, column owned by this leg
    double* __restrict__ fLeg;        // fb + 3 L * fe
    const double (*ctab)[4];
    double* jS[PLAN::kCount > 0 ? PLAN::kCount : 1];  // sparse mode: per-lane base pointer of every index pattern
    double* lds;   // per-lane LDS home of the phased body (slot s at lds[s * 64])
    double* ldsu;  // per-quad home of lane-uniform values (slot s at ldsu[s * 16]): a quarter of the bytes
    // BUF: buffer resource over the Jacobian operand and the per-lane byte offsets that replace the pointers above
    struct BufferState {
#if defined(__HIP_DEVICE_COMPILE__)
        __amdgpu_buffer_rsrc_t jr;
#endif
        int vNode, vLeg, vLegCol[4], vOwnCol;
        int pNode, pLeg, pLegCol[4], pOwnCol;  // PAIR: offsets of the 16-byte stores (odd lanes: the second entry of the pair, one node back)
        unsigned je8;  // element stride in bytes (wave-uniform)
        __host__ __device__ BufferState() {}  // filled in by the kernel when BUF
    } buf;

    __device__ __forceinline__ double qb(int i) const { return xb[i * xe]; }
    __device__ __forceinline__ double vb(int i) const { return xb[(19 + i) * xe]; }
    __device__ __forceinline__ double ql(int i) const { return xb[(7 + 3 * L + i) * xe]; }
    __device__ __forceinline__ double vl(int i) const { return xb[(25 + 3 * L + i) * xe]; }
    __device__ __forceinline__ double ul(int i) const { return ub[(3 * L + i) * ue]; }
    __device__ __forceinline__ double dt() const { return pb[0]; }
    __device__ __forceinline__ double c(int k) const { return ctab[k][L]; }
#ifndef UNGAR_QUAD_NO_PHASE_BARRIER
    __device__ __forceinline__ void phase() const { __builtin_amdgcn_sched_barrier(0); }
#else
    __device__ __forceinline__ void phase() const {}
#endif
    __device__ __forceinline__ void keep(double) const {}
    /// a * b + c with one rounding: the generator decides which products are contracted (tape::FuseMultiplyAdd), the program is compiled with contraction off
    __device__ __forceinline__ double fma(double a, double b, double c) const { return __builtin_fma(a, b, c); }
    __device__ __forceinline__ double ld(int slot) const { return lds[slot * 64]; }
    __device__ __forceinline__ void st(int slot, double v) const { lds[slot * 64] = v; }
    __device__ __forceinline__ double ldu(int slot) const { return ldsu[slot * 16]; }  // same address in the 4 lanes: broadcast
    __device__ __forceinline__ void stu(int slot, double v) const { ldsu[slot * 16] = v; }  // 4 lanes, same address, same value

    // DPP row_ror:n (ctrl 0x120 + n): lane i of a 16-lane row reads lane (i - n) mod 16.  Legs are 4 lanes
    // apart, so leg L reads leg (L + k) & 3 of its own node with n = 16 - 4 k.
    __device__ __forceinline__ double quad_sum(double v) const {
        const double t = v + QuadPerm<0x124>(v);
        return t + QuadPerm<0x128>(t);
    }
    __device__ __forceinline__ double quad_rot1(double v) const { return QuadPerm<0x12C>(v); }  // from leg L + 1
    __device__ __forceinline__ double quad_rot2(double v) const { return QuadPerm<0x128>(v); }  // from leg L + 2
    __device__ __forceinline__ double quad_rot3(double v) const { return QuadPerm<0x124>(v); }  // from leg L + 3

    /// Output store; STREAM = non-temporal (node_kernel.hpp: StoreResult / UseStreamingStores).
    static __device__ __forceinline__ void Put(double* p, double v) {
        StoreResult<STREAM>(p, v);
    }
    /// Cache policy of the buffer stores (aux operand: 1 = sc0, 2 = nt, 16 = sc1): non-temporal for streaming outputs.
#if defined(UNGAR_AMD_MEASUREMENT_STORE_AUX)  // tools/quad_split_bench.hip: sweep of the cache-policy bits
    static constexpr int kStoreAux = UNGAR_AMD_MEASUREMENT_STORE_AUX;
#else
    static constexpr int kStoreAux = STREAM ? 2 : 0;
#endif
    /// Buffer store of entry `e` (wave-uniform) at the lane offset `voff`.
    __device__ __forceinline__ void BufPut(int voff, unsigned e, double v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef int v2i __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), buf.jr, voff, static_cast<int>(e * buf.je8), kStoreAux);
#else
        (void)voff, (void)e, (void)v;
#endif
    }
    /// PAIR: entries e (value v) and e + delta (value v2) of this lane's node and of its partner's, one 16-byte store per lane.
    /// v_permlane16_swap_b32 (gfx950) swaps the odd 16-lane rows of its first operand with the even rows of its second: with the
    /// partner nodes n, n + 1 in the same lane of rows 2 R / 2 R + 1 (QuadNodeInWave<true>), the even row ends up with (v(n), v(n + 1))
    /// and the odd row with (v2(n), v2(n + 1)) -- two instructions for the exchange of both 64-bit values, no select, no LDS.
    __device__ __forceinline__ void BufPut2(int poff, unsigned e, double v, double v2) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef int v4i __attribute__((ext_vector_type(4)));
        const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(v)), static_cast<unsigned>(__double2loint(v2)), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(v)), static_cast<unsigned>(__double2hiint(v2)), false, false);
        const v4i q{static_cast<int>(lo[0]), static_cast<int>(hi[0]), static_cast<int>(lo[1]), static_cast<int>(hi[1])};
        __builtin_amdgcn_raw_buffer_store_b128(q, buf.jr, poff, static_cast<int>(e * buf.je8), kStoreAux);
        // A store of more than 64 bits fetches its data registers AFTER it has issued; a vector instruction that overwrites one of them in
        // the next cycles corrupts what is stored.  The compiler pads that hazard only for stores WITHOUT a scalar offset register
        // (GCNHazardRecognizer: "this hazard only exists if the instruction is not using a register in the soffset field"), which does not
        // hold on gfx950: with two wavefronts per SIMD the low dword of ~1e-5 of the stored values came out overwritten
        // (profiles/r05a_store_data_hazard.log).  The statement below keeps the four registers alive across two wait states.
        asm volatile("s_nop 1" ::"v"(q) : "memory");
#else
        (void)poff, (void)e, (void)v, (void)v2;
#endif
    }
    // base rows / shared columns: all four lanes hold the same value and store it to the same address
    // (merged inside the instruction) -- cheaper than masking three lanes off with exec-mask branches
    __device__ __forceinline__ void f_base(int row, double v) const {
        if (fb) Put(fb + row * fe, v);
    }
    __device__ __forceinline__ void f_leg(int rowBase, double v) const {
        if (fb) Put(fLeg + rowBase * fe, v);
    }
    __device__ __forceinline__ void j_sparse(int k0, int k1, int k2, int k3, double v) const {
        if (k0 < 0 && k1 < 0 && k2 < 0 && k3 < 0) return;  // literal arguments: folds away at compile time
        if (k0 >= 0 && k1 >= 0 && k2 >= 0 && k3 >= 0) {
#pragma unroll
            for (int p = 0; p < PLAN::kCount; ++p)  // literal arguments, constexpr table: exactly one branch survives
                if (k1 - k0 == PLAN::kDeltas[p][1] && k2 - k0 == PLAN::kDeltas[p][2] && k3 - k0 == PLAN::kDeltas[p][3]) {
                    Put(jS[p] + static_cast<unsigned>(k0) * je, v);
                    return;
                }
        }
        const int k = L == 0 ? k0 : L == 1 ? k1 : L == 2 ? k2 : k3;
        if (k0 >= 0 && k1 >= 0 && k2 >= 0 && k3 >= 0)
            Put(jb + static_cast<unsigned>(k) * je, v);
        else if (k >= 0)
            Put(jb + static_cast<unsigned>(k) * je, v);
    }
    __device__ __forceinline__ void j_leg(int rowBase, int colBase, int legMul, int rot, int k0, int k1, int k2, int k3, double v) const {
        if constexpr (SPARSE)
            j_sparse(k0, k1, k2, k3, v);
        else if constexpr (BUF)
            BufPut(legMul ? buf.vLegCol[rot] : buf.vLeg, static_cast<unsigned>(rowBase * 49 + colBase), v);
        else
            Put((legMul ? jLegCol[rot] : jLeg) + static_cast<unsigned>(rowBase * 49 + colBase) * je, v);
    }
    __device__ __forceinline__ void j_base_own(int row, int colBase, int /*legMul*/, int /*rot*/, int k0, int k1, int k2, int k3, double v) const {
        if constexpr (SPARSE)
            j_sparse(k0, k1, k2, k3, v);
        else if constexpr (BUF)
            BufPut(buf.vOwnCol, static_cast<unsigned>(row * 49 + colBase), v);
        else
            Put(jOwnCol + static_cast<unsigned>(row * 49 + colBase) * je, v);
    }
    // paired sinks (quad_leg_program.hpp: pairStores): one 16-byte store with PAIR, otherwise the two entries one by one
    __device__ __forceinline__ void j_leg2(int row, int row2, int colBase, int legMul, int rot, int k0, int k1, int k2, int k3, int m0, int m1, int m2, int m3, double v,
                                           double v2) const {
        if constexpr (PAIR) {
            BufPut2(legMul ? buf.pLegCol[rot] : buf.pLeg, static_cast<unsigned>(row * 49 + colBase), v, v2);  // row2 = row + 18: folded into the odd lanes' offset
        } else {
            j_leg(row, colBase, legMul, rot, k0, k1, k2, k3, v);
            j_leg(row2, colBase, legMul, rot, m0, m1, m2, m3, v2);
        }
    }
    __device__ __forceinline__ void j_base_own2(int row, int row2, int colBase, int legMul, int rot, int k0, int k1, int k2, int k3, int m0, int m1, int m2, int m3, double v,
                                                double v2) const {
        if constexpr (PAIR) {
            BufPut2(buf.pOwnCol, static_cast<unsigned>(row * 49 + colBase), v, v2);  // row2 = row + 1
        } else {
            j_base_own(row, colBase, legMul, rot, k0, k1, k2, k3, v);
            j_base_own(row2, colBase, legMul, rot, m0, m1, m2, m3, v2);
        }
    }
    __device__ __forceinline__ void j_base_shared2(int row, int row2, int colBase, int legMul, int rot, int k0, int k1, int k2, int k3, int m0, int m1, int m2, int m3,
                                                   double v, double v2) const {
        if constexpr (PAIR) {
            BufPut2(buf.pNode, static_cast<unsigned>(row * 49 + colBase), v, v2);
        } else {
            j_base_shared(row, colBase, legMul, rot, k0, k1, k2, k3, v);
            j_base_shared(row2, colBase, legMul, rot, m0, m1, m2, m3, v2);
        }
    }
    /// Four entries of a shared column (same value in the four lanes of a node) in one store instruction:
    /// the lane of leg g writes entry g.
    __device__ __forceinline__ void j_base_shared4(int r0, int r1, int r2, int r3, int col, int k0, int k1, int k2, int k3, double v0, double v1, double v2,
                                                   double v3) const {
        const double v = L == 0 ? v0 : L == 1 ? v1 : L == 2 ? v2 : v3;
        if constexpr (SPARSE) {
            j_sparse(k0, k1, k2, k3, v);
        } else {
            const unsigned e = static_cast<unsigned>((L == 0 ? r0 : L == 1 ? r1 : L == 2 ? r2 : r3) * 49 + col);
            Put(jb + e * je, v);  // per-lane entry: pointer form in both variants (only with --quad-merge-shared)
        }
    }
    __device__ __forceinline__ void j_base_shared(int row, int colBase, int, int, int k0, int, int, int, double v) const {
        if constexpr (SPARSE) {
            if (k0 >= 0) Put(jb + static_cast<unsigned>(k0) * je, v);
        } else if constexpr (BUF) {
            BufPut(buf.vNode, static_cast<unsigned>(row * 49 + colBase), v);
        } else {
            Put(jb + static_cast<unsigned>(row * 49 + colBase) * je, v);
        }
    }
};

/// Offsets of the paired stores: the lane of an even node writes entry e of nodes (n, n + 1) at its own offset; the lane of the odd node
/// n + 1 writes entry e + delta of the same two nodes: one node back, delta entries on (18 rows for the leg rows, one row for the base rows).
template <class BufferState>
__device__ __forceinline__ void QuadPairOffsets(BufferState& buf, int lane, long long je) {
    const bool odd = ((lane >> 4) & 1) != 0;  // QuadNodeInWave<true>: odd nodes in the odd rows
    const int legDelta = odd ? static_cast<int>(static_cast<unsigned>(18LL * 49 * je - 1) * 8u) : 0;
    const int baseDelta = odd ? static_cast<int>(static_cast<unsigned>(49LL * je - 1) * 8u) : 0;
    buf.pLeg = buf.vLeg + legDelta;
    for (int r = 0; r < 4; ++r) buf.pLegCol[r] = buf.vLegCol[r] + legDelta;
    buf.pOwnCol = buf.vOwnCol + baseDelta;
    buf.pNode = buf.vNode + baseDelta;
}

/// Node of a lane inside its wavefront (16 nodes, lane = 16 * row + 4 * leg + j).
///   PAIR = false: node = 4 * row + j        -- 4 adjacent lanes hold the same leg of 4 consecutive nodes: 32-byte runs per 8-byte store;
///   PAIR = true : node = 8 * (row / 2) + 2 * j + row % 2 -- partner nodes (n, n + 1) in the same lane of rows 2 R, 2 R + 1 (what
///                 v_permlane16_swap exchanges); 4 adjacent lanes then write 64 contiguous bytes per 16-byte store.
template <bool PAIR>
__device__ __forceinline__ int QuadNodeInWave(int lane) {
    return PAIR ? 8 * (lane >> 5) + 2 * (lane & 3) + ((lane >> 4) & 1) : 4 * (lane >> 4) + (lane & 3);
}

/// True when the nodes of a launch lie at consecutive addresses of the Jacobian operand (unit-fastest layout, knots of an instance
/// contiguous, instances back to back) and their number is even: what the paired 16-byte stores need.
inline bool QuadPairStoresApply(const NodeLaunch& a) {
    if (a.count % 2 != 0) return false;
    if (a.knots <= 1) return a.jac.bs == 1;
    return a.jac.ks == 1 && a.jac.bs == a.knots;
}

/// GEN is the generated namespace (ValueJacobianQuad, kLegConstantsDev).  BLOCK lanes = BLOCK/4 nodes.
template <int BLOCK, int LDS_SLOTS, int LDS_USLOTS, bool SPARSE, bool STREAM, class Body, class PLAN = NoSparsePlan, class OFF = unsigned, bool BUF = false, bool PAIR = false>
#if defined(UNGAR_QUAD_WAVES_PER_EU)  // experiment knob of tools/quad_bench.hip: ask for N resident wavefronts per SIMD (register budget 512 / N)
#define UNGAR_QUAD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(UNGAR_QUAD_WAVES_PER_EU, UNGAR_QUAD_WAVES_PER_EU)))
#else
#define UNGAR_QUAD_OCCUPANCY
#endif
__global__ __launch_bounds__(BLOCK) UNGAR_QUAD_OCCUPANCY void QuadNodeKernel(const NodeLaunch a, const double (*ctab)[4], Body body) {
    static_assert(BLOCK == 64, "the LDS home is laid out for one wavefront per workgroup");
    __shared__ double lds[(LDS_SLOTS > 0 ? LDS_SLOTS : 1) * BLOCK + LDS_USLOTS * (BLOCK / 4)];
    const int L = (threadIdx.x >> 2) & 3;
    const int nodeInWave = QuadNodeInWave<PAIR>(static_cast<int>(threadIdx.x));
    const long long i = static_cast<long long>(blockIdx.x) * (BLOCK / 4) + nodeInWave;
    if (i >= a.count) return;  // the four lanes of a node leave together
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    double* const fb = a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr;
    double* const jb = a.jac.base + b * a.jac.bs + k * a.jac.ks;
    const long long je = a.jac.es;
    double* const jLeg = jb + 3LL * L * 49 * je;
    QuadIO<SPARSE, STREAM, PLAN, OFF, BUF, PAIR> io{a.x.base + b * a.x.bs + k * a.x.ks,
              a.u.base + b * a.u.bs + k * a.u.ks,
              a.p.base + b * a.p.bs + k * a.p.ks,
              fb,
              jb,
              a.x.es, a.u.es, a.f.es, static_cast<OFF>(je),
              L,
              jLeg,
              {jLeg + 3LL * L * je, jLeg + 3LL * ((L + 1) & 3) * je, jLeg + 3LL * ((L + 2) & 3) * je, jLeg + 3LL * ((L + 3) & 3) * je},
              jb + 3LL * L * je,
              fb ? fb + 3LL * L * a.f.es : nullptr,
              ctab,
              {},
              lds + threadIdx.x,
              lds + LDS_SLOTS * BLOCK + nodeInWave,
              {}};
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BUF) {
        // resource over the operand from the FIRST node of the launch (node 0 of instance 0): every lane offset is >= 0
        io.buf.jr = __builtin_amdgcn_make_buffer_rsrc(a.jac.base, 0, 0xFFFFFFFF, 0x00020000);  // raw buffer, 32-bit data format
        const long long nodeOff = b * a.jac.bs + k * a.jac.ks;  // elements
        io.buf.je8 = static_cast<unsigned>(je) * 8u;
        io.buf.vNode = static_cast<int>(static_cast<unsigned>(nodeOff) * 8u);
        io.buf.vLeg = static_cast<int>(static_cast<unsigned>(nodeOff + 3LL * L * 49 * je) * 8u);
        for (int r = 0; r < 4; ++r) io.buf.vLegCol[r] = static_cast<int>(static_cast<unsigned>(nodeOff + 3LL * L * 49 * je + 3LL * ((L + r) & 3) * je) * 8u);
        io.buf.vOwnCol = static_cast<int>(static_cast<unsigned>(nodeOff + 3LL * L * je) * 8u);
        if constexpr (PAIR) QuadPairOffsets(io.buf, static_cast<int>(threadIdx.x), je);
    }
#endif
    if constexpr (SPARSE) {
#pragma unroll
        for (int p = 0; p < PLAN::kCount; ++p) io.jS[p] = jb + static_cast<long long>(PLAN::kDeltas[p][L]) * je;
    }
    body(io);
}

/// True when every byte offset of the dense 37 x 49 block, measured from the operand's base, fits the 32-bit offsets of
/// buffer instructions: non-negative strides and the last entry of the last node below 4 GiB.
inline bool QuadBufferStoresApply(const NodeLaunch& a) {
    if (a.jac.es < 0 || a.jac.bs < 0 || a.jac.ks < 0) return false;
    const long long instances = a.count / (a.knots > 0 ? a.knots : 1);
    const long long lastNode = (instances - 1) * a.jac.bs + (a.knots - 1) * a.jac.ks;
    return (lastNode + 1813LL * a.jac.es) * 8 < (1LL << 32);
}

/// True when the wave-uniform element offsets of the 37 x 49 block (or of its CSR value array) fit 32 bits.
inline bool QuadOffsetsFit32(const NodeLaunch& a) {
    return a.jac.es >= 0 && a.jac.es * 1813LL < (1LL << 32);
}

}  // namespace ungar_amd::kernels
