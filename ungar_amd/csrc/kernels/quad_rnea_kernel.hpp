// ungar_amd :: kernel skeleton of the lane-per-leg joint-torque program (csrc/codegen/quad_rnea_program.hpp; SURVEY.md section 8(f) N4,
// rbd/quantities/joint_torques.hpp:42-43).  Four lanes own one configuration, one lane per leg; lane layout and leg-to-leg traffic as in
// quad_kernel.hpp (lane = 4 * leg + node % 4 inside each 16-lane DPP row: the legs of a node meet through row_ror, and 4 adjacent lanes
// hold the same leg of 4 consecutive nodes, so unit-fastest stores run 32 bytes contiguous).
//   y (18): rows 0..5 base wrench, 6 + 3 L + k leg torques;   J (18 x 55): columns x = [p | quat | q_leg | v_b | v_leg], u = [a_b | a_leg].
// The centroidal-momentum program (csrc/codegen/quad_centroidal_program.hpp: 6 base rows x 37 columns of x) runs in the same skeleton.
#pragma once

#include <hip/hip_runtime.h>

#include "quad_kernel.hpp"

namespace ungar_amd::kernels {

/// SPARSE: the Jacobian operand is the CSR value array of the model's pattern (every sink carries its index per leg, -1 = structural zero).
/// PLAN (generated: gen::anymal_rnea_quad::SparsePlan): the distinct per-leg index patterns k_L - k_0 of the sparse sinks; one per-lane base
/// pointer per pattern makes a sparse store (lane pointer) + (wave-uniform offset) like a dense one (quad_kernel.hpp).
/// COLS: columns of the node's dense Jacobian block (55 joint torques; 37 centroidal momentum, whose six rows are all base rows).
template <bool SPARSE, bool STREAM, class PLAN = NoSparsePlan, int COLS = 55>
struct QuadRneaIO {
    const double* __restrict__ xb;
    const double* __restrict__ ub;
    double* __restrict__ fb;
    double* __restrict__ jb;
    long long xe, ue, fe;
    unsigned je;
    int L;
    double* __restrict__ jLeg;        // jb + 3 L * 55 * je            : this leg's row block, leg-independent column
    double* __restrict__ jLegCol[4];  // jLeg + 3 ((L + rot) & 3) * je : ... column owned by the leg `rot` lanes away
    double* __restrict__ jOwnCol;     // jb + 3 L * je                  : base rows, column owned by this leg
    double* __restrict__ fLeg;        // fb + 3 L * fe
    const double (*ctab)[4];
    double* lds;
    double* ldsu;
    double* jS[PLAN::kCount > 0 ? PLAN::kCount : 1];  // sparse mode: per-lane base pointer of every index pattern

    __device__ __forceinline__ double qb(int i) const { return xb[i * xe]; }
    __device__ __forceinline__ double vb(int i) const { return xb[(19 + i) * xe]; }
    __device__ __forceinline__ double ab(int i) const { return ub[i * ue]; }
    __device__ __forceinline__ double ql(int i) const { return xb[(7 + 3 * L + i) * xe]; }
    __device__ __forceinline__ double vl(int i) const { return xb[(25 + 3 * L + i) * xe]; }
    __device__ __forceinline__ double al(int i) const { return ub[(6 + 3 * L + i) * ue]; }
    __device__ __forceinline__ double c(int k) const { return ctab[k][L]; }
#ifndef UNGAR_QUAD_NO_PHASE_BARRIER
    __device__ __forceinline__ void phase() const { __builtin_amdgcn_sched_barrier(0); }
#else
    __device__ __forceinline__ void phase() const {}
#endif
    __device__ __forceinline__ void keep(double) const {}
    __device__ __forceinline__ double ld(int slot) const { return lds[slot * 64]; }
    __device__ __forceinline__ void st(int slot, double v) const { lds[slot * 64] = v; }
    __device__ __forceinline__ double ldu(int slot) const { return ldsu[slot * 16]; }
    __device__ __forceinline__ void stu(int slot, double v) const { ldsu[slot * 16] = v; }
    __device__ __forceinline__ double quad_sum(double v) const {
        const double t = v + QuadPerm<0x124>(v);
        return t + QuadPerm<0x128>(t);
    }
    static __device__ __forceinline__ void Put(double* p, double v) { StoreResult<STREAM>(p, v); }
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
        if (k0 >= 0 && k1 >= 0 && k2 >= 0 && k3 >= 0) Put(jb + static_cast<unsigned>(k) * je, v);
        else if (k >= 0) Put(jb + static_cast<unsigned>(k) * je, v);
    }
    __device__ __forceinline__ void j_leg(int rowBase, int colBase, int legMul, int rot, int k0, int k1, int k2, int k3, double v) const {
        if constexpr (SPARSE) j_sparse(k0, k1, k2, k3, v);
        else Put((legMul ? jLegCol[rot] : jLeg) + static_cast<unsigned>(rowBase * COLS + colBase) * je, v);
    }
    __device__ __forceinline__ void j_base_own(int row, int colBase, int k0, int k1, int k2, int k3, double v) const {
        if constexpr (SPARSE) j_sparse(k0, k1, k2, k3, v);
        else Put(jOwnCol + static_cast<unsigned>(row * COLS + colBase) * je, v);
    }
    // shared columns: the four lanes hold the same value and store it to the same address (merged inside the instruction)
    __device__ __forceinline__ void j_base_shared(int row, int col, int k, double v) const {
        if constexpr (SPARSE) {
            if (k >= 0) Put(jb + static_cast<unsigned>(k) * je, v);
        } else {
            Put(jb + static_cast<unsigned>(row * COLS + col) * je, v);
        }
    }
};

/// One wavefront per workgroup, 16 configurations per wavefront.
template <int LDS_SLOTS, int LDS_USLOTS, bool SPARSE, bool STREAM, class Body, class PLAN = NoSparsePlan, int COLS = 55>
__global__ __launch_bounds__(64) void QuadRneaKernel(const NodeLaunch a, const double (*ctab)[4], Body body) {
    __shared__ double lds[(LDS_SLOTS > 0 ? LDS_SLOTS : 1) * 64 + (LDS_USLOTS > 0 ? LDS_USLOTS : 1) * 16];
    const int L = (threadIdx.x >> 2) & 3;
    const int nodeInWave = 4 * (threadIdx.x >> 4) + (threadIdx.x & 3);
    const long long i = static_cast<long long>(blockIdx.x) * 16 + nodeInWave;
    if (i >= a.count) return;  // the four lanes of a configuration leave together
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    double* const fb = a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr;
    double* const jb = a.jac.base + b * a.jac.bs + k * a.jac.ks;
    const long long je = a.jac.es;
    double* const jLeg = jb + 3LL * L * COLS * je;
    QuadRneaIO<SPARSE, STREAM, PLAN, COLS> io{a.x.base + b * a.x.bs + k * a.x.ks,
                                  a.u.base + b * a.u.bs + k * a.u.ks,
                                  fb,
                                  jb,
                                  a.x.es,
                                  a.u.es,
                                  a.f.es,
                                  static_cast<unsigned>(je),
                                  L,
                                  jLeg,
                                  {jLeg + 3LL * L * je, jLeg + 3LL * ((L + 1) & 3) * je, jLeg + 3LL * ((L + 2) & 3) * je, jLeg + 3LL * ((L + 3) & 3) * je},
                                  jb + 3LL * L * je,
                                  fb ? fb + 3LL * L * a.f.es : nullptr,
                                  ctab,
                                  lds + threadIdx.x,
                                  lds + (LDS_SLOTS > 0 ? LDS_SLOTS : 1) * 64 + nodeInWave,
                                  {}};
    if constexpr (SPARSE) {
#pragma unroll
        for (int p = 0; p < PLAN::kCount; ++p) io.jS[p] = jb + static_cast<long long>(PLAN::kDeltas[p][L]) * je;
    }
    body(io);
}

}  // namespace ungar_amd::kernels
