// ungar_amd :: kernel skeleton of the lane-per-leg inertia-matrix program (csrc/codegen/quad_crba_program.hpp; SURVEY.md section 8(f) N4,
// rbd/quantities/joint_space_inertia_matrix.hpp:42-43).  Four lanes own one configuration, one lane per leg (lane layout of quad_kernel.hpp).
//   y (324): M row-major, rows / columns 0..5 base, 6 + 3 L + k leg L;   Jacobian: CSR values of the model's pattern (916 of 324 x 19).
#pragma once

#include <hip/hip_runtime.h>

#include "quad_kernel.hpp"

namespace ungar_amd::kernels {

template <bool STREAM, class PLAN>
struct QuadCrbaIO {
    const double* __restrict__ xb;
    double* __restrict__ fb;
    double* __restrict__ jb;
    long long xe, fe;
    unsigned je;
    int L;
    double* __restrict__ fBL;     // fb + 3 L * fe                                : M[r][6 + 3 L + j]
    double* __restrict__ fLB;     // fb + 3 L * 18 * fe                           : M[6 + 3 L + j][r]
    double* __restrict__ fLL[4];  // fLB + 3 ((L + rot) & 3) * fe                 : M[6 + 3 L + i][6 + 3 ((L + rot) & 3) + j]
    const double (*ctab)[4];
    double* lds;
    double* jS[PLAN::kCount > 0 ? PLAN::kCount : 1];  // per-lane base pointer of every per-leg index pattern

    __device__ __forceinline__ double ql(int i) const { return xb[(7 + 3 * L + i) * xe]; }
    __device__ __forceinline__ double c(int k) const { return ctab[k][L]; }
    __device__ __forceinline__ void phase() const { __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void keep(double) const {}
    __device__ __forceinline__ double ld(int slot) const { return lds[slot * 64]; }
    __device__ __forceinline__ void st(int slot, double v) const { lds[slot * 64] = v; }
    __device__ __forceinline__ double quad_sum(double v) const {
        const double t = v + QuadPerm<0x124>(v);
        return t + QuadPerm<0x128>(t);
    }
    static __device__ __forceinline__ void Put(double* p, double v) { StoreResult<STREAM>(p, v); }
    // (the four lanes hold the same value of a base entry and store it to the same address: merged inside the instruction)
    __device__ __forceinline__ void f_base(int idx, double v) const {
        if (fb) Put(fb + idx * fe, v);
    }
    __device__ __forceinline__ void f_bl(int r, int j, double v) const {
        if (fb) Put(fBL + (r * 18 + 6 + j) * fe, v);
    }
    __device__ __forceinline__ void f_lb(int j, int r, double v) const {
        if (fb) Put(fLB + ((6 + j) * 18 + r) * fe, v);
    }
    __device__ __forceinline__ void f_ll(int i, int j, int rot, double v) const {
        if (fb) Put(fLL[rot] + ((6 + i) * 18 + 6 + j) * fe, v);
    }
    __device__ __forceinline__ void j_sparse(int k0, int k1, int k2, int k3, double v) const {
        if (k0 >= 0 && k1 >= 0 && k2 >= 0 && k3 >= 0) {
#pragma unroll
            for (int p = 0; p < PLAN::kCount; ++p)  // literal arguments, constexpr table: exactly one branch survives
                if (k1 - k0 == PLAN::kDeltas[p][1] && k2 - k0 == PLAN::kDeltas[p][2] && k3 - k0 == PLAN::kDeltas[p][3]) {
                    Put(jS[p] + static_cast<unsigned>(k0) * je, v);
                    return;
                }
        }
        const int k = L == 0 ? k0 : L == 1 ? k1 : L == 2 ? k2 : k3;
        if (k >= 0) Put(jb + static_cast<unsigned>(k) * je, v);
    }
};

/// One wavefront per workgroup, 16 configurations per wavefront.
template <int LDS_SLOTS, bool STREAM, class Body, class PLAN>
__global__ __launch_bounds__(64) void QuadCrbaKernel(const NodeLaunch a, const double (*ctab)[4], Body body) {
    __shared__ double lds[(LDS_SLOTS > 0 ? LDS_SLOTS : 1) * 64];
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
    const long long je = a.jac.es, fe = a.f.es;
    double* const fLB = fb ? fb + 3LL * L * 18 * fe : nullptr;
    QuadCrbaIO<STREAM, PLAN> io{a.x.base + b * a.x.bs + k * a.x.ks,
                                fb,
                                jb,
                                a.x.es,
                                fe,
                                static_cast<unsigned>(je),
                                L,
                                fb ? fb + 3LL * L * fe : nullptr,
                                fLB,
                                {fLB ? fLB + 3LL * L * fe : nullptr, fLB ? fLB + 3LL * ((L + 1) & 3) * fe : nullptr, fLB ? fLB + 3LL * ((L + 2) & 3) * fe : nullptr,
                                 fLB ? fLB + 3LL * ((L + 3) & 3) * fe : nullptr},
                                ctab,
                                lds + threadIdx.x,
                                {}};
#pragma unroll
    for (int p = 0; p < PLAN::kCount; ++p) io.jS[p] = jb + static_cast<long long>(PLAN::kDeltas[p][L]) * je;
    body(io);
}

}  // namespace ungar_amd::kernels
