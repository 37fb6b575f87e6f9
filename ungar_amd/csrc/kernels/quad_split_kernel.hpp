// ungar_amd :: kernel skeleton of the SPLIT lane-per-leg node program.
//
// The fused lane-per-leg program (quad_kernel.hpp) keeps ~258 doubles alive per lane: 512 registers + 40 KiB of LDS per
// wavefront, ONE wavefront per SIMD -- and a lone wavefront issues a v_fma_f64 every 10.2 cycles where two wavefronts issue
// one every 6.7 (profiles/archive/r04e_quad_cycle_model.md).  Here the program is cut along its data flow into two halves that run
// as the two wavefronts of a 128-lane workgroup, each with at most 256 registers and half of the workgroup's 40 KiB:
//
//   producer (wavefront 0)                         consumer (wavefront 1)
//   leg kinematics, bias forces      --- m0 --->   composite inertias, block-arrow U D U^T, primal solve
//   RNEA tangents w.r.t. v_L, v_b    --- m1..9 ->  solves, integrator chain, stores of the 37 x 49 block
//     (need no accelerations)        <-- acc ---   (columns it can do alone are interleaved: u_L, quaternion, position)
//   RNEA tangents w.r.t. q_L         --- m10..12 ->
//
// Same lane layout in both halves (lane = 4 * leg + node % 4 inside each 16-lane row), so a message item is a per-lane LDS
// slot: written by lane l of the producer, read by lane l of the consumer, conflict-free, no cross-lane traffic.  Messages
// travel through a ring of kRing slots; three counters in LDS (posted / consumed / accelerations posted) order the two
// wavefronts -- no workgroup barrier after the first one, and neither half ever waits on vmcnt for the other (LDS operations
// of one wavefront execute in issue order, so a counter written after its items is seen after them).
#pragma once

#include <hip/hip_runtime.h>

#include "quad_kernel.hpp"

namespace ungar_amd::kernels {

/// QuadIO plus the channel between the two halves.
template <bool STREAM, class OFF, bool BUF, int RING, bool PAIR = false>
struct QuadSplitIO : QuadIO<false, STREAM, NoSparsePlan, OFF, BUF, PAIR> {
    using Base = QuadIO<false, STREAM, NoSparsePlan, OFF, BUF, PAIR>;
    // Channel pointers carry the LDS address space explicitly: every access must be a DS instruction.  A generic pointer that the
    // compiler cannot trace back to the __shared__ array (the volatile counters were) becomes a FLAT access, which travels through the
    // vector-memory pipe -- NOT ordered with the DS instructions around it, and followed by s_waitcnt vmcnt(0), i.e. by a drain of
    // the consumer's whole result-store queue.
    using LdsDouble = __attribute__((address_space(3))) double;
    using LdsFlag = __attribute__((address_space(3))) volatile int;
    LdsDouble* ring;     // this lane's column of the ring: item i of message m at ring[((m % RING) * 9 + i) * 64]
    LdsDouble* accNode;  // a_b + gamma, one copy per node: item k at accNode[k * 16]
    LdsDouble* accLane;  // a_L: item k at accLane[k * 64]
    LdsFlag* flags;      // [0] messages posted, [1] messages consumed, [2] accelerations posted

    static __device__ __forceinline__ void CompilerFence() { asm volatile("" ::: "memory"); }
    static __device__ __forceinline__ void Spin(LdsFlag* flag, int atLeast) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UNGAR_AMD_MEASUREMENT_SPLIT_NO_SYNC)  // (timing experiments of tools/quad_split_bench.hip: one half alone)
        while (__builtin_amdgcn_readfirstlane(*flag) < atLeast) __builtin_amdgcn_s_sleep(1);
#else
        (void)flag, (void)atLeast;
#endif
        CompilerFence();
    }
    // producer
    __device__ __forceinline__ void wait_free(int m) const {
        if (m >= RING) Spin(flags + 1, m - RING + 1);
    }
    __device__ __forceinline__ void send(int m, int i, double v) const { ring[((m % RING) * 9 + i) * 64] = v; }
    __device__ __forceinline__ void post(int m) const {
        CompilerFence();
        flags[0] = m + 1;
        CompilerFence();
    }
    __device__ __forceinline__ void wait_acc() const { Spin(flags + 2, 1); }
    __device__ __forceinline__ double acc(int k) const { return k < 6 ? accNode[k * 16] : accLane[(k - 6) * 64]; }
    // consumer
    __device__ __forceinline__ void wait(int m) const { Spin(flags + 0, m + 1); }
    __device__ __forceinline__ double recv(int m, int i) const { return ring[((m % RING) * 9 + i) * 64]; }
    __device__ __forceinline__ void done(int m) const {
        CompilerFence();
        flags[1] = m + 1;
        CompilerFence();
    }
    __device__ __forceinline__ void send_acc(int k, double v) const {
        if (k < 6) accNode[k * 16] = v;
        else accLane[(k - 6) * 64] = v;
    }
    __device__ __forceinline__ void post_acc() const {
        CompilerFence();
        flags[2] = 1;
        CompilerFence();
    }
};

/// GEN-side bodies: PRODUCER / CONSUMER are functors calling gen::anymal_split::ProducerQuad / ConsumerQuad.
/// LDS per workgroup: homes of the two halves, ring, accelerations, counters -- 40 KiB at most (four workgroups per CU).
template <int PS, int PU, int CS, int CU_, int RING, bool STREAM, class Producer, class Consumer, class OFF = unsigned, bool BUF = false, bool PAIR = false>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void QuadSplitKernel(const NodeLaunch a, const double (*ctab)[4], Producer producer,
                                                                                                   Consumer consumer) {
    constexpr int kProducerHome = PS * 64 + PU * 16, kConsumerHome = CS * 64 + CU_ * 16, kRingDoubles = RING * 9 * 64, kAcc = 6 * 16 + 3 * 64;
    constexpr int kDoubles = kProducerHome + kConsumerHome + kRingDoubles + kAcc + 2;
    static_assert(kDoubles * 8 <= 40 * 1024, "four workgroups must fit the CU's 160 KiB of LDS");
    __shared__ double lds[kDoubles];
    using IO = QuadSplitIO<STREAM, OFF, BUF, RING, PAIR>;
    typename IO::LdsDouble* const ldsChannel = (typename IO::LdsDouble*)(lds + kProducerHome + kConsumerHome);  // ring, accelerations, counters
    typename IO::LdsFlag* const flags = (typename IO::LdsFlag*)(ldsChannel + kRingDoubles + kAcc);
    if (threadIdx.x < 4) flags[threadIdx.x] = 0;
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    const int role = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));  // wave-uniform: 0 producer, 1 consumer
#else
    const int role = 0;
#endif
    const int lane = threadIdx.x & 63;
    const int L = (lane >> 2) & 3;
    const int nodeInWave = QuadNodeInWave<PAIR>(lane);
    const long long i = static_cast<long long>(blockIdx.x) * 16 + nodeInWave;
    if (i >= a.count) return;  // the four lanes of a node leave together, in both wavefronts
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    double* const fb = a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr;
    double* const jb = a.jac.base + b * a.jac.bs + k * a.jac.ks;
    const long long je = a.jac.es;
    double* const jLeg = jb + 3LL * L * 49 * je;
    double* const home = lds + (role ? kProducerHome : 0);
    IO io{{a.x.base + b * a.x.bs + k * a.x.ks,
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
                                            home + lane,
                                            home + (role ? CS : PS) * 64 + nodeInWave,
                                            {}},
                                           ldsChannel + lane,
                                           ldsChannel + kRingDoubles + nodeInWave,
                                           ldsChannel + kRingDoubles + 6 * 16 + lane,
                                           flags};
    if (role == 0) {
        producer(io);
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BUF) {
        io.buf.jr = __builtin_amdgcn_make_buffer_rsrc(a.jac.base, 0, 0xFFFFFFFF, 0x00020000);
        const long long nodeOff = b * a.jac.bs + k * a.jac.ks;  // elements
        io.buf.je8 = static_cast<unsigned>(je) * 8u;
        io.buf.vNode = static_cast<int>(static_cast<unsigned>(nodeOff) * 8u);
        io.buf.vLeg = static_cast<int>(static_cast<unsigned>(nodeOff + 3LL * L * 49 * je) * 8u);
        for (int r = 0; r < 4; ++r) io.buf.vLegCol[r] = static_cast<int>(static_cast<unsigned>(nodeOff + 3LL * L * 49 * je + 3LL * ((L + r) & 3) * je) * 8u);
        io.buf.vOwnCol = static_cast<int>(static_cast<unsigned>(nodeOff + 3LL * L * je) * 8u);
        if constexpr (PAIR) QuadPairOffsets(io.buf, lane, je);
    }
#endif
    consumer(io);
}

}  // namespace ungar_amd::kernels
