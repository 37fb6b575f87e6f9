// ungar_amd :: the lane-per-leg node program with the Jacobian stored as REGISTER IMAGES of the wavefront ("wave tiles").
//
// What bounds the lane-per-leg kernel (quad_kernel.hpp) is its result-store path: in the unit-fastest layout a store instruction of the
// wavefront (16 nodes x 4 legs) writes four 128-byte runs that are (count x 8) bytes apart, and 1024 resident wavefronts keep ~1800 such
// streams open each.  Measured with store-only kernels (tools/store_ceiling_tiles.hip, profiles/r06a_store_ceiling_*.log): that pattern
// and every per-wavefront contiguous tile stop at 5.0-5.4 TB/s; what reaches 6.0-6.1 TB/s is (a) one instruction = 1 KiB contiguous and
// (b) the wavefronts that run at the same time filling ONE contiguous band together -- whole DRAM pages are then written within a short
// time window instead of 256 bytes at a time by one wavefront over its whole life.  The layout below is exactly that:
//
//   tile  t = node / 16                       (the 16 nodes of a wavefront; lane = 16 * row + 4 * leg + j holds leg `leg` of node 4 row + j)
//   image s = 0 .. kImages - 1                (one value per lane: entry kEntryOfSlot[4 s + leg] of the lane's node)
//   unit  p = s / 2                           (two images: every lane stores its two values side by side -- 16 bytes per lane, 1 KiB per wavefront,
//                                              ONE buffer_store_dwordx4, no exchange between lanes)
//   band  g = t / kBandTiles, r = t % kBandTiles
//   byte address of unit p of tile t:  ((g * kUnits + p) * kBandTiles + r) * 1024
//   inside the unit: [lane][image s % 2] doubles
//
// i.e. [band][unit][tile of band][64 lanes][2 images]: the kBandTiles wavefronts of a band (consecutive workgroups: resident at the same time,
// round-robin over the XCDs) write 64 KiB contiguous per program step.  Which entry a slot holds is the generated table
// gen::anymal_tiles::kEntryOfSlot (every (row, col) of the dense 37 x 49 block exactly once + 3 padding slots): consumers address the
// operand through it (ungar_tiles_* in include/ungar_amd.h).
// What a store costs the wavefront that issues it (tools/store_issue_cost.hip, profiles/r06a_store_issue_cost.log): the 16-byte store ~4 cycles,
// two 8-byte stores ~52, and the pairwise exchange of the unit-fastest kernel's 16-byte stores (2 x v_permlane16_swap) 32 -- hence two entries of
// the SAME node per store here, not one entry of two nodes.
#pragma once

#include <hip/hip_runtime.h>

#include "quad_kernel.hpp"

namespace ungar_amd::kernels {

inline constexpr int kTileNodes = 16;       // nodes per tile = per wavefront
#if defined(UNGAR_AMD_MEASUREMENT_TILE_BAND)  // tools/quad_tile_bench.hip: sweep of the band size
inline constexpr int kTileBandTiles = UNGAR_AMD_MEASUREMENT_TILE_BAND;
#else
inline constexpr int kTileBandTiles = 64;   // tiles interleaved per band (tools/store_ceiling_tiles.hip: 32-128 is the plateau)
#endif
inline constexpr int kTileUnitBytes = 1024; // two images

/// Doubles of a wave-tile operand holding `count` nodes with `images` images per tile (bands are padded to whole bands).
inline long long TileOperandDoubles(long long count, int images) {
    const long long tiles = (count + kTileNodes - 1) / kTileNodes;
    const long long bands = (tiles + kTileBandTiles - 1) / kTileBandTiles;
    return bands * kTileBandTiles * (images / 2) * (kTileUnitBytes / 8);
}

/// Offset (doubles) of slot `slot` (= 4 * image + leg) of node `i` inside a wave-tile operand.
__host__ __device__ inline long long TileSlotOffset(long long i, int slot, int images) {
    const long long t = i / kTileNodes, g = t / kTileBandTiles, r = t % kTileBandTiles;
    const int s = slot >> 2, leg = slot & 3, p = s >> 1, n = static_cast<int>(i % kTileNodes);
    const int lane = 16 * (n >> 2) + 4 * leg + (n & 3);
    return ((g * (images / 2) + p) * kTileBandTiles + r) * (kTileUnitBytes / 8) + 2 * lane + (s & 1);
}

/// I/O policy of gen::anymal_tiles::ValueJacobianQuadTiles<double>: inputs, LDS home, DPP exchanges and value stores of QuadIO; the
/// Jacobian leaves through t_put2.  GUARD (measurement only): the stores sit behind a wave-uniform flag (compute floor of the program).
template <bool STREAM, bool GUARD = false>
struct QuadTileIO : QuadIO<false, STREAM> {
    using Base = QuadIO<false, STREAM>;
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t tr;  // resource over the band, from unit 0 of this wavefront's tile
#endif
    int tv;            // lane offset inside a unit (bytes)
    bool storeFlag;    // GUARD only
#if defined(UNGAR_AMD_MEASUREMENT_STORE_AUX)
    static constexpr int kAux = UNGAR_AMD_MEASUREMENT_STORE_AUX;
#else
    static constexpr int kAux = STREAM ? 2 : 0;
#endif
    /// Lane of leg q keeps v_q (base rows of the shared columns: the same four values in the four lanes of a node).
    __device__ __forceinline__ double sel4(double v0, double v1, double v2, double v3) const {
        const int L = Base::L;
        return L == 0 ? v0 : L == 1 ? v1 : L == 2 ? v2 : v3;
    }
    /// Images 2 p (value v) and 2 p + 1 (value v2) of the lane's node, side by side: one 16-byte store per lane, 1 KiB contiguous per wavefront.
    __device__ __forceinline__ void t_put2(int p, double v, double v2) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef int v4i __attribute__((ext_vector_type(4)));
        if constexpr (GUARD) {
            if (!storeFlag) {
                asm volatile("" ::"v"(v), "v"(v2));
                return;
            }
        }
        const v4i q{__double2loint(v), __double2hiint(v), __double2loint(v2), __double2hiint(v2)};
        __builtin_amdgcn_raw_buffer_store_b128(q, tr, tv, p * (kTileBandTiles * kTileUnitBytes), kAux);
        asm volatile("s_nop 1" ::"v"(q) : "memory");  // data hazard of 16-byte buffer stores with a scalar offset (quad_kernel.hpp: BufPut2)
#else
        (void)p, (void)v, (void)v2;
#endif
    }
};

/// One wavefront = one tile.  `tiles` is the wave-tile operand of the launch (tile 0 = nodes 0..15 of the launch).  Lanes past the last node
/// recompute the last node (their slots of the last tile are padding).
template <int LDS_SLOTS, int LDS_USLOTS, bool STREAM, class Body, bool GUARD = false>
__global__ __launch_bounds__(64) void QuadTileKernel(const NodeLaunch a, const double (*ctab)[4], double* tiles, int images, int storeFlag, Body body) {
    __shared__ double lds[(LDS_SLOTS > 0 ? LDS_SLOTS : 1) * 64 + LDS_USLOTS * 16];
    const int lane = static_cast<int>(threadIdx.x);
    const int L = (lane >> 2) & 3;
    const int nodeInWave = QuadNodeInWave<false>(lane);
    long long i = static_cast<long long>(blockIdx.x) * kTileNodes + nodeInWave;
    if (i >= a.count) i = a.count - 1;
    long long b = i, k = 0;
    if (a.knots > 1) {
        b = i / a.knots;
        k = i - b * a.knots;
    }
    double* const fb = a.f.base ? a.f.base + b * a.f.bs + k * a.f.ks : nullptr;
    QuadTileIO<STREAM, GUARD> io{{a.x.base + b * a.x.bs + k * a.x.ks,
                                  a.u.base + b * a.u.bs + k * a.u.ks,
                                  a.p.base + b * a.p.bs + k * a.p.ks,
                                  fb,
                                  nullptr,
                                  a.x.es, a.u.es, a.f.es, 0u,
                                  L,
                                  nullptr,
                                  {nullptr, nullptr, nullptr, nullptr},
                                  nullptr,
                                  fb ? fb + 3LL * L * a.f.es : nullptr,
                                  ctab,
                                  {},
                                  lds + threadIdx.x,
                                  lds + LDS_SLOTS * 64 + nodeInWave,
                                  {}}};
#if defined(__HIP_DEVICE_COMPILE__)
    const long long t = blockIdx.x, g = t / kTileBandTiles, r = t % kTileBandTiles;
    char* const first = reinterpret_cast<char*>(tiles) + ((g * (images / 2)) * kTileBandTiles + r) * static_cast<long long>(kTileUnitBytes);
    io.tr = __builtin_amdgcn_make_buffer_rsrc(first, 0, 0xFFFFFFFF, 0x00020000);
    io.tv = lane * 16;
#endif
    io.storeFlag = storeFlag != 0;
#if defined(UNGAR_AMD_MEASUREMENT_TILE_STAGGER) && defined(__HIP_DEVICE_COMPILE__)  // tools/quad_tile_bench.hip: de-phase the first generation of wavefronts
    if (blockIdx.x < 1024) {
        const unsigned n = ((blockIdx.x * 2654435761u) >> 20) % UNGAR_AMD_MEASUREMENT_TILE_STAGGER;
        for (unsigned d = 0; d < n; ++d) __builtin_amdgcn_s_sleep(64);  // 4096 cycles each
    }
#endif
    body(io);
}

}  // namespace ungar_amd::kernels
