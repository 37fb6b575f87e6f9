// ungar_amd :: 'anymal' dense [A|B] block in the WAVE-TILE layout (quad_tile_kernel.hpp; DESIGN.md sections 3 and 4.5): the lane-per-leg node
// program with its results stored as register images of the wavefront, 1 KiB contiguous per store instruction, the wavefronts of a band filling
// 64 KiB together -- and the conversion of a tile operand into any strided operand (ungar_tiles_gather) for consumers that want rows.
#include "../runtime/measurement.hpp"
#include "../gen/anymal_quad_gen.hpp"
#include "../gen/anymal_tiles_gen.hpp"

#include "quad_tile_kernel.hpp"

namespace ungar_amd::kernels {
namespace TG = ungar_amd::gen::anymal_tiles;

struct AnymalTileBody {
    template <class IO>
    __device__ __forceinline__ void operator()(IO& io) const { TG::ValueJacobianQuadTiles<double>(io); }
};

/// dst[node offset + entry * es] = tile slot, one wavefront per tile: every lane reads the 16 bytes it would have written (two images of its
/// (leg, node)) unit by unit -- 1 KiB coalesced per instruction -- and scatters the two entries.
__global__ __launch_bounds__(64) void TilesGatherKernel(const double* __restrict__ tiles, int images, OperandView dst, long long count, long long knots) {
    const int lane = static_cast<int>(threadIdx.x);
    const int leg = (lane >> 2) & 3;
    const long long i = static_cast<long long>(blockIdx.x) * kTileNodes + QuadNodeInWave<false>(lane);
    const long long t = blockIdx.x, g = t / kTileBandTiles, r = t % kTileBandTiles;
    const double2* unit = reinterpret_cast<const double2*>(tiles + ((g * (images / 2)) * kTileBandTiles + r) * (kTileUnitBytes / 8)) + lane;
    if (i >= count) return;
    long long b = i, k = 0;
    if (knots > 1) {
        b = i / knots;
        k = i - b * knots;
    }
    double* const out = dst.base + b * dst.bs + k * dst.ks;
    for (int p = 0; p < images / 2; ++p) {
        const double2 v = unit[static_cast<long long>(p) * kTileBandTiles * (kTileUnitBytes / 16)];
        const int e0 = TG::kEntryOfSlotDev[8 * p + leg], e1 = TG::kEntryOfSlotDev[8 * p + 4 + leg];
        if (e0 >= 0) out[e0 * dst.es] = v.x;
        if (e1 >= 0) out[e1 * dst.es] = v.y;
    }
}
}  // namespace ungar_amd::kernels

extern "C" void ungar_amd_anymal_tile_layout(int* images, const short** entryOfSlot) {
    *images = ungar_amd::gen::anymal_tiles::kImages;
    *entryOfSlot = ungar_amd::gen::anymal_tiles::kEntryOfSlot;
}

extern "C" int ungar_amd_launch_anymal_tiles(const ungar_amd::kernels::NodeLaunch* a, double* tiles, void* stream) {
    using namespace ungar_amd::kernels;
    if (a->count <= 0) return 0;
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(ungar_amd::gen::anymal_quad::kLegConstantsDev));
    if (e != hipSuccess) return static_cast<int>(e);
    const double(*ctab)[4] = static_cast<const double(*)[4]>(sym);
    const dim3 grid(static_cast<unsigned>((a->count + kTileNodes - 1) / kTileNodes)), block(64);
    // non-temporal stores for outputs beyond the last-level cache (node_kernel.hpp: UseStreamingStores); write-back below
    const bool stream_ = TileOperandDoubles(a->count, TG::kImages) * 8 > (256LL << 20);
    NodeLaunch launch = *a;
    launch.jac = {};
    if (stream_)
        hipLaunchKernelGGL((QuadTileKernel<TG::kLdsSlots, TG::kLdsUniformSlots, true, AnymalTileBody>), grid, block, 0, static_cast<hipStream_t>(stream), launch, ctab, tiles,
                           TG::kImages, 1, AnymalTileBody{});
    else
        hipLaunchKernelGGL((QuadTileKernel<TG::kLdsSlots, TG::kLdsUniformSlots, false, AnymalTileBody>), grid, block, 0, static_cast<hipStream_t>(stream), launch, ctab, tiles,
                           TG::kImages, 1, AnymalTileBody{});
    return static_cast<int>(hipGetLastError());
}

extern "C" int ungar_amd_launch_anymal_tiles_gather(const double* tiles, const ungar_amd::kernels::OperandView* dst, long long count, long long knots, void* stream) {
    using namespace ungar_amd::kernels;
    if (count <= 0) return 0;
    const dim3 grid(static_cast<unsigned>((count + kTileNodes - 1) / kTileNodes)), block(64);
    hipLaunchKernelGGL(TilesGatherKernel, grid, block, 0, static_cast<hipStream_t>(stream), tiles, TG::kImages, *dst, count, knots);
    return static_cast<int>(hipGetLastError());
}
