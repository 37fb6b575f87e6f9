// ungar_amd :: the Riccati recursion of ocp_riccati.hpp with ONE WAVEFRONT PER INSTANCE and every matrix of the recursion in REGISTERS, in the operand /
// accumulator layouts of v_mfma_f64_16x16x4_f64 (replaces the OSQP call of soft_sqp.hpp:143-158 for shooting structure).
//
// Why.  The LDS-resident kernels (ocp_riccati.hip) run one workgroup of four wavefronts per instance: 66-78 KB of LDS per instance keep two workgroups on
// a CU, every phase ends in a workgroup barrier and costs its critical-path instruction count -- 0.10-0.15 of the roofline for the
// 37 + 12, 25 + 24 and 13 + 24 blocks.  Here nothing is shared between wavefronts, so there is no barrier at all, and the products feed each other WITHOUT
// any data movement because of how the matrix instruction lays its operands out over the lanes:
//     A[i][k] in lane 16 k + i,   B[k][j] in lane 16 k + j,   D[(lane >> 4) + 4 r][lane & 15] in accumulator element r.
//   * element r of an accumulator tile is rows 4 r .. 4 r + 3 of the tile in exactly the B layout: an accumulator is the B operand of the next product,
//     k-step by k-step (P [A|B] -> [A|B]^T (P [A|B]));
//   * the A and the B layout coincide, so the registers that hold [A|B] as B operand of P [A|B] are the A operand ([A|B]^T) of the second product;
//   * P is symmetric: its accumulator tiles are the A operand of P [A|B] (A[i][k] = P[k][i]).
// The affine parts ride along in HOMOGENEOUS coordinates -- x_e = [x; 1], P_e = [P p; p^T *], [A|B]_e = [A b B; 0 1 0] -- so that
//     H_e = W_e + [A|B]_e^T P_e [A|B]_e = [H_xx h_x H_xu; h_x^T * h_u^T; H_ux h_u R]      (one chain of matrix instructions, no separate vector updates),
//     [K | kff] = -R^-1 [H_ux | h_u],   P_e' = H_e,xx + H_e,xu [K | kff]                    (ocp_riccati.hpp: the same recursion).
// R is factorised by symmetric elimination without pivoting on the tableau [R | H_ux h_u], lane = column, rows in registers, pivot columns broadcast with
// v_readlane: the multipliers and pivots of the right-looking L D L^T of ocp_riccati.hpp (a pivot that is not positive is replaced by 1 and reported).
// LDS (a few KB per wavefront, private to it) is used only to change layouts: accumulator tiles -> tableau columns, gains -> B operand, tile transposes.
#include "../runtime/measurement.hpp"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <mutex>
#include <string>
#include <utility>

#include "../../../include/ungar_amd.h"
#include "../runtime/kernel_jit.hpp"
#include "ocp_riccati_wave_kernel.hpp"

namespace ungar_amd::kernels {
namespace {

template <int NX, int NU, int WAVES_PER_EU>
int LaunchWave(const RiccatiArgs* a, hipStream_t stream) {
    constexpr std::size_t lds = static_cast<std::size_t>(WaveSizes<NX, NU>::kLdsDoubles) * sizeof(double);
    static const bool clocks = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_RICCATI_WAVE_CLOCKS") != nullptr;
    if (clocks) hipLaunchKernelGGL((RiccatiWaveKernel<NX, NU, WAVES_PER_EU, true>), dim3(static_cast<unsigned>(a->batch)), dim3(64), lds, stream, *a);
    else hipLaunchKernelGGL((RiccatiWaveKernel<NX, NU, WAVES_PER_EU>), dim3(static_cast<unsigned>(a->batch)), dim3(64), lds, stream, *a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace
}  // namespace ungar_amd::kernels

using namespace ungar_amd::kernels;

namespace {

/// The instantiation for sizes the library was not compiled for: made by the kernel factory on first use (cached on disk and in the process), occupancy chosen
/// among the candidates its LDS admits as the largest whose kernel keeps the recursion's matrices in registers.  nullptr: the factory failed (reported once).
const ungar_amd::runtime::JitKernel* FactoryKernel(int nx, int nu) {
    // (asked once per size and process: the factory keys an entry by the CONTENT of the kernel sources, which it reads -- not something to do per launch)
    static std::mutex mutex;
    static std::map<std::tuple<int, int, int>, const ungar_amd::runtime::JitKernel*> known;  // per (shape, device): a code object is loaded into one device's context
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0;
    std::lock_guard<std::mutex> guard(mutex);
    if (auto it = known.find({nx, nu, device}); it != known.end()) return it->second;
    const std::size_t lds = static_cast<std::size_t>(RiccatiWaveLdsDoubles(nx, nu)) * sizeof(double);
    ungar_amd::runtime::KernelRequest rq;
    rq.name = "riccati_wave_" + std::to_string(nx) + "_" + std::to_string(nu);
    rq.kernel = "ungar_riccati_wave";
    rq.source = "#include \"kernels/ocp_riccati_wave_kernel.hpp\"\n"
                "extern \"C\" __global__ __launch_bounds__(64, %W%) void ungar_riccati_wave(const ungar_amd::kernels::RiccatiArgs a) {\n"
                "    ungar_amd::kernels::RiccatiWaveBody<" + std::to_string(nx) + ", " + std::to_string(nu) + ">(a);\n}\n";
    const int perSimd = static_cast<int>((160 * 1024 / lds + 3) / 4);  // wavefronts per SIMD the LDS of a CU admits
    for (int w : {1, 2, 4})
        if (w == 1 || w <= perSimd) rq.occupancies.push_back(w);
    const ungar_amd::runtime::JitKernel* k = ungar_amd::runtime::GetKernel(rq);
    if (!k) {
        static bool reported = false;
        if (!reported) std::fprintf(stderr, "[ungar_amd] register-resident Riccati kernel %d + %d not available (%s): the LDS-resident kernels take over\n", nx, nu, ungar_last_error());
        reported = true;
    }
    // an instantiation that needs scratch memory even with the whole register file (tableaus near 64 columns) is not kept: the LDS-resident kernels serve that size
    const bool transient = !k;  // a failed build (compiler missing just now, disk full) is asked again next time; a kernel that needs scratch is a property of the size
    if (k && k->scratchBytes > 0) k = nullptr;
    if (!transient) known[{nx, nu, device}] = k;
    return k;
}

bool Prebuilt(int nx, int nu) {
    return (nx == 37 && nu == 12) || (nx == 25 && nu == 24) || (nx == 13 && nu == 24) || (nx == 13 && nu == 4) || (nx == 17 && nu == 4);
}

/// Stage sizes the register-resident recursion is used for: the tableau must fit the 64 lanes, and below ~12 stage variables the LDS-resident kernels
/// win (8 + 2, 6 + 2 measured: 0.15 / 0.12 ms against 0.12 / 0.11 -- one instance fills 10 of a wavefront's 64 lanes).
bool WaveRouteApplies(int nx, int nu) {
    return RiccatiWaveFits(nx, nu) && nx + nu >= 12;
}

}  // namespace

/// Which recursion a QP of these stage sizes runs: 0 LDS-resident kernels (ocp_riccati.hip), 1 register-resident kernel compiled into the library,
/// 2 register-resident kernel from the kernel factory (built now if `prepare`, otherwise on the first solve).
extern "C" int ungar_amd_riccati_route(int nx, int nu, int ne, int prepare) {
    if (ne != 0 || !WaveRouteApplies(nx, nu)) return 0;
    if (Prebuilt(nx, nu)) return 1;
    if (prepare && !FactoryKernel(nx, nu)) return 0;
    return 2;
}

/// 0 / a HIP error: launched;  -1: no register-resident instantiation for these sizes (the caller falls back to the LDS-resident kernels).
extern "C" int ungar_amd_launch_riccati_wave(const RiccatiArgs* a, void* stream) {
    if (a->ne != 0) return -1;
    if (a->jac.es != 1 || a->b.es != 1 || a->hess.es != 1 || a->grad.es != 1) return -1;  // a knot's operands as contiguous row-major blocks (what the assembly kernels write)
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a->nx == 37 && a->nu == 12) return LaunchWave<37, 12, 1>(a, s);
    if (a->nx == 25 && a->nu == 24) return LaunchWave<25, 24, 1>(a, s);
    if (a->nx == 13 && a->nu == 24) return LaunchWave<13, 24, 2>(a, s);
    // (the small blocks: 13 + 4 takes 0.22 ms per 4096 x 30 knots here against 0.31 LDS-resident; 17 + 4 -- two state tiles for 18 columns -- 0.46 against 0.50 at TWO
    // wavefronts per SIMD (203 registers; asked to fit four it spills and takes 0.85-1.05 ms: operands-to-registers 23 k cycles per knot); 8 + 2 and 6 + 2 measured
    // slower -- 0.15 / 0.12 against 0.12 / 0.11 ms -- and stay with ocp_riccati.hip)
    if (a->nx == 13 && a->nu == 4) return LaunchWave<13, 4, 4>(a, s);
    if (a->nx == 17 && a->nu == 4) return LaunchWave<17, 4, 2>(a, s);
    // any other size the kernel fits: the instantiation of the kernel factory (runtime/kernel_jit.cpp)
    if (!WaveRouteApplies(a->nx, a->nu)) return -1;
    const ungar_amd::runtime::JitKernel* k = FactoryKernel(a->nx, a->nu);
    if (!k || !k->function) return -1;
    RiccatiArgs args = *a;
    void* params[] = {&args};
    const unsigned ldsBytes = static_cast<unsigned>(RiccatiWaveLdsDoubles(a->nx, a->nu) * sizeof(double));
    if (ldsBytes > 64u * 1024u) {  // (60 + 3, 61 + 2, 62 + 1: 65.7-66.8 KB) dynamic LDS beyond 64 KiB has to be asked for
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k->function), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes));
        if (ea != hipSuccess) {
            (void)hipGetLastError();
            return -1;  // the LDS-resident kernels take the solve
        }
    }
    const hipError_t e = hipModuleLaunchKernel(k->function, static_cast<unsigned>(a->batch), 1, 1, 64, 1, 1, ldsBytes, s, params, nullptr);
    return static_cast<int>(e);
}
