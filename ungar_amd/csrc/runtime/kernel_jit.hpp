// ungar_amd :: kernel factory -- size-templated solver kernels instantiated at run time for the sizes a problem declares.
//
// The reference's optimiser takes ANY problem (optimization/concepts.hpp:153-262, soft_sqp.hpp:42-281).  The fast kernels of the batched SQP are
// templates over the stage sizes (register-resident Riccati recursion <NX, NU>, one-wavefront assembly <NZ, NU, NE>): the library ships them
// compiled for the reference's own problems, and everything else is instantiated here on first use with the machinery the function factory
// already has -- `hipcc --genco`, a content-keyed cache entry {meta, code object} published atomically under a lock, hipModuleLoad.  The occupancy
// of an instantiation is not guessed: the candidates (resident wavefronts per SIMD) are compiled side by side and the largest one whose kernel
// needs no scratch memory is kept (what the hand-picked table of round 4 encoded for five sizes).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace ungar_amd::runtime {

struct KernelRequest {
    std::string name;    // entry name, e.g. "riccati_wave_20_9" (file-system safe)
    std::string kernel;  // extern "C" symbol of the kernel in the translation unit
    /// The translation unit; "%W%" stands for the occupancy candidate (resident wavefronts per SIMD).  It #includes kernel headers by their
    /// path relative to the kernel source root (ungar_amd/csrc): every header reachable through #include "..." is part of the cache key.
    std::string source;
    std::vector<int> occupancies;  // candidates, any order
};

struct JitKernel {
    hipFunction_t function = nullptr;  // null under UNGAR_AMD_COMPILE_ONLY (build hosts without a GPU)
    int wavesPerEu = 0, vgprs = 0, scratchBytes = 0;
    bool cacheHit = false;
    std::string object;  // path of the code object
};

/// Finds (in this process, then in the cache folder) or builds the kernel.  Thread-safe; the returned pointer stays valid for the life of the process.
/// nullptr: failure, message in ungar_last_error().
const JitKernel* GetKernel(const KernelRequest& request);

/// .../ungar_amd/csrc: next to the library (ungar_amd/lib/[measurement/]libungar_amd.so), or $UNGAR_AMD_KERNEL_SOURCES.  Empty if not found.
std::string KernelSourceRoot();

}  // namespace ungar_amd::runtime
