// ungar_amd :: run-time function factory -- record -> derive -> HIP codegen -> hipcc -> hipModule.
//
// MI355X-native counterpart of FunctionFactory::Worker (reference
// include/ungar/autodiff/function.hpp:392-605):
//   CppAD tape + optimize            -> tape::Graph (hash-consed, built optimised)
//   ModelCSourceGen sparse Jac/Hess  -> tape::Differentiator (parameters trimmed, upper-tri Hessian)
//   GccCompiler / createDynamicLibrary -> `hipcc --offload-arch=gfx950 --genco`  (process boundary)
//   LinuxDynamicLib / dlopen         -> hipModuleLoad / hipModuleGetFunction
//   existence-only .so cache (function.hpp:420-451, stale-cache hazard)
//                                    -> versioned cache entry {meta (sparsity, kernel table), code objects} keyed by a hash of
//                                       (optimised tape, enabled derivatives, arch, ROCm version, emitter build id, JIT flags):
//                                       a hit skips derive / emit / compile altogether
// Kernels map one lane to one problem instance and address operands through strides, so the same
// code object serves batch = 1 host calls (what Ungar::Autodiff::Function needs) and large batches.
#include "measurement.hpp"
#include <hip/hip_runtime.h>

#include <atomic>
#include <immintrin.h>
#include <chrono>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/ungar_amd.h"
#include "../tape/emit.hpp"
#include "jit_common.hpp"

namespace ungar_amd::runtime {
int Fail(int code, const std::string& msg);  // c_api.cpp
}

using namespace ungar_amd;

struct ungar_function {
    std::string name;
    int64_t n = 0, p = 0, m = 0;
    uint32_t enabled = 0;
    std::vector<int32_t> jacRows, jacCols, hesRows, hesCols;
    // value, Jacobian, Hessian: one code object each -- or several, one per CHUNK of consecutive outputs, when the body exceeds kChunkStatements
    // statements (all compiled concurrently; the chunks of one derivative are launched one after the other on the caller's stream)
    using Kernels = std::vector<hipFunction_t>;
    std::vector<hipModule_t> modules;
    Kernels kValue, kJac, kHes;
    std::string codeObjectPath;
    // the tape as the caller recorded it (ungar_function_get_tape): what a caller needs to build ONE function out of several -- the stage values of a shooting
    // problem evaluated in a single launch (ungar/optimization/batched_soft_sqp.hpp)
    std::vector<ungar_tape_node> tapeNodes;
    std::vector<int32_t> tapeOutputs;
    std::string folder;
    bool cacheHit = false;
    // staging buffers for the single-instance host entry points
    double *dIn = nullptr, *dOut = nullptr;
    int64_t dOutSize = 0;
    // single-instance host calls: pinned input staging, outputs written by the kernel straight into mapped host memory, one private stream
    double *hIn = nullptr, *hOut = nullptr, *hOutDevice = nullptr;
    double* hInDevice = nullptr;                     // device address of hIn (mapped): small inputs are READ by the kernel over the bus, no copy command
    unsigned long long *hFlag = nullptr, *hFlagDevice = nullptr;  // completion word in mapped host memory, written by the stream behind the kernel, polled by the host
    unsigned long long hostCalls = 0;
    int64_t hOutSize = 0;
    hipStream_t hostStream = nullptr;
    // RESIDENT single-instance kernels (value, Jacobian, Hessian: `<kernel>_serve` of the same code object, EmitKernel): a wavefront that stays on the device
    // between host calls, is handed a call through a doorbell word and answers through a word of mapped host memory -- no launch per call.
    // answers: two 64-bit words per derivative in mapped host memory -- [0] last ticket answered, [1] generation of the last launch that has returned.
    // The doorbell and the inputs live in device memory that the host writes through the PCIe aperture where the device exposes all of its memory (large BAR):
    // the wavefront then polls and fetches locally (3.2 against 5.0 us per round trip); in mapped host memory otherwise.
    struct Resident {
        hipFunction_t kernel = nullptr;
        unsigned long long *answers = nullptr, *answersDevice = nullptr;
        unsigned long long tickets = 0, generation = 0;  // generation of the launch that may still be running (0: never launched)
        hipStream_t stream = nullptr;                    // its own: the function's other launches do not queue behind a kernel that is waiting for calls
    } resident[3];
    void* aperture = nullptr;     // device memory written by the host: inputs at 0, the doorbell of derivative w at 2048 + 64 w (null: mapped host memory instead)
    unsigned long long* rings = nullptr;  // doorbells in mapped host memory (no aperture): word 8 w
    unsigned long long* ringsDevice = nullptr;
    bool residentReady = false;
    ~ungar_function() {
        for (Resident& r : resident) {
            if (r.stream) {
                (void)hipStreamSynchronize(r.stream);  // (a resident kernel returns by itself once nobody calls; it reads the buffers freed below)
                (void)hipStreamDestroy(r.stream);
            }
            if (r.answers) (void)hipHostFree(r.answers);
        }
        if (aperture) (void)hipFree(aperture);
        if (rings) (void)hipHostFree(rings);
        if (dIn) (void)hipFree(dIn);
        if (dOut) (void)hipFree(dOut);
        if (hIn) (void)hipHostFree(hIn);
        if (hOut) (void)hipHostFree(hOut);
        if (hFlag) (void)hipHostFree(hFlag);
        if (hostStream) (void)hipStreamDestroy(hostStream);
        for (hipModule_t mod : modules)
            if (mod) (void)hipModuleUnload(mod);
    }
};

namespace {

using runtime::Fail;
using namespace runtime::jit;

constexpr uint32_t kEnableJacobian = 1U << 1;  // EnabledDerivatives::JACOBIAN, autodiff/data_types.hpp:95-100
constexpr uint32_t kEnableHessian = 1U << 2;   // EnabledDerivatives::HESSIAN

// Everything that decides what a cache entry contains, besides the tape itself.  UNGAR_AMD_EMITTER_ID is a hash of the
// tape engine's sources (csrc/tape/*.hpp, this file) injected by the build (ungar_amd/_build.py): editing the derivative
// transforms or the emitter invalidates every entry without anyone remembering to bump a version.
#ifndef UNGAR_AMD_EMITTER_ID
#define UNGAR_AMD_EMITTER_ID "unversioned"
#endif
constexpr const char* kCacheFormat = "ungar_amd-cache-7";  // bump whenever the emitted kernels' argument list (their ABI) or the meta file changes: 3 = ten-argument kernels (knot stride, direct host results), 4 = consecutive outputs leave in 16-byte stores, 5 = a derivative may consist of several kernels (chunks of consecutive outputs; unit tags "jacobian.3"), 6 = fourteen-argument kernels (the parameters through an operand of their own), 7 = fifteen arguments (period of the parameter operand's instance index) + the resident single-instance kernel
constexpr std::size_t kBigKernel = 3000;        // statements above which the machine schedulers are switched off (see below)
// Statements above which a derivative is cut into chunks of consecutive outputs, one kernel and one compiler process each: the compile time of
// a straight-line body grows faster than its length (the equality-constraint Jacobian of the reference's quadruped OCP, 14 167 outputs in one
// body: 85 s of the example's 98 s cold start on the MI355X box), and the chunks compile side by side on the host's cores.
constexpr std::size_t kChunkStatements = 6000;
constexpr std::size_t kMaxChunks = 64;
constexpr int64_t kDirectHostInputs = 128;   // single-instance host calls: inputs up to this many doubles are read by the kernel from mapped host memory (no H2D copy command)
constexpr int64_t kDirectHostResults = 512;  // single-instance host calls: results up to this many doubles are written straight into mapped host memory

/// What a cache entry records besides the code objects: enough to serve every query and launch without the tape.
struct CacheMeta {
    std::string key;
    long long n = 0, p = 0, m = 0, enabled = 0;
    std::vector<int32_t> jacRows, jacCols, hesRows, hesCols;
    struct Unit {
        std::string tag, kernel;
        long long objectSize = 0, statements = 0;
    };
    std::vector<Unit> units;

    std::string Serialise() const {
        std::ostringstream os;
        os << kCacheFormat << "\nkey " << key << "\ndims " << n << ' ' << p << ' ' << m << ' ' << enabled << "\n";
        auto pattern = [&](const char* tag, const std::vector<int32_t>& r, const std::vector<int32_t>& c) {
            os << tag << ' ' << r.size();
            for (std::size_t k = 0; k < r.size(); ++k) os << ' ' << r[k] << ' ' << c[k];
            os << "\n";
        };
        pattern("jac", jacRows, jacCols);
        pattern("hes", hesRows, hesCols);
        os << "units " << units.size() << "\n";
        for (const Unit& u : units) os << "unit " << u.tag << ' ' << u.kernel << ' ' << u.objectSize << ' ' << u.statements << "\n";
        const std::string body = os.str();
        char sum[24];
        std::snprintf(sum, sizeof sum, "%016llx", static_cast<unsigned long long>(Fnv1a(body.data(), body.size(), 1469598103934665603ULL)));
        return body + "end " + sum + "\n";
    }

    /// Parses and validates (format line, key, trailing checksum); false = treat the entry as absent.
    bool Parse(const std::string& text, const std::string& expectKey) {
        const std::size_t endPos = text.rfind("end ");
        if (endPos == std::string::npos) return false;
        const std::string body = text.substr(0, endPos);
        char sum[24];
        std::snprintf(sum, sizeof sum, "%016llx", static_cast<unsigned long long>(Fnv1a(body.data(), body.size(), 1469598103934665603ULL)));
        if (text.compare(endPos + 4, 16, sum) != 0) return false;
        std::istringstream is(body);
        std::string word;
        if (!std::getline(is, word) || word != kCacheFormat) return false;
        if (!(is >> word >> key) || word != "key" || key != expectKey) return false;
        if (!(is >> word >> n >> p >> m >> enabled) || word != "dims") return false;
        auto pattern = [&](const char* tag, std::vector<int32_t>& r, std::vector<int32_t>& c) {
            std::size_t nnz = 0;
            if (!(is >> word >> nnz) || word != tag) return false;
            r.resize(nnz);
            c.resize(nnz);
            for (std::size_t k = 0; k < nnz; ++k)
                if (!(is >> r[k] >> c[k])) return false;
            return true;
        };
        if (!pattern("jac", jacRows, jacCols) || !pattern("hes", hesRows, hesCols)) return false;
        std::size_t count = 0;
        if (!(is >> word >> count) || word != "units" || count > 3 * kMaxChunks) return false;
        units.resize(count);
        for (Unit& u : units)
            if (!(is >> word >> u.tag >> u.kernel >> u.objectSize >> u.statements) || word != "unit") return false;
        return true;
    }
};

/// Emits one `extern "C" __global__` kernel: lane = instance, strided operands.
/// UNGAR_AMD_SCALAR_STORES=1 (part of the cache key): every output as its own 8-byte store -- for a platform whose compute queues are not in the unaligned access mode.
bool PairOutputStores() {
    const char* e = std::getenv("UNGAR_AMD_SCALAR_STORES");
    return !(e && e[0] == '1');
}

/// Outputs [first, last) of `values` (the whole derivative when the body is small enough, one chunk of it otherwise); output k is written at index k whatever the chunk.
/// serve: the code object also gets the resident single-instance kernel `<kernelName>_serve` (below).
std::string EmitKernel(const std::string& kernelName, const tape::Graph& g, int64_t nIndependent, int64_t nIn, const std::vector<tape::Id>& values, std::size_t first,
                       std::size_t last, std::size_t* statements, bool serve) {
    const bool pairs = PairOutputStores();
    std::vector<std::string> inNames;
    inNames.reserve(static_cast<std::size_t>(nIn));
    // inputs are spelled as loads at their uses (`in` derives from a __restrict__ parameter, so the compiler
    // merges repeated loads and places them where they are needed): reading every input into a local up
    // front keeps ~n values alive from the top of a whole-horizon kernel and made the register allocator
    // the dominant compile cost (objective value kernel, 960 inputs: 21 s -> 2 s, 1 KB of scratch -> none)
    // The parameters [n, n + p) come through an operand of their own (`par`): a caller whose parameters do not change with what it varies in the independent
    // variables -- the candidate steps of a line search over the same nodes -- keeps ONE image of them instead of a copy per candidate
    // (ungar_function_*_nodes_split); the one-operand entry points pass par = xp + n * xes with the same strides.
    for (int64_t i = 0; i < nIn; ++i)
        inNames.push_back(i < nIndependent ? "in[" + std::to_string(i) + " * xes]" : "par[" + std::to_string(i - nIndependent) + " * pes]");
    // Outputs are delivered in index order (Emitter::Emit), so consecutive outputs 2 m, 2 m + 1 leave as ONE 16-byte store where the output operand is
    // contiguous per instance (oes == 1: the node-major sparse Jacobians / Hessians the batched SQP assembles from).  One lane per instance writes its own
    // run of doubles: every store instruction is 64 separate transactions at L2 whatever its width, and their RATE bounds these kernels (with coalesced
    // unit-fastest outputs the quadruped's Jacobian kernels take 60 instead of 112 us) -- pairs halve the transactions.  The run of an instance starts at a
    // multiple of 8 bytes, not of 16: gfx950 under ROCm executes global_store_dwordx4 at any 4-byte-aligned address (SH_MEM_CONFIG alignment mode
    // "unaligned", what KFD programs for compute queues; tools/unaligned_store_probe.hip checks it on the box).  The vector TYPE is declared with
    // 8-byte alignment, so the store is defined C++ at every run start and it is the compiler, which knows the target's access mode
    // (amdhsa: unaligned-access-mode), that emits one dwordx4 store -- or two dwordx2 stores on a target without that mode.
    std::vector<tape::OutputSlot> slots;
    for (std::size_t k = first; k < last; ++k) {  // (chunks start at even k: the pairs below never straddle two kernels)
        const std::string ks = std::to_string(k);
        if (pairs && k % 2 == 0 && k + 1 < last) slots.push_back({values[k], "const double o" + ks + " = %s;"});
        else if (pairs && k % 2 == 1) slots.push_back({values[k], "UNGAR_STORE2(" + std::to_string(k - 1) + ", o" + std::to_string(k - 1) + ", %s);"});
        else slots.push_back({values[k], "out[" + ks + " * oes] = %s;"});
    }
    std::ostringstream os;
    os << "#ifndef UNGAR_STORE2\n"
          "typedef double ungar_d2 __attribute__((ext_vector_type(2), aligned(8)));\n"
          "#define UNGAR_STORE2(K, A, B) do { if (oes == 1) { ungar_d2 t_; t_.x = (A); t_.y = (B); *reinterpret_cast<ungar_d2*>(out + (K)) = t_; } "
          "else { out[(K) * oes] = (A); out[((K) + 1) * oes] = (B); } } while (0)\n"
          "#endif\n";
    // launched with 64-lane workgroups (LaunchFn): tell the compiler, so that it may use the full register file
    os << "extern \"C\" __global__ __launch_bounds__(64) void " << kernelName
       << "(const double* __restrict__ xp, long long xbs, long long xes, double* __restrict__ outBase, long long obs, long long oes, long long "
          "batch, long long knots, long long xks, long long oks, const double* __restrict__ pp, long long pbs, long long pes, long long pks, long long pmod) {\n"
       << "    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n"
       << "    if (i >= batch) return;\n"
       // node i = (instance i / knots, knot i % knots): the shooting nodes of a batch of horizons, each instance a strided run of knots
       << "    const long long ib = knots > 1 ? i / knots : i, ik = i - ib * knots;\n"
       << "    const double* __restrict__ in = xp + ib * xbs + ik * xks;\n"
       // (pmod > 0: the parameter operand holds pmod instances and instance ib reads number ib % pmod -- the candidates of a line search stacked instance-wise
       // over one image of the instances' parameters, ungar_function_forward_zero_nodes_periodic)
       << "    const double* __restrict__ par = pp + (pmod > 0 ? ib % pmod : ib) * pbs + ik * pks;\n"
       << "    double* __restrict__ out = outBase + ib * obs + ik * oks;\n";
    tape::Emitter em{g, inNames};
    const std::string body = em.Emit(slots);
    os << body << "}\n\n";
    if (statements) *statements = em.Stats().statements;
    if (serve) {
        // The RESIDENT form of the same body for single-instance calls from host memory (ungar_function_eval_host): ONE wavefront that stays on the device between
        // calls.  The host announces call t by writing t into *ring behind the inputs; the wavefront fetches the inputs with one request (lane i reads input i),
        // runs the body out of LDS in all 64 lanes alike -- the statements are scalar code: uniform LDS reads, the same cost as one lane -- writes the results with
        // coalesced stores and answers t in answers[0] behind them (system-scope acquire / release around the call).  Measured (tools/resident_pingpong.hip): the
        // round trip alone 5.0 us with the doorbell in host memory, 3.2 us with it in device memory the host writes through the PCIe aperture; one lane that
        // reads its inputs where the statements use them pays a bus round trip per use, +5 us on 37 inputs.  The wavefront returns when nobody has called for
        // `idle` ticks of the 100 MHz wall clock, or `life` ticks after it started (every device-wide synchronisation waits for it), and records the generation of
        // its launch in answers[1] so that the host knows to launch again.
        const std::size_t nOut = last - first;
        os << "extern \"C\" __global__ __launch_bounds__(64) void " << kernelName
           << "_serve(const double* hostIn, double* hostOut, unsigned long long* ring, unsigned long long* answers, unsigned long long served, unsigned long long generation, "
              "unsigned long long idle, unsigned long long life) {\n"
           << "    __shared__ __attribute__((aligned(16))) double arguments[" << std::max<int64_t>(nIn, 1) << "];\n"
           << "    __shared__ __attribute__((aligned(16))) double results[" << std::max<std::size_t>(nOut, 1) << "];\n"
           << "    const long long xes = 1, pes = 1, oes = 1;\n"
           << "    const int lane = threadIdx.x;\n"
           << "    if (blockIdx.x != 0) return;\n"
           << "    const unsigned long long born = wall_clock64();\n"
           << "    unsigned long long since = born;\n"
           << "    for (;;) {\n"
           << "        const unsigned long long asked = __shfl(__hip_atomic_load(ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), 0);  // (one address for the 64 lanes: one request)\n"
           << "        if (asked == served) {\n"
           << "            if (wall_clock64() - since > idle) break;\n"
           << "            continue;\n"
           << "        }\n"
           << "        __atomic_thread_fence(__ATOMIC_ACQUIRE);\n"
           << "        const double* from = hostIn;\n"
           << "        double* to = hostOut;\n"
           << "        asm volatile(\"\" : \"+v\"(from), \"+v\"(to) : : \"memory\");  // (this call's inputs: nothing read through an earlier copy of the pointer is reused)\n"
           << "        for (int i = lane; i < " << nIn << "; i += 64) arguments[i] = from[i];\n"
           << "        __syncthreads();\n"
           << "        {\n"
           << "            const double* __restrict__ in = arguments;\n"
           << "            const double* __restrict__ par = arguments + " << nIndependent << ";\n"
           << "            double* __restrict__ out = results;\n"
           << body
           << "        }\n"
           << "        __syncthreads();\n"
           << "        for (int k = lane; k < " << nOut << "; k += 64) to[k] = results[k];\n"
           << "        __atomic_thread_fence(__ATOMIC_RELEASE);\n"
           << "        __syncthreads();\n"
           << "        if (lane == 0) __hip_atomic_store(answers, asked, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n"
           << "        served = asked;\n"
           << "        since = wall_clock64();\n"
           << "        if (since - born > life) break;\n"
           << "    }\n"
           << "    if (lane == 0) __hip_atomic_store(answers + 1, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n"
           << "}\n\n";
    }
    return os.str();
}


}  // namespace

extern "C" {

int ungar_function_make(const ungar_tape_node* nodes, int64_t num_nodes, const int32_t* outputs, int64_t m, int64_t n, int64_t p,
                        const char* name, uint32_t enabled_derivatives, const char* folder, int recompile, ungar_function** out) {
    if (!nodes || !outputs || !name || !out || num_nodes <= 0 || m <= 0 || n < 0 || p < 0)
        return Fail(UNGAR_E_INVALID, "ungar_function_make: bad argument");
    // ---- rebuild (and thereby optimise) the tape --------------------------------------------------
    tape::Tape t;
    tape::Graph& g = t.graph;
    std::vector<tape::Id> map(static_cast<std::size_t>(num_nodes), tape::kNoId);
    for (int64_t i = 0; i < n + p; ++i) t.inputs.push_back(g.Input());
    auto ref = [&](int32_t id, int64_t self) -> tape::Id {
        if (id < 0 || id >= self) return tape::kNoId;
        return map[static_cast<std::size_t>(id)];
    };
    for (int64_t i = 0; i < num_nodes; ++i) {
        const ungar_tape_node& nd = nodes[i];
        if (nd.op < 0 || nd.op > static_cast<int32_t>(tape::Op::CondGt)) return Fail(UNGAR_E_INVALID, "ungar_function_make: unknown op code");
        const tape::Op op = static_cast<tape::Op>(nd.op);
        tape::Id r = tape::kNoId;
        const int ar = tape::Arity(op);
        const tape::Id a = ar >= 1 ? ref(nd.a, i) : tape::kNoId, b = ar >= 2 ? ref(nd.b, i) : tape::kNoId;
        const tape::Id c = ar == 4 ? ref(nd.c, i) : tape::kNoId, d = ar == 4 ? ref(nd.d, i) : tape::kNoId;
        if ((ar >= 1 && a == tape::kNoId) || (ar >= 2 && b == tape::kNoId) || (ar == 4 && (c == tape::kNoId || d == tape::kNoId)))
            return Fail(UNGAR_E_INVALID, "ungar_function_make: node " + std::to_string(i) + " references a later or invalid node");
        if (op == tape::Op::Const) r = g.Constant(nd.value);
        else if (op == tape::Op::Input) {
            if (nd.a < 0 || nd.a >= n + p) return Fail(UNGAR_E_INVALID, "ungar_function_make: input index out of range");
            r = t.inputs[static_cast<std::size_t>(nd.a)];
        } else if (ar == 1) r = g.Unary(op, a);
        else if (ar == 2) r = g.Binary(op, a, b);
        else r = g.Cond(op, a, b, c, d);
        map[static_cast<std::size_t>(i)] = r;
    }
    for (int64_t i = 0; i < m; ++i) {
        if (outputs[i] < 0 || outputs[i] >= num_nodes) return Fail(UNGAR_E_INVALID, "ungar_function_make: output index out of range");
        t.outputs.push_back(map[static_cast<std::size_t>(outputs[i])]);
    }
    if ((enabled_derivatives & kEnableHessian) && m != 1)
        return Fail(UNGAR_E_UNSUPPORTED, "The Hessian is implemented only for scalar functions.");  // function.hpp:136-137

    // names become file names and travel through the hipcc driver script, which re-expands $, ` and " in its arguments
    for (const char* text : {name, folder ? folder : ""})
        for (const char* c = text; *c; ++c)
            if (*c == '$' || *c == '`' || *c == '"' || *c == '\\' || *c == '\n' || (text == name && (*c == '/' || *c == '\'')))
                return Fail(UNGAR_E_INVALID, std::string("ungar_function_make: unsupported character '") + *c + "' in the function name or code-generation folder");

    // ---- cache key: the optimised tape and everything that decides what is generated from it --------------------
    const char* custom = std::getenv("UNGAR_AMD_JIT_FLAGS");
    KeyHasher key;
    key.Str(kCacheFormat);
    key.Str(UNGAR_AMD_EMITTER_ID);
    key.Str(kArch);
    key.Str(ToolchainVersion());  // the ROCm release whose compiler produces the code objects
    key.Str(custom ? custom : "");
    {
        const char* compiler = std::getenv("UNGAR_HIPCC");  // a different compiler (or a wrapper that adds flags) must not share entries
        key.Str(compiler ? compiler : "");
    }
    // accumulation mode of the Jacobian: 0 = cheaper of the two by the tape's cost estimate, 1 = forward, 2 = reverse (tests force each one)
    int jacobianMode = 0;
    if (const char* forced = std::getenv("UNGAR_AMD_JACOBIAN_MODE")) jacobianMode = std::atoi(forced) == 1 ? 1 : std::atoi(forced) == 2 ? 2 : 0;
    key.Int(jacobianMode);
    key.Int(PairOutputStores() ? 1 : 0);
    key.Int(n);
    key.Int(p);
    key.Int(m);
    key.Int(enabled_derivatives);
    key.Int(static_cast<long long>(g.Size()));
    for (const tape::Node& nd : g.Nodes()) {
        const std::int32_t head[5] = {static_cast<std::int32_t>(nd.op), nd.a, nd.b, nd.c, nd.d};
        key.Bytes(head, sizeof head);
        key.Bytes(&nd.value, sizeof nd.value);
    }
    for (tape::Id o : t.outputs) key.Int(o);
    const std::string keyHex = key.Hex();

    auto fn = std::make_unique<ungar_function>();
    fn->name = name;
    fn->tapeNodes.assign(nodes, nodes + num_nodes);
    fn->tapeOutputs.assign(outputs, outputs + m);
    fn->folder = folder ? folder : "";
    fn->n = n;
    fn->p = p;
    fn->m = m;
    fn->enabled = enabled_derivatives;
    const bool verbose = std::getenv("UNGAR_AMD_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();

    // ---- cache entry (layout mirrors <folder>/<name>/cppad_cg/<name>_lib.so, function.hpp:433-435) ------------------
    //   <dir>/<name>_<key>.meta           sparsity patterns + kernel table, written LAST (commit marker)
    //   <dir>/<name>_<key>_<tag>.hsaco    one code object per kernel (value / jacobian / hessian)
    const std::string dir = std::string(folder && *folder ? folder : DefaultFolder()) + "/" + fn->name + "/ungar_amd";
    const std::string base = dir + "/" + fn->name + "_" + keyHex;
    const std::string metaPath = base + ".meta";
    fn->codeObjectPath = base + "_value.hsaco";
    CacheMeta meta;
    auto lookup = [&] {  // a complete, self-consistent entry for this key?
        std::ifstream in(metaPath);
        if (!in) return false;
        std::stringstream ss;
        ss << in.rdbuf();
        bool ok = meta.Parse(ss.str(), keyHex) && meta.n == n && meta.p == p && meta.m == m && meta.enabled == static_cast<long long>(enabled_derivatives) &&
                  !meta.units.empty();
        for (std::size_t k = 0; ok && k < meta.units.size(); ++k) {
            long long size = 0;
            ok = FileSize(base + "_" + meta.units[k].tag + ".hsaco", &size) && size == meta.units[k].objectSize && size > 0;
        }
        return ok;
    };
    bool hit = !recompile && lookup();
    // Builders of the same entry are serialised by an advisory lock (released by the kernel if the holder dies): with one
    // process per GPU every rank asks for the same functions at start-up; the first one compiles, the others wait and
    // then hit -- no duplicated compile work, and never a meta file that describes another process's code objects.
    struct EntryLock {
        int fd = -1;
        ~EntryLock() {
            if (fd >= 0) {
                (void)flock(fd, LOCK_UN);
                (void)close(fd);
            }
        }
    } lock;
    if (!hit) {
        if (!MakeDirs(dir)) return Fail(UNGAR_E_IO, "cannot create code-generation folder '" + dir + "'");
        lock.fd = open((base + ".lock").c_str(), O_CREAT | O_RDWR, 0666);
        if (lock.fd >= 0) (void)flock(lock.fd, LOCK_EX);
        if (!recompile) hit = lookup();  // published by another builder while this one waited
    }

    if (!hit) {
        // ---- derivatives (parameters trimmed: columns [0, n) only; function.hpp:529-574) -------------
        meta = CacheMeta{};
        meta.key = keyHex;
        meta.n = n;
        meta.p = p;
        meta.m = m;
        meta.enabled = enabled_derivatives;
        const std::vector<tape::Id> valueIds = t.outputs;
        tape::SparseEntries jac, hes;
        tape::Differentiator diff{t};
        if (enabled_derivatives & kEnableJacobian) {
            jac = diff.Jacobian(static_cast<int>(n), jacobianMode);
            meta.jacRows.assign(jac.row.begin(), jac.row.end());
            meta.jacCols.assign(jac.col.begin(), jac.col.end());
        }
        if (enabled_derivatives & kEnableHessian) {
            hes = diff.Hessian(0, static_cast<int>(n));
            meta.hesRows.assign(hes.row.begin(), hes.row.end());
            meta.hesCols.assign(hes.col.begin(), hes.col.end());
        }

        // ---- HIP source: one translation unit per kernel so that the compiler runs on all of them at once ----
        struct Unit {
            std::string kernel, tag;  // chunk c > 0 of a derivative: kernel "<name>_c<c>", tag "<tag>.<c>"
            const std::vector<tape::Id>* values = nullptr;
            std::size_t first = 0, last = 0;  // outputs [first, last) of *values
            std::string source, object, tmpObject, log, flags;
            FILE* pipe = nullptr;
            std::size_t statements = 0;
        };
        std::vector<Unit> units;
        // One unit per derivative, or one per chunk of consecutive outputs whose cone (distinct tape nodes it needs) stays below kChunkStatements: greedy over the
        // outputs in index order, chunk boundaries at even indices; what two chunks share is computed by both.
        auto addUnits = [&](const char* kernel, const char* tag, const std::vector<tape::Id>& values) {
            std::vector<std::size_t> starts{0};
            {
                std::vector<int> seenIn(g.Size(), -1);
                std::size_t inChunk = 0, inAllChunks = 0;
                std::vector<char> seenAtAll(g.Size(), 0);
                std::size_t distinct = 0;
                std::vector<tape::Id> stack;
                for (std::size_t k = 0; k < values.size(); ++k) {
                    const int chunk = static_cast<int>(starts.size()) - 1;
                    stack.push_back(values[k]);
                    while (!stack.empty()) {
                        const tape::Id id = stack.back();
                        stack.pop_back();
                        if (id == tape::kNoId || seenIn[static_cast<std::size_t>(id)] == chunk) continue;
                        seenIn[static_cast<std::size_t>(id)] = chunk;
                        const tape::Node& nd = g.At(id);
                        if (tape::Arity(nd.op) == 0) continue;
                        ++inChunk;
                        ++inAllChunks;
                        if (!seenAtAll[static_cast<std::size_t>(id)]) {
                            seenAtAll[static_cast<std::size_t>(id)] = 1;
                            ++distinct;
                        }
                        for (tape::Id o : {nd.a, nd.b, nd.c, nd.d}) stack.push_back(o);
                    }
                    if (inChunk > kChunkStatements && k % 2 == 1 && k + 1 < values.size() && starts.size() < kMaxChunks) {
                        starts.push_back(k + 1);
                        inChunk = 0;
                    }
                }
                // What two chunks share is computed by both.  Outputs with little in common (the block rows of a whole-horizon constraint Jacobian) lose almost
                // nothing; a dense Jacobian whose every entry hangs on one long primal (forward dynamics taped through ABA: 25 s as one body, 103 s as 20 chunks
                // that each repeat the primal) is left in one piece.
                if (inAllChunks > distinct + distinct / 4) starts.assign(1, 0);
            }
            for (std::size_t c = 0; c < starts.size(); ++c) {
                Unit u;
                u.kernel = c ? std::string(kernel) + "_c" + std::to_string(c) : std::string(kernel);
                u.tag = c ? std::string(tag) + "." + std::to_string(c) : std::string(tag);
                u.values = &values;
                u.first = starts[c];
                u.last = c + 1 < starts.size() ? starts[c + 1] : values.size();
                units.push_back(std::move(u));
            }
        };
        addUnits("ungar_fn_forward_zero", "value", valueIds);
        if (!jac.value.empty()) addUnits("ungar_fn_sparse_jacobian", "jacobian", jac.value);
        if (!hes.value.empty()) addUnits("ungar_fn_sparse_hessian", "hessian", hes.value);
        if (!MakeDirs(dir)) return Fail(UNGAR_E_IO, "cannot create code-generation folder '" + dir + "'");
        // Compile flags.  The machine instruction schedulers (pre- and post-RA) account for > 95 % of the compile
        // time of a large straight-line kernel (a whole-horizon constraint Jacobian of 23 k statements: 64 s -> 6 s
        // without them) and the emitter already orders statements depth-first, so they are switched off above
        // kBigKernel statements; node-sized functions -- the ones evaluated in large batches -- keep the full
        // pipeline.  UNGAR_AMD_JIT_FLAGS replaces the optimisation flags altogether.
        const char* hipcc = std::getenv("UNGAR_HIPCC");
        const std::string unique = "." + std::to_string(getpid()) + "." + std::to_string(reinterpret_cast<std::uintptr_t>(fn.get()) & 0xFFFFFF) + ".tmp";
        auto cleanup = [&units] {  // every error path: reap the compilers that were started, remove what they wrote
            for (Unit& u : units) {
                if (u.pipe) {
                    char buf[512];
                    while (fgets(buf, sizeof buf, u.pipe)) {
                    }
                    (void)pclose(u.pipe);
                    u.pipe = nullptr;
                }
                if (!u.source.empty()) (void)std::remove(u.source.c_str());
                if (!u.tmpObject.empty()) (void)std::remove(u.tmpObject.c_str());
            }
        };
        // Compilers run concurrently, in WAVES of at most UNGAR_AMD_JIT_JOBS (default: the host's hardware threads): a function whose three derivatives are cut into
        // 64 chunks each would otherwise start 192 hipcc processes at once -- well over a gigabyte each at -O3 -- and an out-of-memory kill reads as a compile error.
        std::size_t maxJobs = std::max(1u, std::thread::hardware_concurrency());
        if (const char* jobs = std::getenv("UNGAR_AMD_JIT_JOBS")) maxJobs = static_cast<std::size_t>(std::max(1L, std::atol(jobs)));
        std::string failure;
        auto reap = [&](Unit& u) {
            if (!u.pipe) return;
            char buf[512];
            while (fgets(buf, sizeof buf, u.pipe)) u.log += buf;
            const int st = pclose(u.pipe);
            u.pipe = nullptr;
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : -1;
            if (rc != 0) failure += "hipcc failed (" + std::to_string(rc) + ") for the " + u.tag + " kernel of function '" + fn->name + "':\n" + u.log;
        };
        std::size_t started = 0, reaped = 0;
        for (Unit& u : units) {
            if (started - reaped >= maxJobs) reap(units[reaped++]);  // the oldest running compiler finishes before the next one starts
            ++started;
            const std::string src = "// generated by ungar_amd (runtime/function.cpp) for function '" + fn->name + "'\n#include <hip/hip_runtime.h>\n\n" +
                                    EmitKernel(u.kernel, g, n, n + p, *u.values, u.first, u.last, &u.statements,
                                               /* a whole derivative whose operands the single-instance host call keeps in mapped host memory */
                                               u.first == 0 && u.last == u.values->size() && n + p <= kDirectHostInputs && static_cast<int64_t>(u.values->size()) <= kDirectHostResults);
            u.flags = std::string("--offload-arch=") + kArch + " -std=c++17 " +
                      (custom ? custom : u.statements > kBigKernel ? "-O3 -mllvm -enable-misched=false -mllvm -enable-post-misched=false" : "-O3");
            u.object = base + "_" + u.tag + ".hsaco";
            // process-unique names for BOTH the source and the object: ranks that build the same function at start-up
            // never read or truncate each other's files; the rename below is the atomic publish (function.hpp:485-487, 501-502)
            u.source = base + "_" + u.tag + unique + ".hip";
            u.tmpObject = u.object + unique;
            {
                std::ofstream f(u.source);
                f << src;
                if (!f) {
                    cleanup();
                    return Fail(UNGAR_E_IO, "cannot write '" + u.source + "'");
                }
            }
            const std::string cmd = std::string(hipcc ? hipcc : "hipcc") + " " + u.flags + " --genco -o " + ShellQuote(u.tmpObject) + " " + ShellQuote(u.source) + " 2>&1";
            u.pipe = popen(cmd.c_str(), "r");
            if (!u.pipe) {
                cleanup();
                return Fail(UNGAR_E_COMPILE, "cannot start hipcc for function '" + fn->name + "'");
            }
        }
        for (Unit& u : units) reap(u);
        if (!failure.empty()) {
            cleanup();
            return Fail(UNGAR_E_COMPILE, failure);
        }
        const bool keepSource = std::getenv("UNGAR_AMD_KEEP_SOURCE") != nullptr;
        for (Unit& u : units) {
            long long size = 0;
            if (!FileSize(u.tmpObject, &size) || size <= 0 || std::rename(u.tmpObject.c_str(), u.object.c_str()) != 0) {
                cleanup();
                return Fail(UNGAR_E_IO, "cannot publish '" + u.object + "'");
            }
            u.tmpObject.clear();
            if (keepSource) (void)std::rename(u.source.c_str(), (base + "_" + u.tag + ".hip").c_str());
            else (void)std::remove(u.source.c_str());
            u.source.clear();
            meta.units.push_back({u.tag, u.kernel, size, static_cast<long long>(u.statements)});
        }
        {  // the meta file commits the entry: temp + rename, after every code object is in place
            const std::string tmpMeta = metaPath + unique;
            std::ofstream f(tmpMeta);
            f << meta.Serialise();
            f.close();
            if (!f || std::rename(tmpMeta.c_str(), metaPath.c_str()) != 0) {
                (void)std::remove(tmpMeta.c_str());
                return Fail(UNGAR_E_IO, "cannot publish '" + metaPath + "'");
            }
        }
    }
    fn->cacheHit = hit;
    fn->jacRows = meta.jacRows;
    fn->jacCols = meta.jacCols;
    fn->hesRows = meta.hesRows;
    fn->hesCols = meta.hesCols;
    if (verbose)
        std::fprintf(stderr, "[ungar_amd] function '%s': %lld tape nodes, n=%lld p=%lld m=%lld, jac nnz %zu, hes nnz %zu, %s in %.2f s (key %s)\n",
                     fn->name.c_str(), static_cast<long long>(num_nodes), static_cast<long long>(n), static_cast<long long>(p), static_cast<long long>(m),
                     fn->jacRows.size(), fn->hesRows.size(), hit ? "cache hit: derive / emit / compile skipped" : "derived, emitted and compiled",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), keyHex.c_str());

    // ---- load ---------------------------------------------------------------------------------------
    // UNGAR_AMD_COMPILE_ONLY: stop after publishing the cache entry (cache warm-up on a machine without a
    // GPU, e.g. a build host); the returned function reports its sparsity but every evaluation fails.
    if (std::getenv("UNGAR_AMD_COMPILE_ONLY")) {
        *out = fn.release();
        return UNGAR_OK;
    }
    for (std::size_t k = 0; k < meta.units.size(); ++k) {  // (the chunks of a derivative are listed in output order)
        const CacheMeta::Unit& u = meta.units[k];
        const std::string object = base + "_" + u.tag + ".hsaco";
        const std::string derivative = u.tag.substr(0, u.tag.find('.'));
        ungar_function::Kernels& handles = derivative == "value" ? fn->kValue : derivative == "jacobian" ? fn->kJac : fn->kHes;
        hipModule_t module = nullptr;
        hipError_t e = hipModuleLoad(&module, object.c_str());
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("hipModuleLoad('") + object + "'): " + hipGetErrorString(e));
        fn->modules.push_back(module);
        hipFunction_t handle = nullptr;
        e = hipModuleGetFunction(&handle, module, u.kernel.c_str());
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("hipModuleGetFunction: ") + hipGetErrorString(e));
        handles.push_back(handle);
        if (u.tag == derivative && n + p <= kDirectHostInputs) {  // (one kernel for the whole derivative: it may come with its resident form)
            hipFunction_t serve = nullptr;
            if (hipModuleGetFunction(&serve, module, (u.kernel + "_serve").c_str()) == hipSuccess) fn->resident[derivative == "value" ? 0 : derivative == "jacobian" ? 1 : 2].kernel = serve;
            else (void)hipGetLastError();
        }
    }
    *out = fn.release();
    return UNGAR_OK;
}

void ungar_function_free(ungar_function* fn) {
    delete fn;
}

int ungar_function_get_info(const ungar_function* fn, ungar_function_info* info) {
    if (!fn || !info) return Fail(UNGAR_E_INVALID, "ungar_function_get_info: null argument");
    info->n = fn->n;
    info->p = fn->p;
    info->m = fn->m;
    info->jac_nnz = static_cast<int64_t>(fn->jacRows.size());
    info->hes_nnz = static_cast<int64_t>(fn->hesRows.size());
    info->cache_hit = fn->cacheHit ? 1 : 0;
    return UNGAR_OK;
}

int ungar_function_get_tape(const ungar_function* fn, const ungar_tape_node** nodes, int64_t* num_nodes, const int32_t** outputs, const char** folder) {
    if (!fn || !nodes || !num_nodes || !outputs) return Fail(UNGAR_E_INVALID, "ungar_function_get_tape: null argument");
    *nodes = fn->tapeNodes.data();
    *num_nodes = static_cast<int64_t>(fn->tapeNodes.size());
    *outputs = fn->tapeOutputs.data();
    if (folder) *folder = fn->folder.c_str();
    return UNGAR_OK;
}

const char* ungar_function_code_object(const ungar_function* fn) {
    return fn ? fn->codeObjectPath.c_str() : "";
}

int ungar_function_jacobian_sparsity(const ungar_function* fn, const int32_t** rows, const int32_t** cols, int64_t* nnz) {
    if (!fn || !rows || !cols || !nnz) return Fail(UNGAR_E_INVALID, "ungar_function_jacobian_sparsity: null argument");
    if (fn->kJac.empty() && !(fn->enabled & kEnableJacobian)) return Fail(UNGAR_E_UNSUPPORTED, "function '" + fn->name + "' was made without JACOBIAN");
    *rows = fn->jacRows.data();
    *cols = fn->jacCols.data();
    *nnz = static_cast<int64_t>(fn->jacRows.size());
    return UNGAR_OK;
}

int ungar_function_hessian_sparsity(const ungar_function* fn, const int32_t** rows, const int32_t** cols, int64_t* nnz) {
    if (!fn || !rows || !cols || !nnz) return Fail(UNGAR_E_INVALID, "ungar_function_hessian_sparsity: null argument");
    if (!(fn->enabled & kEnableHessian)) return Fail(UNGAR_E_UNSUPPORTED, "function '" + fn->name + "' was made without HESSIAN");
    *rows = fn->hesRows.data();
    *cols = fn->hesCols.data();
    *nnz = static_cast<int64_t>(fn->hesRows.size());
    return UNGAR_OK;
}

static int LaunchFn(const ungar_function* fn, const ungar_function::Kernels* kernels, const char* what, const ungar_operand* xp, const ungar_operand* out,
                    int64_t batch, void* stream, int64_t knotsPerInstance = 1, const ungar_operand* parameters = nullptr, int64_t parameterPeriod = 0) {
    if (!fn || !xp || !out) return Fail(UNGAR_E_INVALID, std::string(what) + ": null argument");
    if (!kernels || kernels->empty()) return Fail(UNGAR_E_UNSUPPORTED, std::string(what) + ": kernel not available for function '" + fn->name + "' (derivative not enabled, or made with UNGAR_AMD_COMPILE_ONLY)");
    if (batch < 0 || knotsPerInstance < 1 || parameterPeriod < 0) return Fail(UNGAR_E_INVALID, std::string(what) + ": negative batch, knots < 1 or a negative period");
    if (batch == 0) return UNGAR_OK;
    if (!xp->base || !out->base) return Fail(UNGAR_E_INVALID, std::string(what) + ": null operand base");
    const double* in = xp->base;
    long long xbs = xp->instance_stride, xes = xp->element_stride, obs = out->instance_stride, oes = out->element_stride, b = batch;
    double* o = out->base;
    long long knots = knotsPerInstance, xks = xp->knot_stride, oks = out->knot_stride;
    // parameters: their own operand, or (one-operand entry points) the elements [n, n + p) of xp
    if (parameters && fn->p > 0 && !parameters->base) return Fail(UNGAR_E_INVALID, std::string(what) + ": null parameter operand");
    const double* par = parameters ? parameters->base : xp->base + fn->n * xp->element_stride;
    long long pbs = parameters ? parameters->instance_stride : xp->instance_stride, pes = parameters ? parameters->element_stride : xp->element_stride,
              pks = parameters ? parameters->knot_stride : xp->knot_stride;
    long long pmod = parameterPeriod;
    void* args[] = {&in, &xbs, &xes, &o, &obs, &oes, &b, &knots, &xks, &oks, &par, &pbs, &pes, &pks, &pmod};
    const unsigned block = 64;
    for (hipFunction_t k : *kernels) {  // every chunk of the derivative writes its own outputs of the same operand
        const hipError_t e = hipModuleLaunchKernel(k, static_cast<unsigned>((batch + block - 1) / block), 1, 1, block, 1, 1, 0, static_cast<hipStream_t>(stream), args, nullptr);
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    return UNGAR_OK;
}

int ungar_function_forward_zero(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* y, int64_t batch, void* stream) {
    return LaunchFn(fn, fn ? &fn->kValue : nullptr, "ungar_function_forward_zero", xp, y, batch, stream);
}
// A derivative that is enabled but structurally empty (Jacobian of a function of the parameters only, Hessian of a
// linear objective) has no kernel and nothing to write: the reference returns an empty sparse matrix there
// (function.hpp:216-230, 236-259), so the batched calls succeed without a launch.
int ungar_function_sparse_jacobian(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* jac, int64_t batch, void* stream) {
    if (fn && xp && jac && batch >= 0 && (fn->enabled & kEnableJacobian) && fn->jacRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? &fn->kJac : nullptr, "ungar_function_sparse_jacobian", xp, jac, batch, stream);
}
int ungar_function_sparse_hessian(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* hes, int64_t batch, void* stream) {
    if (fn && xp && hes && batch >= 0 && (fn->enabled & kEnableHessian) && fn->hesRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? &fn->kHes : nullptr, "ungar_function_sparse_hessian", xp, hes, batch, stream);
}

// Shooting nodes: node i = (instance i / knots, knot i % knots) at base + instance * instance_stride + knot * knot_stride.
int ungar_function_forward_zero_nodes(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* y, int64_t count, int64_t knots, void* stream) {
    return LaunchFn(fn, fn ? &fn->kValue : nullptr, "ungar_function_forward_zero_nodes", xp, y, count, stream, knots);
}
int ungar_function_sparse_jacobian_nodes(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* jac, int64_t count, int64_t knots, void* stream) {
    if (fn && xp && jac && count >= 0 && (fn->enabled & kEnableJacobian) && fn->jacRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? &fn->kJac : nullptr, "ungar_function_sparse_jacobian_nodes", xp, jac, count, stream, knots);
}
int ungar_function_sparse_hessian_nodes(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* hes, int64_t count, int64_t knots, void* stream) {
    if (fn && xp && hes && count >= 0 && (fn->enabled & kEnableHessian) && fn->hesRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? &fn->kHes : nullptr, "ungar_function_sparse_hessian_nodes", xp, hes, count, stream, knots);
}

// The independent variables [0, n) and the parameters [n, n + p) through separate operands (same node decomposition for both: a parameter operand with instance
// stride 0 serves every "instance" -- e.g. every candidate step of a line search over the same nodes -- from one image).
int ungar_function_forward_zero_nodes_split(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, const ungar_operand* y, int64_t count, int64_t knots, void* stream) {
    if (!p) return Fail(UNGAR_E_INVALID, "ungar_function_forward_zero_nodes_split: null argument");
    return LaunchFn(fn, fn ? &fn->kValue : nullptr, "ungar_function_forward_zero_nodes_split", x, y, count, stream, knots, p);
}
int ungar_function_forward_zero_nodes_periodic(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, int64_t parameter_instances, const ungar_operand* y,
                                               int64_t count, int64_t knots, void* stream) {
    if (!p || parameter_instances < 1) return Fail(UNGAR_E_INVALID, "ungar_function_forward_zero_nodes_periodic: null parameter operand or fewer than one parameter instance");
    return LaunchFn(fn, fn ? &fn->kValue : nullptr, "ungar_function_forward_zero_nodes_periodic", x, y, count, stream, knots, p, parameter_instances);
}
int ungar_function_sparse_jacobian_nodes_split(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, const ungar_operand* jac, int64_t count, int64_t knots,
                                               void* stream) {
    if (!p) return Fail(UNGAR_E_INVALID, "ungar_function_sparse_jacobian_nodes_split: null argument");
    if (fn && x && jac && count >= 0 && (fn->enabled & kEnableJacobian) && fn->jacRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? &fn->kJac : nullptr, "ungar_function_sparse_jacobian_nodes_split", x, jac, count, stream, knots, p);
}
int ungar_function_sparse_hessian_nodes_split(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, const ungar_operand* hes, int64_t count, int64_t knots,
                                              void* stream) {
    if (!p) return Fail(UNGAR_E_INVALID, "ungar_function_sparse_hessian_nodes_split: null argument");
    if (fn && x && hes && count >= 0 && (fn->enabled & kEnableHessian) && fn->hesRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? &fn->kHes : nullptr, "ungar_function_sparse_hessian_nodes_split", x, hes, count, stream, knots, p);
}

int32_t ungar_function_host_call_resident(const ungar_function* fn, int32_t what) {
    if (!fn || what < 0 || what > 2 || !fn->resident[what].kernel) return 0;
    const char* v = std::getenv("UNGAR_AMD_HOST_CALL_RESIDENT_US");
    if (v && std::atol(v) <= 0) return 0;
    const int64_t nOut = what == 0 ? fn->m : what == 1 ? static_cast<int64_t>(fn->jacRows.size()) : static_cast<int64_t>(fn->hesRows.size());
    return fn->n + fn->p <= kDirectHostInputs && nOut > 0 && nOut <= kDirectHostResults ? 1 : 0;
}

/// Single-instance host call (ungar_amd.h): the resident kernel for node-sized functions, one launch on a private stream otherwise.
int ungar_function_eval_host(ungar_function* fn, int32_t what, const double* xp_host, double* out_host) {
    if (!fn || !xp_host) return Fail(UNGAR_E_INVALID, "ungar_function_eval_host: null argument");
    const int64_t nIn = fn->n + fn->p;
    const int64_t nOut = what == 0 ? fn->m : what == 1 ? static_cast<int64_t>(fn->jacRows.size()) : static_cast<int64_t>(fn->hesRows.size());
    const ungar_function::Kernels* k = what == 0 ? &fn->kValue : what == 1 ? &fn->kJac : &fn->kHes;
    if (what < 0 || what > 2) return Fail(UNGAR_E_INVALID, "ungar_function_eval_host: what must be 0 (value), 1 (Jacobian) or 2 (Hessian)");
    if (nOut == 0 && what != 0 && (fn->enabled & (what == 1 ? kEnableJacobian : kEnableHessian))) return UNGAR_OK;  // enabled but structurally empty (out_host may be null)
    if (!out_host) return Fail(UNGAR_E_INVALID, "ungar_function_eval_host: null output");
    if (k->empty()) return Fail(UNGAR_E_UNSUPPORTED, "ungar_function_eval_host: kernel not available for function '" + fn->name + "' (derivative not enabled, or made with UNGAR_AMD_COMPILE_ONLY)");
    if (nOut == 0) return UNGAR_OK;
    // One launch and ONE synchronisation per call: the inputs go through a pinned staging buffer (asynchronous copy on the function's own
    // stream), the kernel writes its results straight into mapped host memory (posted writes over PCIe; nothing to copy back), and the stream
    // is awaited once.  (Three blocking steps -- copy in, launch, copy out -- measured 27 us per call; the reference's in-process C call has no
    // such floor, so this is what a drop-in user of the single-instance API sees first.)  UNGAR_AMD_HOST_CALL_COPIES=1 restores the copies.
    static const bool copies = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_HOST_CALL_COPIES") != nullptr;
    hipError_t e = hipSuccess;
    if (!fn->dIn) e = hipMalloc(&fn->dIn, static_cast<std::size_t>(std::max<int64_t>(nIn, 1)) * sizeof(double));
    if (copies) {
        if (e == hipSuccess && fn->dOutSize < nOut) {
            if (fn->dOut) (void)hipFree(fn->dOut);
            e = hipMalloc(&fn->dOut, static_cast<std::size_t>(nOut) * sizeof(double));
            fn->dOutSize = nOut;
        }
        if (e == hipSuccess && nIn > 0) e = hipMemcpy(fn->dIn, xp_host, static_cast<std::size_t>(nIn) * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
        const ungar_operand in{fn->dIn, nIn, 0, 1}, outOp{fn->dOut, nOut, 0, 1};
        const int rc = LaunchFn(fn, k, "ungar_function_eval_host", &in, &outOp, 1, nullptr);
        if (rc != UNGAR_OK) return rc;
        e = hipMemcpy(out_host, fn->dOut, static_cast<std::size_t>(nOut) * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
        return UNGAR_OK;
    }
    if (e == hipSuccess && !fn->hostStream) e = hipStreamCreateWithFlags(&fn->hostStream, hipStreamNonBlocking);
    if (e == hipSuccess && !fn->hIn) {
        e = hipHostMalloc(reinterpret_cast<void**>(&fn->hIn), static_cast<std::size_t>(std::max<int64_t>(nIn, 1)) * sizeof(double), hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&fn->hInDevice), fn->hIn, 0);
    }
    // Completion: a 64-bit word in mapped host memory that the STREAM writes behind the kernel (hipStreamWriteValue64) and the host polls -- no
    // hipStreamSynchronize (an interrupt-driven wait costs more than the launch).  A runtime without stream memory operations falls back to the wait.
    static std::atomic<bool> streamWrites{UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_HOST_CALL_SYNCHRONIZE") == nullptr};
    if (e == hipSuccess && !fn->hFlag && streamWrites.load(std::memory_order_relaxed)) {
        e = hipHostMalloc(reinterpret_cast<void**>(&fn->hFlag), 64, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&fn->hFlagDevice), fn->hFlag, 0);
        if (e == hipSuccess) *fn->hFlag = 0;
    }
    if (e == hipSuccess && fn->hOutSize < nOut) {
        // sized once for the largest of the three results: a resident kernel of another derivative keeps the address it was launched with
        const int64_t largest = std::max<int64_t>({fn->m, static_cast<int64_t>(fn->jacRows.size()), static_cast<int64_t>(fn->hesRows.size()), nOut});
        for (ungar_function::Resident& r : fn->resident)
            if (r.stream && fn->hOut) (void)hipStreamSynchronize(r.stream);
        if (fn->hOut) (void)hipHostFree(fn->hOut);
        fn->hOut = nullptr;
        e = hipHostMalloc(reinterpret_cast<void**>(&fn->hOut), static_cast<std::size_t>(largest) * sizeof(double), hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&fn->hOutDevice), fn->hOut, 0);
        fn->hOutSize = e == hipSuccess ? largest : 0;
    }
    // small inputs: the kernel's one lane reads them from the mapped buffer itself (a handful of bus reads in flight at once); large ones (a whole-horizon
    // function: hundreds of doubles, one dependent bus read each would dominate) go to device memory by one asynchronous copy on the same stream
    const bool directIn = nIn <= kDirectHostInputs && fn->hInDevice;
    if (e == hipSuccess && nIn > 0) {
        std::memcpy(fn->hIn, xp_host, static_cast<std::size_t>(nIn) * sizeof(double));
        if (!directIn) e = hipMemcpyAsync(fn->dIn, fn->hIn, static_cast<std::size_t>(nIn) * sizeof(double), hipMemcpyHostToDevice, fn->hostStream);
    }
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
    // small results are written by the kernel into the mapped buffer itself; large ones (a whole-horizon Jacobian: thousands of 8-byte stores of
    // one lane, each its own PCIe write) go to device memory and come back as one asynchronous copy on the same stream
    const bool direct = nOut <= kDirectHostResults;
    // ---- node-sized functions: the RESIDENT kernel (EmitKernel).  A call is a ticket written into mapped host memory and an answer polled from it -- two bus
    // crossings and the body, no launch, no HIP call at all while the kernel is there.  It is launched on the first call and again whenever the previous launch
    // has returned (nobody called for UNGAR_AMD_HOST_CALL_RESIDENT_US microseconds -- default 200, 0: never resident -- or it reached its 50 ms life).
    static const long residentUs = [] {
        const char* v = std::getenv("UNGAR_AMD_HOST_CALL_RESIDENT_US");
        return v ? std::max(0L, std::atol(v)) : 100L;
    }();
    ungar_function::Resident& r = fn->resident[what];
    if (r.kernel && residentUs > 0 && directIn && direct) {
        if (!fn->residentReady) {  // once per function: where the doorbells and the inputs live
            int device = 0, largeBar = 0;
            const bool hostMemoryOnly = UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_HOST_CALL_NO_APERTURE") != nullptr;  // (the route of a device without a large BAR, for the tests)
            if (!hostMemoryOnly && hipGetDevice(&device) == hipSuccess && hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && largeBar == 1) {
                if (hipExtMallocWithFlags(&fn->aperture, 4096, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(fn->aperture, 0, 4096) != hipSuccess ||
                    hipDeviceSynchronize() != hipSuccess) {
                    (void)hipGetLastError();
                    if (fn->aperture) (void)hipFree(fn->aperture);
                    fn->aperture = nullptr;
                }
            } else {
                (void)hipGetLastError();
            }
            if (!fn->aperture) {
                e = hipHostMalloc(reinterpret_cast<void**>(&fn->rings), 256, hipHostMallocMapped);
                if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&fn->ringsDevice), fn->rings, 0);
                if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
                std::memset(fn->rings, 0, 256);
            }
            fn->residentReady = true;
        }
        if (!r.answers) {
            e = hipHostMalloc(reinterpret_cast<void**>(&r.answers), 64, hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&r.answersDevice), r.answers, 0);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking);
            if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
            r.answers[0] = r.answers[1] = 0;
        }
        volatile unsigned long long* answers = r.answers;
        const unsigned long long ticket = ++r.tickets;
        unsigned long long* ring = fn->aperture ? reinterpret_cast<unsigned long long*>(static_cast<char*>(fn->aperture) + 2048 + 64 * what) : fn->rings + 8 * what;
        if (fn->aperture) {  // write-combined stores: the inputs are pushed out before the doorbell, the doorbell at once
            std::memcpy(fn->aperture, xp_host, static_cast<std::size_t>(nIn) * sizeof(double));
            _mm_sfence();
            *static_cast<volatile unsigned long long*>(ring) = ticket;
            _mm_sfence();
        } else {
            __atomic_store_n(ring, ticket, __ATOMIC_RELEASE);  // (the inputs were written into the mapped buffer above)
        }
        auto launch = [&]() -> hipError_t {
            const double* in = fn->aperture ? static_cast<const double*>(fn->aperture) : fn->hInDevice;
            double* outDevice = fn->hOutDevice;
            unsigned long long* ringDevice = fn->aperture ? ring : fn->ringsDevice + 8 * what;
            unsigned long long* answersDevice = r.answersDevice;
            unsigned long long served = ticket - 1, generation = ++r.generation, idle = static_cast<unsigned long long>(residentUs) * 100ull, life = 5000000ull;  // 100 MHz
            void* args[] = {&in, &outDevice, &ringDevice, &answersDevice, &served, &generation, &idle, &life};
            return hipModuleLaunchKernel(r.kernel, 1, 1, 1, 64, 1, 1, 0, r.stream, args, nullptr);
        };
        if (r.generation == 0 || answers[1] == r.generation) e = launch();
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
        unsigned spins = 0;
        while (e == hipSuccess && answers[0] != ticket) {
            if (answers[1] == r.generation) {  // the launch returned: before this ticket was seen (launch again), or right behind its answer
                std::atomic_thread_fence(std::memory_order_acquire);
                if (answers[0] == ticket) break;
                e = launch();
            } else if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() > deadline) {
                r.kernel = nullptr;  // (the launches of the stream path from now on)
                return Fail(UNGAR_E_HIP, "ungar_function_eval_host: the resident kernel of function '" + fn->name + "' did not answer within 2 s");
            }
        }
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(out_host, fn->hOut, static_cast<std::size_t>(nOut) * sizeof(double));
        return UNGAR_OK;
    }
    if (!direct && fn->dOutSize < nOut) {
        if (fn->dOut) (void)hipFree(fn->dOut);
        e = hipMalloc(&fn->dOut, static_cast<std::size_t>(nOut) * sizeof(double));
        fn->dOutSize = e == hipSuccess ? nOut : 0;
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
    }
    const ungar_operand in{directIn ? fn->hInDevice : fn->dIn, nIn, 0, 1}, outOp{direct ? fn->hOutDevice : fn->dOut, nOut, 0, 1};
    const int rc = LaunchFn(fn, k, "ungar_function_eval_host", &in, &outOp, 1, fn->hostStream);
    if (rc != UNGAR_OK) return rc;
    if (!direct) e = hipMemcpyAsync(fn->hOut, fn->dOut, static_cast<std::size_t>(nOut) * sizeof(double), hipMemcpyDeviceToHost, fn->hostStream);
    bool polled = false;
    if (e == hipSuccess && fn->hFlag && streamWrites.load(std::memory_order_relaxed)) {
        const unsigned long long ticket = ++fn->hostCalls;
        if (hipStreamWriteValue64(fn->hostStream, fn->hFlagDevice, ticket, 0) == hipSuccess) {
            volatile unsigned long long* flag = fn->hFlag;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
            unsigned spins = 0;
            while (*flag != ticket) {
                if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() > deadline) break;  // a stuck queue: let the runtime report it through the wait below
            }
            polled = *flag == ticket;
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            (void)hipGetLastError();
            streamWrites.store(false, std::memory_order_relaxed);
        }
    }
    if (e == hipSuccess && !polled) e = hipStreamSynchronize(fn->hostStream);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_function_eval_host: ") + hipGetErrorString(e));
    std::memcpy(out_host, fn->hOut, static_cast<std::size_t>(nOut) * sizeof(double));
    return UNGAR_OK;
}

}  // extern "C"
