// ungar_amd :: run-time function factory -- record -> derive -> HIP codegen -> hipcc -> hipModule.
//
// MI355X-native counterpart of FunctionFactory::Worker (reference
// include/ungar/autodiff/function.hpp:392-605):
//   CppAD tape + optimize            -> tape::Graph (hash-consed, built optimised)
//   ModelCSourceGen sparse Jac/Hess  -> tape::Differentiator (parameters trimmed, upper-tri Hessian)
//   GccCompiler / createDynamicLibrary -> `hipcc --offload-arch=gfx950 --genco`  (process boundary)
//   LinuxDynamicLib / dlopen         -> hipModuleLoad / hipModuleGetFunction
//   existence-only .so cache (function.hpp:420-451, stale-cache hazard)
//                                    -> cache keyed by a hash of (generated source, arch, flags)
// Kernels map one lane to one problem instance and address operands through strides, so the same
// code object serves batch = 1 host calls (what Ungar::Autodiff::Function needs) and large batches.
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/ungar_amd.h"
#include "../tape/emit.hpp"

namespace ungar_amd::runtime {
int Fail(int code, const std::string& msg);  // c_api.cpp
}

using namespace ungar_amd;

struct ungar_function {
    std::string name;
    int64_t n = 0, p = 0, m = 0;
    uint32_t enabled = 0;
    std::vector<int32_t> jacRows, jacCols, hesRows, hesCols;
    hipModule_t modules[3] = {nullptr, nullptr, nullptr};  // value, Jacobian, Hessian: one code object each (compiled concurrently)
    hipFunction_t kValue = nullptr, kJac = nullptr, kHes = nullptr;
    std::string codeObjectPath;
    bool cacheHit = false;
    // staging buffers for the single-instance host entry points
    double *dIn = nullptr, *dOut = nullptr;
    int64_t dOutSize = 0;
    ~ungar_function() {
        if (dIn) (void)hipFree(dIn);
        if (dOut) (void)hipFree(dOut);
        for (hipModule_t mod : modules)
            if (mod) (void)hipModuleUnload(mod);
    }
};

namespace {

using runtime::Fail;

constexpr uint32_t kEnableJacobian = 1U << 1;  // EnabledDerivatives::JACOBIAN, autodiff/data_types.hpp:95-100
constexpr uint32_t kEnableHessian = 1U << 2;   // EnabledDerivatives::HESSIAN

std::uint64_t Fnv1a(const std::string& s, std::uint64_t h = 1469598103934665603ULL) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ULL;
    }
    return h;
}

bool MakeDirs(const std::string& path) {
    std::string cur;
    for (std::size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
        }
        if (i < path.size()) cur += path[i];
    }
    return true;
}

std::string DefaultFolder() {
    // reference: UNGAR_CODEGEN_FOLDER else $TMPDIR/ungar_codegen (data_types.hpp:39-41)
    if (const char* e = std::getenv("UNGAR_CODEGEN_FOLDER")) return e;
    const char* tmp = std::getenv("TMPDIR");
    return std::string(tmp ? tmp : "/tmp") + "/ungar_codegen";
}

/// Emits one `extern "C" __global__` kernel: lane = instance, strided operands.
std::string EmitKernel(const std::string& kernelName, const tape::Graph& g, int64_t nIn, const std::vector<tape::Id>& values, std::size_t* statements) {
    std::vector<std::string> inNames;
    inNames.reserve(static_cast<std::size_t>(nIn));
    // inputs are spelled as loads at their uses (`in` derives from a __restrict__ parameter, so the compiler
    // merges repeated loads and places them where they are needed): reading every input into a local up
    // front keeps ~n values alive from the top of a whole-horizon kernel and made the register allocator
    // the dominant compile cost (objective value kernel, 960 inputs: 21 s -> 2 s, 1 KB of scratch -> none)
    for (int64_t i = 0; i < nIn; ++i) inNames.push_back("in[" + std::to_string(i) + " * xes]");
    std::vector<tape::OutputSlot> slots;
    for (std::size_t k = 0; k < values.size(); ++k) slots.push_back({values[k], "out[" + std::to_string(k) + " * oes] = %s;"});
    std::ostringstream os;
    // launched with 64-lane workgroups (LaunchFn): tell the compiler, so that it may use the full register file
    os << "extern \"C\" __global__ __launch_bounds__(64) void " << kernelName
       << "(const double* __restrict__ xp, long long xbs, long long xes, double* __restrict__ outBase, long long obs, long long oes, long long "
          "batch) {\n"
       << "    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n"
       << "    if (i >= batch) return;\n"
       << "    const double* __restrict__ in = xp + i * xbs;\n"
       << "    double* __restrict__ out = outBase + i * obs;\n";
    tape::Emitter em{g, inNames};
    os << em.Emit(slots) << "}\n\n";
    if (statements) *statements = em.Stats().statements;
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

    // ---- derivatives (parameters trimmed: columns [0, n) only; function.hpp:529-574) -------------
    auto fn = std::make_unique<ungar_function>();
    fn->name = name;
    fn->n = n;
    fn->p = p;
    fn->m = m;
    fn->enabled = enabled_derivatives;
    const std::vector<tape::Id> valueIds = t.outputs;
    tape::SparseEntries jac, hes;
    tape::Differentiator diff{t};
    if (enabled_derivatives & kEnableJacobian) {
        jac = diff.Jacobian(static_cast<int>(n));
        fn->jacRows.assign(jac.row.begin(), jac.row.end());
        fn->jacCols.assign(jac.col.begin(), jac.col.end());
    }
    if (enabled_derivatives & kEnableHessian) {
        hes = diff.Hessian(0, static_cast<int>(n));
        fn->hesRows.assign(hes.row.begin(), hes.row.end());
        fn->hesCols.assign(hes.col.begin(), hes.col.end());
    }

    // ---- HIP source: one translation unit per kernel so that the compiler runs on all of them at once ----
    struct Unit {
        const char* kernel = nullptr;
        const char* tag = nullptr;
        const std::vector<tape::Id>* values = nullptr;
        hipFunction_t* handle = nullptr;
        std::string src, object, tmp, log, flags;
        FILE* pipe = nullptr;
        std::size_t statements = 0;
    };
    std::vector<Unit> units;
    auto unit = [](const char* kernel, const char* tag, const std::vector<tape::Id>* values, hipFunction_t* handle) {
        Unit u;
        u.kernel = kernel;
        u.tag = tag;
        u.values = values;
        u.handle = handle;
        return u;
    };
    units.push_back(unit("ungar_fn_forward_zero", "value", &valueIds, &fn->kValue));
    if (!jac.value.empty()) units.push_back(unit("ungar_fn_sparse_jacobian", "jacobian", &jac.value, &fn->kJac));
    if (!hes.value.empty()) units.push_back(unit("ungar_fn_sparse_hessian", "hessian", &hes.value, &fn->kHes));
    // Compile flags.  The machine instruction schedulers (pre- and post-RA) account for > 95 % of the compile
    // time of a large straight-line kernel (a whole-horizon constraint Jacobian of 23 k statements: 64 s -> 6 s
    // without them) and the emitter already orders statements depth-first, so they are switched off above
    // kBigKernel statements; node-sized functions -- the ones evaluated in large batches -- keep the full
    // pipeline.  UNGAR_AMD_JIT_FLAGS replaces the optimisation flags altogether.
    constexpr std::size_t kBigKernel = 3000;
    const char* custom = std::getenv("UNGAR_AMD_JIT_FLAGS");
    std::uint64_t hash = 1469598103934665603ULL;
    for (Unit& u : units) {
        u.src = "// generated by ungar_amd (runtime/function.cpp) for function '" + fn->name + "'\n#include <hip/hip_runtime.h>\n\n" +
                EmitKernel(u.kernel, g, n + p, *u.values, &u.statements);
        u.flags = std::string("--offload-arch=gfx950 -std=c++17 ") +
                  (custom ? custom : u.statements > kBigKernel ? "-O3 -mllvm -enable-misched=false -mllvm -enable-post-misched=false" : "-O3");
        hash = Fnv1a(u.flags, Fnv1a(u.src, hash));
    }
    char hashHex[32];
    std::snprintf(hashHex, sizeof hashHex, "%016llx", static_cast<unsigned long long>(hash));
    const bool verbose = std::getenv("UNGAR_AMD_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();

    // ---- compile or reuse (layout mirrors <folder>/<name>/cppad_cg/<name>_lib.so, function.hpp:433-435)
    const std::string dir = std::string(folder && *folder ? folder : DefaultFolder()) + "/" + fn->name + "/ungar_amd";
    const std::string base = dir + "/" + fn->name + "_" + hashHex;
    fn->codeObjectPath = base + "_value.hsaco";
    bool have = true;
    for (Unit& u : units) {
        u.object = base + "_" + u.tag + ".hsaco";
        struct stat st {};
        have = have && stat(u.object.c_str(), &st) == 0 && st.st_size > 0;
    }
    if (!have || recompile) {
        if (!MakeDirs(dir)) return Fail(UNGAR_E_IO, "cannot create code-generation folder '" + dir + "'");
        const char* hipcc = std::getenv("UNGAR_HIPCC");
        for (Unit& u : units) {
            const std::string hip = base + "_" + u.tag + ".hip";
            {
                std::ofstream f(hip);
                f << u.src;
                if (!f) return Fail(UNGAR_E_IO, "cannot write '" + hip + "'");
            }
            // temp name + rename = atomic publish (function.hpp:485-487, 501-502)
            u.tmp = u.object + "." + std::to_string(getpid()) + ".tmp";
            const std::string cmd = std::string(hipcc ? hipcc : "hipcc") + " " + u.flags + " --genco -o '" + u.tmp + "' '" + hip + "' 2>&1";
            u.pipe = popen(cmd.c_str(), "r");  // all compilers start now and run concurrently
            if (!u.pipe) return Fail(UNGAR_E_COMPILE, "cannot start hipcc for function '" + fn->name + "'");
        }
        std::string failure;
        for (Unit& u : units) {
            char buf[512];
            while (fgets(buf, sizeof buf, u.pipe)) u.log += buf;
            const int st = pclose(u.pipe);
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : -1;
            if (rc != 0) failure += "hipcc failed (" + std::to_string(rc) + ") for the " + u.tag + " kernel of function '" + fn->name + "':\n" + u.log;
        }
        if (!failure.empty()) return Fail(UNGAR_E_COMPILE, failure);
        for (Unit& u : units)
            if (std::rename(u.tmp.c_str(), u.object.c_str()) != 0) return Fail(UNGAR_E_IO, "cannot publish '" + u.object + "'");
    } else {
        fn->cacheHit = true;
    }
    if (verbose)
        std::fprintf(stderr, "[ungar_amd] function '%s': %lld tape nodes, n=%lld p=%lld m=%lld, jac nnz %zu, hes nnz %zu, %s in %.1f s\n", fn->name.c_str(),
                     static_cast<long long>(num_nodes), static_cast<long long>(n), static_cast<long long>(p), static_cast<long long>(m), jac.value.size(),
                     hes.value.size(), fn->cacheHit ? "code objects reused" : "compiled",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

    // ---- load ---------------------------------------------------------------------------------------
    // UNGAR_AMD_COMPILE_ONLY: stop after publishing the code objects (cache warm-up on a machine without a
    // GPU, e.g. a build host); the returned function reports its sparsity but every evaluation fails.
    if (std::getenv("UNGAR_AMD_COMPILE_ONLY")) {
        *out = fn.release();
        return UNGAR_OK;
    }
    for (std::size_t k = 0; k < units.size(); ++k) {
        hipError_t e = hipModuleLoad(&fn->modules[k], units[k].object.c_str());
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("hipModuleLoad('") + units[k].object + "'): " + hipGetErrorString(e));
        e = hipModuleGetFunction(units[k].handle, fn->modules[k], units[k].kernel);
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("hipModuleGetFunction: ") + hipGetErrorString(e));
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

const char* ungar_function_code_object(const ungar_function* fn) {
    return fn ? fn->codeObjectPath.c_str() : "";
}

int ungar_function_jacobian_sparsity(const ungar_function* fn, const int32_t** rows, const int32_t** cols, int64_t* nnz) {
    if (!fn || !rows || !cols || !nnz) return Fail(UNGAR_E_INVALID, "ungar_function_jacobian_sparsity: null argument");
    if (!fn->kJac && !(fn->enabled & kEnableJacobian)) return Fail(UNGAR_E_UNSUPPORTED, "function '" + fn->name + "' was made without JACOBIAN");
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

static int LaunchFn(const ungar_function* fn, hipFunction_t k, const char* what, const ungar_operand* xp, const ungar_operand* out,
                    int64_t batch, void* stream) {
    if (!fn || !xp || !out) return Fail(UNGAR_E_INVALID, std::string(what) + ": null argument");
    if (!k) return Fail(UNGAR_E_UNSUPPORTED, std::string(what) + ": kernel not available for function '" + fn->name + "' (derivative not enabled, or made with UNGAR_AMD_COMPILE_ONLY)");
    if (batch < 0) return Fail(UNGAR_E_INVALID, std::string(what) + ": negative batch");
    if (batch == 0) return UNGAR_OK;
    if (!xp->base || !out->base) return Fail(UNGAR_E_INVALID, std::string(what) + ": null operand base");
    const double* in = xp->base;
    long long xbs = xp->instance_stride, xes = xp->element_stride, obs = out->instance_stride, oes = out->element_stride, b = batch;
    double* o = out->base;
    void* args[] = {&in, &xbs, &xes, &o, &obs, &oes, &b};
    const unsigned block = 64;
    const hipError_t e = hipModuleLaunchKernel(k, static_cast<unsigned>((batch + block - 1) / block), 1, 1, block, 1, 1, 0,
                                               static_cast<hipStream_t>(stream), args, nullptr);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return UNGAR_OK;
}

int ungar_function_forward_zero(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* y, int64_t batch, void* stream) {
    return LaunchFn(fn, fn ? fn->kValue : nullptr, "ungar_function_forward_zero", xp, y, batch, stream);
}
// A derivative that is enabled but structurally empty (Jacobian of a function of the parameters only, Hessian of a
// linear objective) has no kernel and nothing to write: the reference returns an empty sparse matrix there
// (function.hpp:216-230, 236-259), so the batched calls succeed without a launch.
int ungar_function_sparse_jacobian(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* jac, int64_t batch, void* stream) {
    if (fn && xp && jac && batch >= 0 && (fn->enabled & kEnableJacobian) && fn->jacRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? fn->kJac : nullptr, "ungar_function_sparse_jacobian", xp, jac, batch, stream);
}
int ungar_function_sparse_hessian(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* hes, int64_t batch, void* stream) {
    if (fn && xp && hes && batch >= 0 && (fn->enabled & kEnableHessian) && fn->hesRows.empty()) return UNGAR_OK;
    return LaunchFn(fn, fn ? fn->kHes : nullptr, "ungar_function_sparse_hessian", xp, hes, batch, stream);
}

/// Single-instance host call: H2D, batch-1 launch, D2H, on the null stream, synchronous.
int ungar_function_eval_host(ungar_function* fn, int32_t what, const double* xp_host, double* out_host) {
    if (!fn || !xp_host || !out_host) return Fail(UNGAR_E_INVALID, "ungar_function_eval_host: null argument");
    const int64_t nIn = fn->n + fn->p;
    const int64_t nOut = what == 0 ? fn->m : what == 1 ? static_cast<int64_t>(fn->jacRows.size()) : static_cast<int64_t>(fn->hesRows.size());
    hipFunction_t k = what == 0 ? fn->kValue : what == 1 ? fn->kJac : fn->kHes;
    if (what < 0 || what > 2) return Fail(UNGAR_E_INVALID, "ungar_function_eval_host: what must be 0 (value), 1 (Jacobian) or 2 (Hessian)");
    if (nOut == 0 && what != 0 && (fn->enabled & (what == 1 ? kEnableJacobian : kEnableHessian))) return UNGAR_OK;  // enabled but structurally empty
    if (!k) return Fail(UNGAR_E_UNSUPPORTED, "ungar_function_eval_host: kernel not available for function '" + fn->name + "' (derivative not enabled, or made with UNGAR_AMD_COMPILE_ONLY)");
    if (nOut == 0) return UNGAR_OK;
    hipError_t e = hipSuccess;
    if (!fn->dIn) e = hipMalloc(&fn->dIn, static_cast<std::size_t>(std::max<int64_t>(nIn, 1)) * sizeof(double));
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

}  // extern "C"
