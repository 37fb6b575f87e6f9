// ungar_amd :: kernel factory (see kernel_jit.hpp).
#include "kernel_jit.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>

#include "../../../include/ungar_amd.h"
#include "jit_common.hpp"

namespace ungar_amd::runtime {
int Fail(int code, const std::string& msg);  // c_api.cpp

namespace {

using namespace jit;

constexpr const char* kKernelCacheFormat = "ungar_amd-kernel-cache-1";

std::string ReadFile(const std::string& path, bool* ok) {
    std::ifstream f(path, std::ios::binary);
    std::ostringstream os;
    os << f.rdbuf();
    *ok = static_cast<bool>(f);
    return os.str();
}

/// Adds `text` and, recursively, every file it names with #include "..." (relative to the including file's folder, then to the root) to the key.
bool HashIncludes(const std::string& text, const std::string& folder, const std::string& root, std::set<std::string>& seen, KeyHasher& key, std::string* missing) {
    std::size_t pos = 0;
    while ((pos = text.find("#include \"", pos)) != std::string::npos) {
        const std::size_t begin = pos + 10, end = text.find('"', begin);
        if (end == std::string::npos) break;
        const std::string name = text.substr(begin, end - begin);
        pos = end;
        std::string path;
        bool ok = false;
        std::string content;
        for (const std::string& base : {folder, root}) {
            path = base + "/" + name;
            content = ReadFile(path, &ok);
            if (ok) break;
        }
        if (!ok) {  // an include inside a comment or a disabled #if block, or a system-style header on the compiler's own path: not part of the key
            if (missing->empty()) *missing = name;
            continue;
        }
        char resolved[4096];
        const std::string canonical = realpath(path.c_str(), resolved) ? std::string(resolved) : path;
        if (!seen.insert(canonical).second) continue;
        key.Str(name);
        key.Str(content);
        if (!HashIncludes(content, canonical.substr(0, canonical.rfind('/')), root, seen, key, missing)) return false;
    }
    return true;
}

struct Resources {
    int vgprs = -1, scratch = -1;
};
/// "remark: ...:     VGPRs: 128" / "ScratchSize [bytes/lane]: 0" of -Rpass-analysis=kernel-resource-usage, for the kernel named `kernel`.
Resources ParseResources(const std::string& log, const std::string& kernel) {
    Resources r;
    const std::size_t at = log.find("Function Name: " + kernel);
    if (at == std::string::npos) return r;
    auto number = [&](const char* label) {
        const std::size_t p = log.find(label, at);
        return p == std::string::npos ? -1 : std::atoi(log.c_str() + p + std::strlen(label));
    };
    r.vgprs = number(" VGPRs: ");
    const int agprs = number("AGPRs: ");
    if (r.vgprs >= 0 && agprs > 0) r.vgprs += agprs;
    r.scratch = number("ScratchSize [bytes/lane]: ");
    return r;
}

std::mutex g_mutex;
std::map<std::string, std::unique_ptr<JitKernel>> g_kernels;  // by entry name + key

}  // namespace

std::string KernelSourceRoot() {
    static const std::string root = [] {
        if (const char* e = std::getenv("UNGAR_AMD_KERNEL_SOURCES")) return std::string(e);
        Dl_info info{};
        if (!dladdr(reinterpret_cast<const void*>(&KernelSourceRoot), &info) || !info.dli_fname) return std::string();
        char resolved[4096];
        std::string dir = realpath(info.dli_fname, resolved) ? std::string(resolved) : std::string(info.dli_fname);
        for (int up = 0; up < 4; ++up) {  // lib/libungar_amd.so, lib/measurement/libungar_amd.so
            const std::size_t slash = dir.rfind('/');
            if (slash == std::string::npos) break;
            dir.erase(slash);
            long long size = 0;
            if (FileSize(dir + "/csrc/kernels/ocp_riccati.hpp", &size)) return dir + "/csrc";
        }
        return std::string();
    }();
    return root;
}

const JitKernel* GetKernel(const KernelRequest& rq) {
    const std::string root = KernelSourceRoot();
    if (root.empty()) {
        Fail(UNGAR_E_IO, "kernel factory: the kernel sources (ungar_amd/csrc) were not found next to the library; set UNGAR_AMD_KERNEL_SOURCES");
        return nullptr;
    }
    const char* custom = std::getenv("UNGAR_AMD_JIT_FLAGS");
    const char* hipcc = std::getenv("UNGAR_HIPCC");
    KeyHasher key;
    key.Str(kKernelCacheFormat);
    key.Str(kArch);
    key.Str(ToolchainVersion());
    key.Str(custom ? custom : "");
    key.Str(hipcc ? hipcc : "");
    key.Str(rq.kernel);
    key.Str(rq.source);
    for (int w : rq.occupancies) key.Int(w);
    {
        std::set<std::string> seen;
        std::string missing;
        (void)HashIncludes(rq.source, root, root, seen, key, &missing);  // (a header that is really missing fails the compile below, with the compiler's message)
    }
    const std::string keyHex = key.Hex();
    // the code object is loaded into the CURRENT device's context: the in-process entry carries the device (the on-disk entry is shared)
    int device = 0;
    if (!std::getenv("UNGAR_AMD_COMPILE_ONLY") && hipGetDevice(&device) != hipSuccess) {
        (void)hipGetLastError();
        device = 0;
    }
    const std::string processKey = rq.name + keyHex + "@" + std::to_string(device);
    std::lock_guard<std::mutex> guard(g_mutex);
    if (auto it = g_kernels.find(processKey); it != g_kernels.end()) return it->second.get();

    const std::string dir = DefaultFolder() + "/ungar_amd_kernels";
    const std::string base = dir + "/" + rq.name + "_" + keyHex, object = base + ".hsaco", metaPath = base + ".meta";
    auto kernel = std::make_unique<JitKernel>();
    kernel->object = object;
    auto lookup = [&] {
        bool ok = false;
        const std::string text = ReadFile(metaPath, &ok);
        long long size = 0;
        std::istringstream is(text);
        std::string format, word;
        long long recorded = 0;
        if (!ok || !std::getline(is, format) || format != kKernelCacheFormat) return false;
        if (!(is >> word >> kernel->wavesPerEu >> kernel->vgprs >> kernel->scratchBytes >> recorded) || word != "kernel") return false;
        return FileSize(object, &size) && size == recorded && size > 0;
    };
    bool hit = lookup();
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
        if (!MakeDirs(dir)) {
            Fail(UNGAR_E_IO, "kernel factory: cannot create '" + dir + "'");
            return nullptr;
        }
        lock.fd = open((base + ".lock").c_str(), O_CREAT | O_RDWR, 0666);
        if (lock.fd >= 0) (void)flock(lock.fd, LOCK_EX);
        hit = lookup();  // published by another process while this one waited
    }
    if (!hit) {
        struct Candidate {
            int waves = 0;
            std::string source, tmpObject, log;
            FILE* pipe = nullptr;
            Resources res;
            int rc = -1;
        };
        std::vector<Candidate> candidates;
        const std::string unique = "." + std::to_string(getpid()) + ".tmp";
        for (int w : rq.occupancies) {
            Candidate c;
            c.waves = w;
            std::string text = rq.source;
            for (std::size_t p = 0; (p = text.find("%W%", p)) != std::string::npos;) text.replace(p, 3, std::to_string(w));
            c.source = base + "_w" + std::to_string(w) + unique + ".hip";
            c.tmpObject = base + "_w" + std::to_string(w) + unique + ".hsaco";
            {
                std::ofstream f(c.source);
                f << "// generated by ungar_amd (runtime/kernel_jit.cpp): kernel '" << rq.name << "', " << w << " wavefront(s) per SIMD\n" << text;
            }
            const std::string cmd = std::string(hipcc ? hipcc : "hipcc") + " --offload-arch=" + kArch + " -std=c++20 " + (custom ? custom : "-O3") + " -I " + ShellQuote(root) +
                                    " -Rpass-analysis=kernel-resource-usage --genco -o " + ShellQuote(c.tmpObject) + " " + ShellQuote(c.source) + " 2>&1";
            c.pipe = popen(cmd.c_str(), "r");  // all candidates compile concurrently
            candidates.push_back(std::move(c));
        }
        std::string failure;
        for (Candidate& c : candidates) {
            if (!c.pipe) {
                failure += "cannot start hipcc; ";
                continue;
            }
            char buf[512];
            while (fgets(buf, sizeof buf, c.pipe)) c.log += buf;
            const int st = pclose(c.pipe);
            c.rc = WIFEXITED(st) ? WEXITSTATUS(st) : -1;
            c.res = ParseResources(c.log, rq.kernel);
            (void)std::remove(c.source.c_str());
            if (c.rc != 0) failure += "hipcc failed (" + std::to_string(c.rc) + ") for kernel '" + rq.name + "' at " + std::to_string(c.waves) + " wavefronts per SIMD:\n" + c.log;
        }
        // the largest occupancy whose kernel keeps everything in registers; failing that, the candidate with the least scratch
        const Candidate* best = nullptr;
        for (const Candidate& c : candidates) {
            if (c.rc != 0 || c.res.scratch < 0) continue;
            if (!best) best = &c;
            else if (c.res.scratch == 0 && (best->res.scratch != 0 || c.waves > best->waves)) best = &c;
            else if (c.res.scratch != 0 && best->res.scratch != 0 && c.res.scratch < best->res.scratch) best = &c;
        }
        if (!best) {
            for (const Candidate& c : candidates) (void)std::remove(c.tmpObject.c_str());
            Fail(UNGAR_E_COMPILE, failure.empty() ? "kernel factory: no resource report for kernel '" + rq.name + "'" : failure);
            return nullptr;
        }
        long long size = 0;
        const bool published = FileSize(best->tmpObject, &size) && size > 0 && std::rename(best->tmpObject.c_str(), object.c_str()) == 0;
        for (const Candidate& c : candidates)
            if (&c != best) (void)std::remove(c.tmpObject.c_str());
        if (!published) {
            Fail(UNGAR_E_IO, "kernel factory: cannot publish '" + object + "'");
            return nullptr;
        }
        kernel->wavesPerEu = best->waves;
        kernel->vgprs = best->res.vgprs;
        kernel->scratchBytes = best->res.scratch;
        const std::string tmpMeta = metaPath + unique;
        {
            std::ofstream f(tmpMeta);
            f << kKernelCacheFormat << "\nkernel " << kernel->wavesPerEu << ' ' << kernel->vgprs << ' ' << kernel->scratchBytes << ' ' << size << "\nname " << rq.name << ' '
              << rq.kernel << "\n";
        }
        if (std::rename(tmpMeta.c_str(), metaPath.c_str()) != 0) {
            Fail(UNGAR_E_IO, "kernel factory: cannot publish '" + metaPath + "'");
            return nullptr;
        }
    }
    kernel->cacheHit = hit;
    if (std::getenv("UNGAR_AMD_VERBOSE"))
        std::fprintf(stderr, "[ungar_amd] kernel '%s': %s, %d wavefront(s) per SIMD, %d registers, %d bytes of scratch (key %s)\n", rq.name.c_str(),
                     hit ? "cache hit" : "compiled", kernel->wavesPerEu, kernel->vgprs, kernel->scratchBytes, keyHex.c_str());
    if (!std::getenv("UNGAR_AMD_COMPILE_ONLY")) {
        hipModule_t module = nullptr;
        hipError_t e = hipModuleLoad(&module, object.c_str());
        if (e == hipSuccess) e = hipModuleGetFunction(&kernel->function, module, rq.kernel.c_str());
        if (e != hipSuccess) {
            Fail(UNGAR_E_HIP, std::string("kernel factory: loading '") + object + "': " + hipGetErrorString(e));
            return nullptr;
        }
    }
    const JitKernel* out = kernel.get();
    g_kernels[processKey] = std::move(kernel);
    return out;
}

}  // namespace ungar_amd::runtime
