// ungar_amd :: helpers shared by the two run-time compilers of the library -- the function factory (function.cpp: tapes -> kernels) and the kernel
// factory (kernel_jit.cpp: size-templated solver kernels instantiated for the sizes a problem declares): content keys, the toolchain's identity,
// the cache folder, shell quoting.
#pragma once

#include <sys/stat.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

namespace ungar_amd::runtime::jit {

inline constexpr const char* kArch = "gfx950";

inline std::uint64_t Fnv1a(const void* data, std::size_t n, std::uint64_t h) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (std::size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ULL;
    }
    return h;
}

/// 128-bit content key: two FNV-1a lanes with different offsets (the second lane also sees the running first lane).
struct KeyHasher {
    std::uint64_t a = 1469598103934665603ULL, b = 0x9E3779B97F4A7C15ULL;
    void Bytes(const void* data, std::size_t n) {
        a = Fnv1a(data, n, a);
        b = Fnv1a(data, n, b ^ (a >> 7));
    }
    void Str(const std::string& s) {
        Bytes(s.data(), s.size());
        Int(static_cast<long long>(s.size()));
    }
    void Int(long long v) {
        Bytes(&v, sizeof v);
    }
    std::string Hex() const {
        char buf[40];
        std::snprintf(buf, sizeof buf, "%016llx%016llx", static_cast<unsigned long long>(a), static_cast<unsigned long long>(b));
        return buf;
    }
};

inline std::string ShellQuote(const std::string& s) {
    std::string q = "'";
    for (char c : s) {
        if (c == '\'') q += "'\\''";
        else q += c;
    }
    return q + "'";
}

inline bool FileSize(const std::string& path, long long* size) {
    struct stat st {};
    if (stat(path.c_str(), &st) != 0) return false;
    *size = static_cast<long long>(st.st_size);
    return true;
}

/// Identity of the toolchain that compiles the kernels: the ROCm release file, else the compiler's own banner.  (NOT
/// hipRuntimeGetVersion: a process that has PyTorch loaded resolves the HIP runtime to torch's bundled copy, so two
/// processes on one machine would disagree about the key of the same function.)
inline const std::string& ToolchainVersion() {
    static const std::string version = [] {
        const char* root = std::getenv("ROCM_PATH");
        std::ifstream f(std::string(root && *root ? root : "/opt/rocm") + "/.info/version");
        std::string v;
        if (f && std::getline(f, v) && !v.empty()) return "rocm-" + v;
        const char* hipcc = std::getenv("UNGAR_HIPCC");
        if (FILE* p = popen((std::string(hipcc ? hipcc : "hipcc") + " --version 2>/dev/null").c_str(), "r")) {
            char buf[256];
            while (fgets(buf, sizeof buf, p)) v += buf;
            (void)pclose(p);
        }
        return v.empty() ? std::string("unknown-toolchain") : v;
    }();
    return version;
}

inline bool MakeDirs(const std::string& path) {
    std::string cur;
    for (std::size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
        }
        if (i < path.size()) cur += path[i];
    }
    return true;
}

inline std::string DefaultFolder() {
    // reference: UNGAR_CODEGEN_FOLDER else $TMPDIR/ungar_codegen (data_types.hpp:39-41)
    if (const char* e = std::getenv("UNGAR_CODEGEN_FOLDER")) return e;
    const char* tmp = std::getenv("TMPDIR");
    return std::string(tmp ? tmp : "/tmp") + "/ungar_codegen";
}

}  // namespace ungar_amd::runtime::jit
