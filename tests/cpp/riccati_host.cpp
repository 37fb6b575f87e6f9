// TEST INFRASTRUCTURE: runs the batched Riccati recursion of ungar_amd/csrc/kernels/ocp_riccati.hpp -- the very source the
// gfx950 kernel executes -- with a sequential host execution policy, so that the algorithm can be pinned against a dense KKT
// solve without a GPU (tests/test_ocp_sqp.py).  Not part of the product: libungar_amd.so has no host path.
#include <vector>

#include "../../ungar_amd/csrc/kernels/ocp_riccati.hpp"

using namespace ungar_amd::kernels;

namespace {
template <bool PREFETCH>
struct HostExec {
    static constexpr bool kPrefetch = PREFETCH;
    static constexpr bool kAhead = true;
    static constexpr int kLanes = 64;
    using StageAB = std::vector<double>;
    using StageW = std::vector<double>;
    using StageV = std::vector<double>;
    template <class F>
    void ForEach(int n, F f) {
        for (int i = 0; i < n; ++i) f(i);
    }
    template <class F>
    void ForEachNoSync(int n, F f) {
        for (int i = 0; i < n; ++i) f(i);
    }
    void GlobalSync() {}
    void Barrier() {}
    template <class F>
    void Fetch(int n, F f, std::vector<double>& s) {
        s.resize(static_cast<std::size_t>(n));
        for (int i = 0; i < n; ++i) s[static_cast<std::size_t>(i)] = f(i);
    }
    void Commit(int n, const std::vector<double>& s, double* dst) {
        for (int i = 0; i < n; ++i) dst[i] = s[static_cast<std::size_t>(i)];
    }
};

/// The asynchronous-copy protocol of the device policy (kDma), sequentially: a copy lands either AT ONCE (a destination that is
/// still live is clobbered as early as possible) or only when its owner WAITS for it (a destination read before the wait still holds
/// the old data) -- the recursion must produce the same bits under both schedules.
template <bool DEFERRED>
struct HostDmaExec : HostExec<false> {
    static constexpr bool kDma = true;
    static constexpr int kDmaOwners = 3, kLanes = 256;  // the work-item mapping of the four-wavefront kernels
    struct Copy {
        const double* from;
        double* to;
    };
    std::vector<Copy> pending[kDmaOwners + 1];  // [0]: the all-hands copies
    template <class F>
    void Issue(std::vector<Copy>& q, int n, F addr, double* dst) {
        for (int i = 0; i < n; ++i) {
            if (DEFERRED) q.push_back({addr(i), dst + i});
            else dst[i] = *addr(i);
        }
    }
    static void Land(std::vector<Copy>& q) {
        for (const Copy& c : q) *c.to = *c.from;
        q.clear();
    }
    template <class F>
    void DmaFetch(int n, F addr, double* dst) {
        Issue(pending[0], n, addr, dst);
    }
    void DmaWait() { Land(pending[0]); }
    template <class F>
    void DmaFetchOne(int owner, int n, F addr, double* dst) {
        Issue(pending[1 + owner % kDmaOwners], n, addr, dst);
    }
    void DmaWaitOne(int owner) { Land(pending[1 + owner % kDmaOwners]); }
};

/// The single-wavefront kernels' forward pass: the computing lanes issue their own copies, 32 doubles per instruction, which land in
/// issue order; a wait names how many of the YOUNGEST instructions may still be in flight.  DEFERRED: nothing lands before a wait
/// demands it; otherwise everything lands at once.
template <bool DEFERRED>
struct HostSelfDmaExec : HostExec<true> {
    static constexpr bool kDmaSelf = true;
    struct Copy {
        const double* from;
        double* to;
    };
    std::vector<std::vector<Copy>> inFlight;  // one entry per copy instruction, oldest first
    template <class F>
    void DmaFetchSelf(int n, F addr, double* dst) {
        for (int c = 0; c < n; c += 32) {
            std::vector<Copy> instr;
            for (int i = c; i < n && i < c + 32; ++i) {
                if (DEFERRED) instr.push_back({addr(i), dst + i});
                else dst[i] = *addr(i);
            }
            inFlight.push_back(std::move(instr));
        }
    }
    template <int YOUNGER>
    void DmaWaitSelf() {
        while (inFlight.size() > static_cast<std::size_t>(YOUNGER)) {
            for (const Copy& c : inFlight.front()) *c.to = *c.from;
            inFlight.erase(inFlight.begin());
        }
    }
};

/// One wavefront per instance with asynchronous operand copies in BOTH passes (device variant "fixedq"): the backward protocol of
/// HostDmaExec on 64 lanes plus the counted forward pipeline.
template <bool DEFERRED>
struct HostDmaSelfExec : HostDmaExec<DEFERRED> {
    static constexpr int kLanes = 64;
    static constexpr bool kDmaSelf = true;
    HostSelfDmaExec<DEFERRED> self;
    template <class F>
    void DmaFetchSelf(int n, F addr, double* dst) {
        self.DmaFetchSelf(n, addr, dst);
    }
    template <int YOUNGER>
    void DmaWaitSelf() {
        self.template DmaWaitSelf<YOUNGER>();
    }
};
}  // namespace

/// Node-major contiguous arrays: jac [batch][N][nx*(nx+nu)], b [batch][N][nx], hess [batch][N][n*n], grad [batch][N][n],
/// hessN [batch][nx*nx] (or null), gradN [batch][nx] (or null), dx0 [batch][nx]; outputs dX [batch][N+1][nx], dU [batch][N][nu].
/// prefetch != 0 runs the variant that stages the next knot's operands (what the device kernel does for small problems).
extern "C" int riccati_host_solve_variant(int prefetch, int nx, int nu, int N, long long batch, double* jac, double* b, double* hess, double* grad, double* hessN,
                                          double* gradN, double* dx0, double regularization, double* dX, double* dU, int* status);
extern "C" int riccati_host_solve(int nx, int nu, int N, long long batch, double* jac, double* b, double* hess, double* grad, double* hessN, double* gradN,
                                  double* dx0, double regularization, double* dX, double* dU, int* status) {
    return riccati_host_solve_variant(1, nx, nu, N, batch, jac, b, hess, grad, hessN, gradN, dx0, regularization, dX, dU, status);
}
extern "C" int riccati_host_solve_variant(int prefetch, int nx, int nu, int N, long long batch, double* jac, double* b, double* hess, double* grad, double* hessN, double* gradN,
                                          double* dx0, double regularization, double* dX, double* dU, int* status) {
    const int n = nx + nu;
    std::vector<double> gains(static_cast<std::size_t>(batch) * N * nu * (nx + 1));
    RiccatiArgs a{nx, nu, N, batch,
                  {jac, static_cast<long long>(N) * nx * n, static_cast<long long>(nx) * n, 1},
                  {b, static_cast<long long>(N) * nx, nx, 1},
                  {hess, static_cast<long long>(N) * n * n, static_cast<long long>(n) * n, 1},
                  {grad, static_cast<long long>(N) * n, n, 1},
                  {hessN, static_cast<long long>(nx) * nx, 0, 1},
                  {gradN, nx, 0, 1},
                  {dx0, nx, 0, 1},
                  {dX, static_cast<long long>(N + 1) * nx, nx, 1},
                  {dU, static_cast<long long>(N) * nu, nu, 1},
                  gains.data(), regularization, status};
    std::vector<double> scratch(static_cast<std::size_t>(RiccatiScratchDoubles(nx, nu)));
    if (prefetch == 2) {  // the instantiations with sizes fixed at compile time (what the device launches for the reference's OCPs): in-register Cholesky for nu <= 8
        HostExec<false> ex;
        for (long long i = 0; i < batch; ++i) {
            if (nx == 13 && nu == 4) RiccatiInstance<HostExec<false>, 13, 4>(a, i, scratch.data(), ex);
            else if (nx == 6 && nu == 2) RiccatiInstance<HostExec<false>, 6, 2>(a, i, scratch.data(), ex);
            else if (nx == 37 && nu == 12) RiccatiInstance<HostExec<false>, 37, 12>(a, i, scratch.data(), ex);  // 2 x 4 register tiles
            else if (nx == 13 && nu == 24) RiccatiInstance<HostExec<false>, 13, 24>(a, i, scratch.data(), ex);
            else return 1;
        }
    } else if (prefetch == 7 || prefetch == 8) {  // one wavefront, asynchronous copies in both passes
        auto run = [&](auto ex) {
            using E = decltype(ex);
            for (long long i = 0; i < batch; ++i) {
                if (nx == 13 && nu == 4) RiccatiInstance<E, 13, 4>(a, i, scratch.data(), ex);
                else if (nx == 6 && nu == 2) RiccatiInstance<E, 6, 2>(a, i, scratch.data(), ex);
                else return 1;
            }
            return 0;
        };
        return prefetch == 7 ? run(HostDmaSelfExec<false>{}) : run(HostDmaSelfExec<true>{});
    } else if (prefetch == 5 || prefetch == 6) {  // one-wavefront kernels: staged backward pass, forward pass on self-issued copies with counted waits
        auto run = [&](auto ex) {
            using E = decltype(ex);
            for (long long i = 0; i < batch; ++i) {
                if (nx == 13 && nu == 4) RiccatiInstance<E, 13, 4>(a, i, scratch.data(), ex);
                else if (nx == 6 && nu == 2) RiccatiInstance<E, 6, 2>(a, i, scratch.data(), ex);
                else return 1;
            }
            return 0;
        };
        return prefetch == 5 ? run(HostSelfDmaExec<false>{}) : run(HostSelfDmaExec<true>{});
    } else if (prefetch == 3 || prefetch == 4) {  // compile-time sizes with the asynchronous-copy protocol (3: copies land at once, 4: at the wait)
        auto run = [&](auto ex) {
            using E = decltype(ex);
            for (long long i = 0; i < batch; ++i) {
                if (nx == 13 && nu == 4) RiccatiInstance<E, 13, 4>(a, i, scratch.data(), ex);
                else if (nx == 37 && nu == 12) RiccatiInstance<E, 37, 12>(a, i, scratch.data(), ex);
                else if (nx == 13 && nu == 24) RiccatiInstance<E, 13, 24>(a, i, scratch.data(), ex);
                else return 1;
            }
            return 0;
        };
        return prefetch == 3 ? run(HostDmaExec<false>{}) : run(HostDmaExec<true>{});
    } else if (prefetch) {
        HostExec<true> ex;
        for (long long i = 0; i < batch; ++i) RiccatiInstance(a, i, scratch.data(), ex);
    } else {
        HostExec<false> ex;
        for (long long i = 0; i < batch; ++i) RiccatiInstance(a, i, scratch.data(), ex);
    }
    return 0;
}

/// Stage equality rows E_k [dx; du] + e_k = 0 (eq [batch][N][ne*n], eqv [batch][N][ne]) next to the operands above; variant 0: run-time
/// sizes, 1: run-time sizes with staged operands, 2: the compile-time instantiations of the reference's OCPs with carried quantities
/// (quadrotor 17 + 4, RC car 8 + 2, quadruped 25 + 24 with 16 foot-contact rows), 3 / 4: those under the asynchronous-copy protocol.
extern "C" int riccati_host_solve_eq(int variant, int nx, int nu, int ne, int N, long long batch, double* jac, double* b, double* hess, double* grad, double* hessN, int hessNld,
                                     double* gradN, double* dx0, double* eq, double* eqv, double regularization, double* dX, double* dU, int* status) {
    const int n = nx + nu;
    std::vector<double> gains(static_cast<std::size_t>(batch) * N * nu * (nx + 1));
    const int ldN = hessNld > 0 ? hessNld : nx;
    RiccatiArgs a{nx, nu, N, batch,
                  {jac, static_cast<long long>(N) * nx * n, static_cast<long long>(nx) * n, 1},
                  {b, static_cast<long long>(N) * nx, nx, 1},
                  {hess, static_cast<long long>(N) * n * n, static_cast<long long>(n) * n, 1},
                  {grad, static_cast<long long>(N) * n, n, 1},
                  {hessN, static_cast<long long>(nx) * ldN, 0, 1},
                  {gradN, nx, 0, 1},
                  {dx0, nx, 0, 1},
                  {dX, static_cast<long long>(N + 1) * nx, nx, 1},
                  {dU, static_cast<long long>(N) * nu, nu, 1},
                  gains.data(), regularization, status};
    a.ne = ne;
    a.eq = {eq, static_cast<long long>(N) * ne * n, static_cast<long long>(ne) * n, 1};
    a.eqv = {eqv, static_cast<long long>(N) * ne, ne, 1};
    a.hessNld = hessNld;
    std::vector<double> scratch(static_cast<std::size_t>(RiccatiScratchDoubles(nx, nu, ne)));
    auto fixed = [&](auto ex) {
        using E = decltype(ex);
        for (long long i = 0; i < batch; ++i) {
            if (nx == 17 && nu == 4 && ne == 0) RiccatiInstance<E, 17, 4, 0>(a, i, scratch.data(), ex);
            else if (nx == 8 && nu == 2 && ne == 0) RiccatiInstance<E, 8, 2, 0>(a, i, scratch.data(), ex);
            else if (nx == 25 && nu == 24 && ne == 16) RiccatiInstance<E, 25, 24, 16>(a, i, scratch.data(), ex);
            else if (nx == 25 && nu == 24 && ne == 0) RiccatiInstance<E, 25, 24, 0>(a, i, scratch.data(), ex);  // rows eliminated before the recursion: blocked Cholesky
            else return 1;
        }
        return 0;
    };
    if (variant == 2) return fixed(HostExec<false>{});
    if (variant == 3) return fixed(HostDmaExec<false>{});
    if (variant == 4) return fixed(HostDmaExec<true>{});
    if (variant == 1) {
        HostExec<true> ex;
        for (long long i = 0; i < batch; ++i) RiccatiInstance(a, i, scratch.data(), ex);
    } else {
        HostExec<false> ex;
        for (long long i = 0; i < batch; ++i) RiccatiInstance(a, i, scratch.data(), ex);
    }
    return 0;
}

/// The folded-triangle index maps of ocp_riccati.hpp, for tests/test_ocp_sqp.py.
extern "C" int riccati_folded_index(int n, int r, int c) { return RiccatiFoldedIndex(n, r, c); }
extern "C" int riccati_folded_source(int n, int i) { return RiccatiFoldedSource(n, i); }
