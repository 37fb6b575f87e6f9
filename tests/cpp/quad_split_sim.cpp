// CPU simulator of the SPLIT lane-per-leg program (ungar_amd/csrc/gen/anymal_split_gen.hpp): the two halves run as two
// threads, one 4-lane quad each, and hand their messages over through the same protocol as the two wavefronts of the
// GPU workgroup (quad_split_kernel.hpp): a ring of kRing message slots, posted / consumed counters, the solved
// accelerations on the way back.  Pins the hand-over ORDER (a half that reads an item before it was sent, or a ring
// slot that is overwritten before it was consumed, shows up as a wrong entry or as a deadlock caught by the test's
// timeout) and the math of the two halves against the golden vectors, without a GPU.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <thread>

#include "anymal_quad_gen.hpp"  // kLegConstants
#include "anymal_split_gen.hpp"

namespace {

struct Quad {
    double v[4];
    Quad() : v{0, 0, 0, 0} {}
    Quad(double s) : v{s, s, s, s} {}  // NOLINT
};
#define QUAD_BIN(op)                                               \
    inline Quad operator op(const Quad& a, const Quad& b) {       \
        Quad r;                                                    \
        for (int l = 0; l < 4; ++l) r.v[l] = a.v[l] op b.v[l];     \
        return r;                                                  \
    }
QUAD_BIN(+)
QUAD_BIN(-)
QUAD_BIN(*)
QUAD_BIN(/)
inline Quad operator-(const Quad& a) {
    Quad r;
    for (int l = 0; l < 4; ++l) r.v[l] = -a.v[l];
    return r;
}
#define QUAD_FN(fn)                                           \
    inline Quad fn(const Quad& a) {                           \
        Quad r;                                               \
        for (int l = 0; l < 4; ++l) r.v[l] = std::fn(a.v[l]); \
        return r;                                             \
    }
QUAD_FN(sin)
QUAD_FN(cos)
QUAD_FN(sqrt)

constexpr int kRing = 3;

struct Channel {
    Quad ring[kRing][9];
    Quad acc[9];
    std::atomic<int> posted{0}, consumed{0}, accPosted{0};
};

struct SimIO {
    const double *x, *u, *p;
    double *f, *J;
    Channel* ch;
    Quad qb(int i) const { return Quad{x[i]}; }
    Quad vb(int i) const { return Quad{x[19 + i]}; }
    Quad ql(int i) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = x[7 + 3 * l + i]; return r; }
    Quad vl(int i) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = x[25 + 3 * l + i]; return r; }
    Quad ul(int i) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = u[3 * l + i]; return r; }
    Quad dt() const { return Quad{p[0]}; }
    void phase() const {}
    void keep(const Quad&) const {}
    mutable Quad slots[512];
    Quad ld(int s) const { return slots[s]; }
    void st(int s, const Quad& v) const { slots[s] = v; }
    mutable double uslots[512];
    Quad ldu(int s) const { return Quad{uslots[s]}; }
    void stu(int s, const Quad& v) const { uslots[s] = (v.v[0] == v.v[1] && v.v[1] == v.v[2] && v.v[2] == v.v[3]) ? v.v[0] : NAN; }
    Quad c(int k) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = ungar_amd::gen::anymal_quad::kLegConstants[k][l]; return r; }
    Quad quad_sum(const Quad& a) const { return Quad{a.v[0] + a.v[1] + a.v[2] + a.v[3]}; }
    Quad rot(const Quad& a, int r) const { Quad o; for (int l = 0; l < 4; ++l) o.v[l] = a.v[(l + r) & 3]; return o; }
    Quad quad_rot1(const Quad& a) const { return rot(a, 1); }
    Quad quad_rot2(const Quad& a) const { return rot(a, 2); }
    Quad quad_rot3(const Quad& a) const { return rot(a, 3); }
    // ---- channel (same protocol as QuadSplitIO) ----
    static void Spin(const std::atomic<int>& flag, int atLeast) {
        while (flag.load(std::memory_order_acquire) < atLeast) std::this_thread::yield();
    }
    void wait_free(int m) const { if (m >= kRing) Spin(ch->consumed, m - kRing + 1); }
    void send(int m, int i, const Quad& v) const { ch->ring[m % kRing][i] = v; }
    void post(int m) const { ch->posted.store(m + 1, std::memory_order_release); }
    void wait_acc() const { Spin(ch->accPosted, 1); }
    Quad acc(int k) const { return ch->acc[k]; }
    void wait(int m) const { Spin(ch->posted, m + 1); }
    Quad recv(int m, int i) const { return ch->ring[m % kRing][i]; }
    void done(int m) const { ch->consumed.store(m + 1, std::memory_order_release); }
    void send_acc(int k, const Quad& v) const { ch->acc[k] = v; }
    void post_acc() const { ch->accPosted.store(1, std::memory_order_release); }
    // ---- sinks (dense addressing) ----
    void f_base(int row, const Quad& v) const { f[row] = v.v[row & 3]; }
    void f_leg(int rowBase, const Quad& v) const { for (int l = 0; l < 4; ++l) f[rowBase + 3 * l] = v.v[l]; }
    void j_leg(int rowBase, int colBase, int legMul, int rot_, int, int, int, int, const Quad& v) const {
        for (int l = 0; l < 4; ++l) J[(rowBase + 3 * l) * 49 + colBase + 3 * legMul * ((l + rot_) & 3)] = v.v[l];
    }
    void j_base_own(int row, int colBase, int legMul, int, int, int, int, int, const Quad& v) const {
        for (int l = 0; l < 4; ++l) J[row * 49 + colBase + 3 * legMul * l] = v.v[l];
    }
    void j_base_shared(int row, int colBase, int, int, int, int, int, int, const Quad& v) const { J[row * 49 + colBase] = v.v[(row + colBase) & 3]; }
    void j_leg2(int row, int row2, int colBase, int legMul, int rot_, int, int, int, int, int, int, int, int, const Quad& v, const Quad& v2) const {
        if (row2 != row + 18) std::abort();
        j_leg(row, colBase, legMul, rot_, 0, 0, 0, 0, v);
        j_leg(row2, colBase, legMul, rot_, 0, 0, 0, 0, v2);
    }
    void j_base_own2(int row, int row2, int colBase, int legMul, int rot_, int, int, int, int, int, int, int, int, const Quad& v, const Quad& v2) const {
        if (row2 != row + 1) std::abort();
        j_base_own(row, colBase, legMul, rot_, 0, 0, 0, 0, v);
        j_base_own(row2, colBase, legMul, rot_, 0, 0, 0, 0, v2);
    }
    void j_base_shared2(int row, int row2, int colBase, int legMul, int rot_, int, int, int, int, int, int, int, int, const Quad& v, const Quad& v2) const {
        if (row2 != row + 1) std::abort();
        j_base_shared(row, colBase, legMul, rot_, 0, 0, 0, 0, v);
        j_base_shared(row2, colBase, legMul, rot_, 0, 0, 0, 0, v2);
    }
};

}  // namespace

extern "C" void anymal_split_sim(const double* x, const double* u, const double* p, double* f, double* J) {
    for (int i = 0; i < 37; ++i) f[i] = NAN;
    for (int i = 0; i < 37 * 49; ++i) J[i] = NAN;  // every entry must be written by the consumer
    static Channel ch;  // (large: not on the stack)
    ch.posted = 0;
    ch.consumed = 0;
    ch.accPosted = 0;
    static SimIO producer, consumer;
    producer = SimIO{x, u, p, nullptr, nullptr, &ch};
    consumer = SimIO{x, u, p, f, J, &ch};
    std::thread first([] { ungar_amd::gen::anymal_split::ProducerQuad<Quad>(producer); });
    std::thread second([] { ungar_amd::gen::anymal_split::ConsumerQuad<Quad>(consumer); });
    first.join();
    second.join();
}
