// CPU 4-lane simulator of the lane-per-leg joint-torque program (ungar_amd/csrc/gen/anymal_rnea_quad_gen.hpp): the generated body is
// generic over the value type; here T = Quad (one value per lane of a quad), quad_sum is a loop, and the sinks scatter into the
// node-level y[18] / J[18][55] (and the CSR value array through the per-lane indices every sink carries).  Built by
// tests/test_quad_program.py with g++ (no GPU): pins the SPMD math -- ownership of rows / columns, the base-wrench sums, the
// structural zeros -- against the oracle's golden vectors before the same text is compiled for gfx950.
#include <cmath>

#ifdef QUAD_SIM_CENTROIDAL  // the centroidal-momentum program runs in the same skeleton: 6 base rows x 37 columns of x, no u
#include "anymal_centroidal_quad_gen.hpp"
namespace quad_gen = ungar_amd::gen::anymal_centroidal_quad;
constexpr int kSimRows = 6, kSimCols = 37;
#else
#include "anymal_rnea_quad_gen.hpp"
namespace quad_gen = ungar_amd::gen::anymal_rnea_quad;
constexpr int kSimRows = 18, kSimCols = 55;
#endif

namespace {

struct Quad {
    double v[4];
    Quad() : v{0, 0, 0, 0} {}
    Quad(double s) : v{s, s, s, s} {}  // NOLINT
};
#define QUAD_BIN(op)                                                   \
    inline Quad operator op(const Quad& a, const Quad& b) {           \
        Quad r;                                                        \
        for (int l = 0; l < 4; ++l) r.v[l] = a.v[l] op b.v[l];         \
        return r;                                                      \
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
#define QUAD_FN(fn)                                          \
    inline Quad fn(const Quad& a) {                          \
        Quad r;                                              \
        for (int l = 0; l < 4; ++l) r.v[l] = std::fn(a.v[l]); \
        return r;                                            \
    }
QUAD_FN(sin)
QUAD_FN(cos)
QUAD_FN(sqrt)

struct SimIO {
    const double *x, *u;
    double *y, *J, *Jsparse;
    Quad perLeg(const double* base, int i) const {
        Quad r;
        for (int l = 0; l < 4; ++l) r.v[l] = base[3 * l + i];
        return r;
    }
    Quad qb(int i) const { return Quad{x[i]}; }
    Quad vb(int i) const { return Quad{x[19 + i]}; }
    Quad ab(int i) const { return Quad{u[i]}; }
    Quad ql(int i) const { return perLeg(x + 7, i); }
    Quad vl(int i) const { return perLeg(x + 25, i); }
    Quad al(int i) const { return perLeg(u + 6, i); }
    Quad c(int k) const {
        Quad r;
        for (int l = 0; l < 4; ++l) r.v[l] = quad_gen::kLegConstants[k][l];
        return r;
    }
    void phase() const {}
    void keep(const Quad&) const {}
    mutable Quad slots[512];
    Quad ld(int s) const { return slots[s]; }
    void st(int s, const Quad& v) const { slots[s] = v; }
    // compact slots hold ONE copy per quad: a value that is not identical in the four lanes poisons everything downstream
    mutable double uslots[512];
    Quad ldu(int s) const { return Quad{uslots[s]}; }
    void stu(int s, const Quad& v) const { uslots[s] = (v.v[0] == v.v[1] && v.v[1] == v.v[2] && v.v[2] == v.v[3]) ? v.v[0] : NAN; }
    Quad quad_sum(const Quad& a) const { return Quad{a.v[0] + a.v[1] + a.v[2] + a.v[3]}; }
    void f_base(int row, const Quad& v) const { y[row] = v.v[row & 3]; }
    void f_leg(int rowBase, const Quad& v) const {
        for (int l = 0; l < 4; ++l) y[rowBase + 3 * l] = v.v[l];
    }
    void put(int r, int c, int k, double v) const {
        J[r * kSimCols + c] = v;
        if (Jsparse && k >= 0) Jsparse[k] = v;
    }
    void j_leg(int rowBase, int colBase, int legMul, int rot, int k0, int k1, int k2, int k3, const Quad& v) const {
        const int ks[4] = {k0, k1, k2, k3};
        for (int l = 0; l < 4; ++l) put(rowBase + 3 * l, colBase + 3 * legMul * ((l + rot) & 3), ks[l], v.v[l]);
    }
    void j_base_own(int row, int colBase, int k0, int k1, int k2, int k3, const Quad& v) const {
        const int ks[4] = {k0, k1, k2, k3};
        for (int l = 0; l < 4; ++l) put(row, colBase + 3 * l, ks[l], v.v[l]);
    }
    void j_base_shared(int row, int col, int k, const Quad& v) const { put(row, col, k, v.v[(row + col) & 3]); }
};

}  // namespace

extern "C" void anymal_rnea_quad_sim(const double* x, const double* u, double* y, double* J, double* Jsparse, int nnz) {
    for (int i = 0; i < kSimRows; ++i) y[i] = NAN;
    for (int i = 0; i < kSimRows * kSimCols; ++i) J[i] = NAN;  // every entry must be written by the program
    for (int i = 0; i < nnz; ++i) Jsparse[i] = NAN;
    SimIO io{x, u, y, J, nnz > 0 ? Jsparse : nullptr};
    quad_gen::ValueJacobianQuad<Quad>(io);
}
