// CPU 4-lane simulator of the lane-per-leg inertia-matrix program (ungar_amd/csrc/gen/anymal_crba_quad_gen.hpp; T = one value per lane of a
// quad): the sinks scatter into the node-level M[324] and, through the per-leg CSR indices every Jacobian sink carries, into the value array of
// d M / d q.  Built by tests/test_quad_program.py with g++ (no GPU).
#include <cmath>

#include "anymal_crba_quad_gen.hpp"

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
    const double* x;
    double *y, *Js;
    Quad ql(int i) const {
        Quad r;
        for (int l = 0; l < 4; ++l) r.v[l] = x[7 + 3 * l + i];
        return r;
    }
    Quad c(int k) const {
        Quad r;
        for (int l = 0; l < 4; ++l) r.v[l] = ungar_amd::gen::anymal_crba_quad::kLegConstants[k][l];
        return r;
    }
    void phase() const {}
    void keep(const Quad&) const {}
    mutable Quad slots[512];
    Quad ld(int s) const { return slots[s]; }
    void st(int s, const Quad& v) const { slots[s] = v; }
    Quad quad_sum(const Quad& a) const { return Quad{a.v[0] + a.v[1] + a.v[2] + a.v[3]}; }
    void f_base(int idx, const Quad& v) const { y[idx] = v.v[idx & 3]; }
    void f_bl(int r, int j, const Quad& v) const {
        for (int l = 0; l < 4; ++l) y[r * 18 + 6 + 3 * l + j] = v.v[l];
    }
    void f_lb(int j, int r, const Quad& v) const {
        for (int l = 0; l < 4; ++l) y[(6 + 3 * l + j) * 18 + r] = v.v[l];
    }
    void f_ll(int i, int j, int rot, const Quad& v) const {
        for (int l = 0; l < 4; ++l) y[(6 + 3 * l + i) * 18 + 6 + 3 * ((l + rot) & 3) + j] = v.v[l];
    }
    void j_sparse(int k0, int k1, int k2, int k3, const Quad& v) const {
        const int ks[4] = {k0, k1, k2, k3};
        for (int l = 0; l < 4; ++l)
            if (ks[l] >= 0) Js[ks[l]] = v.v[l];
    }
};

}  // namespace

extern "C" void anymal_crba_quad_sim(const double* x, double* y, double* Js, int nnz) {
    for (int i = 0; i < 324; ++i) y[i] = NAN;  // every entry must be written by the program
    for (int i = 0; i < nnz; ++i) Js[i] = NAN;
    SimIO io{x, y, Js};
    ungar_amd::gen::anymal_crba_quad::ValueJacobianQuad<Quad>(io);
}
