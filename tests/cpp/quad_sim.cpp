// CPU 4-lane simulator of the lane-per-leg SPMD program (ungar_amd/csrc/gen/anymal_quad_gen.hpp).
// The generated body is generic over the value type: here T = Quad (one value per lane of a quad),
// quad_sum / quad_rot are plain loops, and the sinks scatter into node-level f[37] / J[37][49].
// Built by tests/test_quad_program.py with g++ (no GPU needed): validates the SPMD MATH -- block-arrow
// factorisation, ownership of rows/columns, rotations -- against the oracle's golden vectors before
// the same text is compiled for gfx950.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "anymal_quad_gen.hpp"
#include "anymal_tiles_gen.hpp"

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
    const double *x, *u, *p;
    double *f, *J;
    Quad qb(int i) const { return Quad{x[i]}; }
    Quad vb(int i) const { return Quad{x[19 + i]}; }
    Quad ql(int i) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = x[7 + 3 * l + i]; return r; }
    Quad vl(int i) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = x[25 + 3 * l + i]; return r; }
    Quad ul(int i) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = u[3 * l + i]; return r; }
    Quad dt() const { return Quad{p[0]}; }
    void phase() const {}
    void keep(const Quad&) const {}
    Quad fma(const Quad& a, const Quad& b, const Quad& c) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = std::fma(a.v[l], b.v[l], c.v[l]); return r; }
    // tile program: image s holds entry kEntryOfSlot[4 s + l] in lane l; t_put2 stores images 2 p and 2 p + 1; sel4 keeps the l-th of four quad-uniform values in lane l
    const short* tileTable = nullptr;
    Quad sel4(const Quad& v0, const Quad& v1, const Quad& v2, const Quad& v3) const { Quad r; r.v[0] = v0.v[0]; r.v[1] = v1.v[1]; r.v[2] = v2.v[2]; r.v[3] = v3.v[3]; return r; }
    void t_put2(int pair, const Quad& v, const Quad& v2) const {
        const Quad* vs[2] = {&v, &v2};
        for (int h = 0; h < 2; ++h)
            for (int l = 0; l < 4; ++l) {
                const int e = tileTable[4 * (2 * pair + h) + l];
                if (e < 0) continue;
                if (!std::isnan(J[e])) std::abort();  // every entry exactly once
                J[e] = vs[h]->v[l];
            }
    }
    mutable Quad slots[512];
    Quad ld(int s) const { return slots[s]; }
    void st(int s, const Quad& v) const { slots[s] = v; }
    // compact slots hold ONE copy per quad: only legal for values identical in the four lanes -- a value
    // that is not poisons everything downstream, so the golden-vector test catches a wrong uniformity analysis
    mutable double uslots[512];
    Quad ldu(int s) const { return Quad{uslots[s]}; }
    void stu(int s, const Quad& v) const { uslots[s] = (v.v[0] == v.v[1] && v.v[1] == v.v[2] && v.v[2] == v.v[3]) ? v.v[0] : NAN; }
    Quad c(int k) const { Quad r; for (int l = 0; l < 4; ++l) r.v[l] = ungar_amd::gen::anymal_quad::kLegConstants[k][l]; return r; }
    Quad quad_sum(const Quad& a) const { return Quad{a.v[0] + a.v[1] + a.v[2] + a.v[3]}; }
    Quad rot(const Quad& a, int r) const { Quad o; for (int l = 0; l < 4; ++l) o.v[l] = a.v[(l + r) & 3]; return o; }
    Quad quad_rot1(const Quad& a) const { return rot(a, 1); }
    Quad quad_rot2(const Quad& a) const { return rot(a, 2); }
    Quad quad_rot3(const Quad& a) const { return rot(a, 3); }
    void f_base(int row, const Quad& v) const { f[row] = v.v[row & 3]; }
    void f_leg(int rowBase, const Quad& v) const { for (int l = 0; l < 4; ++l) f[rowBase + 3 * l] = v.v[l]; }
    // every sink carries both addressings: dense (row, col) and the CSR index k per lane (-1 = structural zero)
    double* Jsparse = nullptr;
    void put(int l, int r, int c, int k, double v) const {
        J[r * 49 + c] = v;
        if (Jsparse && k >= 0) Jsparse[k] = v;
    }
    void j_leg(int rowBase, int colBase, int legMul, int rot_, int k0, int k1, int k2, int k3, const Quad& v) const {
        const int ks[4] = {k0, k1, k2, k3};
        for (int l = 0; l < 4; ++l) put(l, rowBase + 3 * l, colBase + 3 * legMul * ((l + rot_) & 3), ks[l], v.v[l]);
    }
    void j_base_own(int row, int colBase, int legMul, int, int k0, int k1, int k2, int k3, const Quad& v) const {
        const int ks[4] = {k0, k1, k2, k3};
        for (int l = 0; l < 4; ++l) put(l, row, colBase + 3 * legMul * l, ks[l], v.v[l]);
    }
    void j_base_shared4(int r0, int r1, int r2, int r3, int col, int k0, int k1, int k2, int k3, const Quad& v0, const Quad& v1, const Quad& v2,
                        const Quad& v3) const {
        const int rs[4] = {r0, r1, r2, r3}, ks[4] = {k0, k1, k2, k3};
        const Quad* vs[4] = {&v0, &v1, &v2, &v3};
        for (int l = 0; l < 4; ++l) put(l, rs[l], col, ks[l], vs[l]->v[l]);  // the lane of leg l writes entry l
    }
    void j_base_shared(int row, int colBase, int, int, int k0, int, int, int, const Quad& v) const { put(0, row, colBase, k0, v.v[(row + colBase) & 3]); }
    // paired sinks (two entries of a column per statement: one 16-byte store on the GPU): here the two entries one after the other
    void j_leg2(int row, int row2, int colBase, int legMul, int rot_, int k0, int k1, int k2, int k3, int m0, int m1, int m2, int m3, const Quad& v, const Quad& v2) const {
        if (row2 != row + 18) std::abort();  // the odd lanes' offset of the paired store assumes this distance
        j_leg(row, colBase, legMul, rot_, k0, k1, k2, k3, v);
        j_leg(row2, colBase, legMul, rot_, m0, m1, m2, m3, v2);
    }
    void j_base_own2(int row, int row2, int colBase, int legMul, int rot_, int k0, int k1, int k2, int k3, int m0, int m1, int m2, int m3, const Quad& v,
                     const Quad& v2) const {
        if (row2 != row + 1) std::abort();
        j_base_own(row, colBase, legMul, rot_, k0, k1, k2, k3, v);
        j_base_own(row2, colBase, legMul, rot_, m0, m1, m2, m3, v2);
    }
    void j_base_shared2(int row, int row2, int colBase, int legMul, int rot_, int k0, int k1, int k2, int k3, int m0, int m1, int m2, int m3, const Quad& v,
                        const Quad& v2) const {
        if (row2 != row + 1) std::abort();
        j_base_shared(row, colBase, legMul, rot_, k0, k1, k2, k3, v);
        j_base_shared(row2, colBase, legMul, rot_, m0, m1, m2, m3, v2);
    }
};

}  // namespace

extern "C" void anymal_quad_sim_sparse(const double* x, const double* u, const double* p, double* f, double* J, double* Jsparse, int nnz);

extern "C" void anymal_quad_sim(const double* x, const double* u, const double* p, double* f, double* J) {
    anymal_quad_sim_sparse(x, u, p, f, J, nullptr, 0);
}

extern "C" void anymal_quad_sim_sparse(const double* x, const double* u, const double* p, double* f, double* J, double* Jsparse, int nnz) {
    for (int i = 0; i < 37; ++i) f[i] = NAN;
    for (int i = 0; i < 37 * 49; ++i) J[i] = NAN;  // every entry must be written by the program
    for (int i = 0; i < nnz; ++i) Jsparse[i] = NAN;
    SimIO io{x, u, p, f, J};
    io.Jsparse = Jsparse;
    ungar_amd::gen::anymal_quad::ValueJacobianQuad<Quad>(io);
}

/// The value-only program (the value sinks of the same recording): what forward_zero launches for this model.
extern "C" void anymal_quad_sim_value(const double* x, const double* u, const double* p, double* f) {
    for (int i = 0; i < 37; ++i) f[i] = NAN;
    double unused[1] = {0.0};
    SimIO io{x, u, p, f, unused};
    ungar_amd::gen::anymal_quad::ValueQuad<Quad>(io);
}

/// The tile program (quad_leg_program.hpp: tileStores): the same node through register images, read back through the generated slot table.
extern "C" void anymal_quad_sim_tiles(const double* x, const double* u, const double* p, double* f, double* J) {
    for (int i = 0; i < 37; ++i) f[i] = NAN;
    for (int i = 0; i < 37 * 49; ++i) J[i] = NAN;
    SimIO io{x, u, p, f, J};
    io.tileTable = ungar_amd::gen::anymal_tiles::kEntryOfSlot;
    ungar_amd::gen::anymal_tiles::ValueJacobianQuadTiles<Quad>(io);
}
