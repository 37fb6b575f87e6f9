// ungar_amd :: ahead-of-time code generator for the built-in shooting-node models.
//
// Plays the part of FunctionFactory::Worker::CreateModelsImpl (reference
// include/ungar/autodiff/function.hpp:453-503) for the four BASELINE workloads: record the node
// function on AD scalars, derive the sparse Jacobian over the (x,u) columns, and lower the tape
// to (a) a HIP device-function body per model  -> ungar_amd/csrc/gen/<model>_gen.hpp
//    (b) with --c-oracle DIR, plain C for the CPU baseline/checker -> DIR/<model>_cg.c
// Run by __graft_entry__.build(); the HIP output is compiled by hipcc into libungar_amd.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../models/nodes.hpp"
#include "../models/rbd_nodes.hpp"
#include "../rbd/rnea_crba.hpp"
#include "../tape/emit.hpp"
#include "quad_leg_program.hpp"
#include "quad_centroidal_program.hpp"
#include "quad_crba_program.hpp"
#include "quad_rnea_program.hpp"

using namespace ungar_amd;
using tape::AD;

namespace {

struct NodeSpec {
    models::NodeDims dims;
    std::function<void(const AD*, const AD*, const AD*, const AD*, AD*)> fn;
    bool valueOnly = false;  // no Jacobian program (ungar_model_has_sparse_jacobian() == 0)
    int phasedLdsSlots = 0;  // > 0: additionally emit the phased body with this many per-lane LDS home slots
    int jacMode = 0;         // 0 = fewer statements decides; 1 = forward, 2 = reverse accumulation
    // The three settings above were measured per model on MI355X (tools/run_rbd_variants.sh, profiles/archive/r02k_rbd_variants.log):
    // phases are Jacobian columns for forward programs and Jacobian rows (one adjoint sweep each) for reverse ones.
};

struct Generated {
    models::NodeDims dims;
    tape::Tape tape;
    tape::SparseEntries jac;
    int jacMode = 0;
};

Generated Record(const NodeSpec& spec, int jacMode) {
    const auto& d = spec.dims;
    const int nIn = d.nx + d.nu + d.nw + d.np;
    std::vector<AD> in = tape::Independent(nIn);
    std::vector<AD> out(static_cast<std::size_t>(d.Ny()));
    spec.fn(in.data(), in.data() + d.nx, in.data() + d.nx + d.nu, in.data() + d.nx + d.nu + d.nw, out.data());
    Generated g{d, tape::MakeTape(out), {}, 0};
    if (spec.valueOnly) {
        g.jac.rows = d.Ny();
        g.jac.cols = d.nx + d.nu;
        return g;
    }
    tape::Differentiator diff{g.tape};
    g.jac = diff.Jacobian(d.nx + d.nu, jacMode);
    g.jacMode = diff.LastMode();
    return g;
}

/// Structured floating-base node: same function and same Jacobian pattern as the
/// taped-ABA version `ad`, but the derivative expressions come from implicit differentiation of
///     M(q) a + h(q, v) = [0; u]
/// i.e.  da/du = M^-1 S^T,  da/dq_j = -M^-1 dRNEA(q,v,a)/dq_j,  da/dv = -M^-1 dRNEA/dv, and
/// da_base/dquat = -d(R^T a0)/dquat in closed form, composed with the integrator by the chain rule.
Generated RecordFloatingBaseStructured(const rbd::Model& model, const Generated& ad, const char* name, int dModeRequest) {
    const int nq = model.nq, nv = model.nv, nx = nq + nv, nu = nv - 6;
    const int nIn = nx + nu + 1;  // + dt
    std::vector<AD> in = tape::Independent(nIn + nv);  // aux inputs: a_in
    const AD* x = in.data();
    const AD dt = in[static_cast<std::size_t>(nx + nu)];
    std::vector<AD> q(in.begin(), in.begin() + nq), v(in.begin() + nq, in.begin() + nx), ain(in.begin() + nIn, in.end());
    std::vector<AD> tau(static_cast<std::size_t>(nv), AD{0.0}), zero(static_cast<std::size_t>(nv), AD{0.0});
    for (int k = 0; k < nu; ++k) tau[static_cast<std::size_t>(6 + k)] = in[static_cast<std::size_t>(nx + k)];

    tape::Graph& g = tape::CurrentGraph();
    std::vector<tape::Id> inputIds;
    for (const AD& i : in) inputIds.push_back(i.Node());
    tape::Differentiator diff{g, inputIds};

    // ---- primal acceleration: a = M^-1 (tau - h) ---------------------------------------------------
    const auto liMi = rbd::JointPlacements(model, q);
    const auto M = rbd::Crba(model, liMi);
    const auto F = rbd::FactorUdut(M);
    const std::vector<AD> h = rbd::Rnea(model, liMi, v, zero, true);
    std::vector<AD> rhs(static_cast<std::size_t>(nv));
    for (int k = 0; k < nv; ++k) rhs[static_cast<std::size_t>(k)] = tau[static_cast<std::size_t>(k)] - h[static_cast<std::size_t>(k)];
    const std::vector<AD> a = rbd::SolveUdut(F, rhs);

    // ---- stage functions of (x, a_in) and their partial derivatives ----------------------------------
    const std::vector<AD> tauId = rbd::Rnea(model, liMi, v, ain, true);
    std::vector<AD> gx(static_cast<std::size_t>(nx));
    models::IntegrateFloatingBase(model, x, ain.data(), dt, gx.data());
    std::vector<tape::Id> tauIds, gIds;
    for (const AD& t : tauId) tauIds.push_back(t.Node());
    for (const AD& t : gx) gIds.push_back(t.Node());
    std::vector<int> qvCols;  // joint angles and all velocities
    for (int k = 7; k < nx; ++k) qvCols.push_back(k);
    const tape::SparseEntries D = diff.Jacobian(tauIds, qvCols, dModeRequest);
    const int dMode = diff.LastMode();
    std::vector<int> gCols;
    for (int k = 0; k < nx; ++k) gCols.push_back(k);
    for (int k = 0; k < nv; ++k) gCols.push_back(nIn + k);
    const tape::SparseEntries G = diff.Jacobian(gIds, gCols);
    // gravity seen from the base: gamma = R(quat)^T a0, a0 = -gravity
    std::vector<tape::Id> gammaIds;
    for (std::size_t k = 0; k < 3; ++k) {
        AD acc{0.0};
        for (std::size_t r = 0; r < 3; ++r) acc = acc + liMi[1].R[r][k] * (-model.gravity[r]);
        gammaIds.push_back(acc.Node());
    }
    const tape::SparseEntries dGamma = diff.Jacobian(gammaIds, std::vector<int>{3, 4, 5, 6});

    // substitute a_in := a in everything that was built on the auxiliary inputs
    std::vector<std::pair<int, tape::Id>> sub;
    for (int k = 0; k < nv; ++k) sub.emplace_back(nIn + k, a[static_cast<std::size_t>(k)].Node());
    const std::vector<tape::Id> Dv = diff.Substitute(D.value, sub), Gv = diff.Substitute(G.value, sub), fv = diff.Substitute(gIds, sub);

    // ---- A = da/d(x,u), column by column (nv x (nx+nu), zero where untouched) ---------------------------
    const int ncols = nx + nu;
    std::vector<std::vector<AD>> A(static_cast<std::size_t>(ncols), std::vector<AD>(static_cast<std::size_t>(nv), AD{0.0}));
    for (std::size_t e = 0; e < dGamma.Nnz(); ++e)  // quaternion columns, base-linear rows
        A[static_cast<std::size_t>(3 + dGamma.col[e])][static_cast<std::size_t>(dGamma.row[e])] = -AD::FromId(dGamma.value[e]);
    for (std::size_t cj = 0; cj < qvCols.size(); ++cj) {
        std::vector<AD> r(static_cast<std::size_t>(nv), AD{0.0});
        bool any = false;
        for (std::size_t e = 0; e < D.Nnz(); ++e)
            if (D.col[e] == static_cast<int>(cj)) {
                r[static_cast<std::size_t>(D.row[e])] = -AD::FromId(Dv[e]);
                any = true;
            }
        if (any) A[static_cast<std::size_t>(qvCols[cj])] = rbd::SolveUdut(F, r);
    }
    for (int k = 0; k < nu; ++k) {
        std::vector<AD> r(static_cast<std::size_t>(nv), AD{0.0});
        r[static_cast<std::size_t>(6 + k)] = AD{1.0};
        A[static_cast<std::size_t>(nx + k)] = rbd::SolveUdut(F, r);
    }

    // ---- J = G_x + G_a A ---------------------------------------------------------------------------------
    std::vector<std::vector<AD>> J(static_cast<std::size_t>(nx), std::vector<AD>(static_cast<std::size_t>(ncols), AD{0.0}));
    for (std::size_t e = 0; e < G.Nnz(); ++e) {
        const std::size_t r = static_cast<std::size_t>(G.row[e]);
        const int gc = G.col[e];
        const AD ge = AD::FromId(Gv[e]);
        if (gc < nx) {
            J[r][static_cast<std::size_t>(gc)] = J[r][static_cast<std::size_t>(gc)] + ge;
        } else {
            const std::size_t k = static_cast<std::size_t>(gc - nx);  // row of A
            for (int c = 0; c < ncols; ++c) J[r][static_cast<std::size_t>(c)] = J[r][static_cast<std::size_t>(c)] + ge * A[static_cast<std::size_t>(c)][k];
        }
    }

    // ---- package with the taped-ABA pattern so that both kernels are interchangeable -----------------------
    std::vector<AD> outs;
    for (tape::Id id : fv) outs.push_back(AD::FromId(id));
    std::vector<std::vector<char>> inPattern(static_cast<std::size_t>(nx), std::vector<char>(static_cast<std::size_t>(ncols), 0));
    for (std::size_t e = 0; e < ad.jac.Nnz(); ++e) {
        inPattern[static_cast<std::size_t>(ad.jac.row[e])][static_cast<std::size_t>(ad.jac.col[e])] = 1;
        outs.push_back(J[static_cast<std::size_t>(ad.jac.row[e])][static_cast<std::size_t>(ad.jac.col[e])]);
    }
    for (int r = 0; r < nx; ++r)
        for (int c = 0; c < ncols; ++c)
            if (!inPattern[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)]) {
                const AD& e = J[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)];
                if (!(e.IsLiteral() && e.Literal() == 0.0)) {
                    std::fprintf(stderr, "structured Jacobian has an entry (%d,%d) outside the taped-ABA pattern\n", r, c);
                    std::exit(1);
                }
            }
    Generated out{ad.dims, tape::MakeTape(outs), ad.jac, 10 + dMode};
    out.dims.name = name;
    for (std::size_t e = 0; e < out.jac.Nnz(); ++e) out.jac.value[e] = out.tape.outputs[static_cast<std::size_t>(nx) + e];
    out.tape.outputs.resize(static_cast<std::size_t>(nx));
    return out;
}

std::vector<std::string> InputNames(const models::NodeDims& d, bool cDialect) {
    std::vector<std::string> names;
    auto add = [&](const char* base, int n) {
        for (int i = 0; i < n; ++i)
            names.push_back(cDialect ? std::string(base) + "[" + std::to_string(i) + "]" : std::string(base) + std::to_string(i));
    };
    add("x", d.nx);
    add("u", d.nu);
    add("w", d.nw);
    add("p", d.np);
    add("aux_unused", 64);  // auxiliary inputs of staged recordings are substituted away before emission
    return names;
}

/// Which inputs does the sub-DAG under `roots` read?
std::vector<char> UsedInputs(const tape::Graph& g, const std::vector<tape::Id>& roots) {
    std::vector<char> live(g.Size(), 0), used(static_cast<std::size_t>(g.NumInputs()), 0);
    for (tape::Id r : roots) live[static_cast<std::size_t>(r)] = 1;
    for (std::size_t i = g.Size(); i-- > 0;) {
        if (!live[i]) continue;
        const tape::Node& nd = g.At(static_cast<tape::Id>(i));
        if (nd.op == tape::Op::Input) {
            used[static_cast<std::size_t>(nd.a)] = 1;
            continue;
        }
        if (nd.op == tape::Op::Const) continue;
        for (tape::Id o : {nd.a, nd.b, nd.c, nd.d})
            if (o != tape::kNoId) live[static_cast<std::size_t>(o)] = 1;
    }
    return used;
}

std::string HipPrologue(const models::NodeDims& d, const std::vector<char>& used) {
    std::ostringstream os;
    int k = 0;
    auto block = [&](const char* base, int n) {
        for (int i = 0; i < n; ++i, ++k)
            if (used[static_cast<std::size_t>(k)]) os << "    const double " << base << i << " = io." << base << "(" << i << ");\n";
    };
    block("x", d.nx);
    block("u", d.nu);
    block("w", d.nw);
    block("p", d.np);
    return os.str();
}

void EmitHip(const Generated& g, const std::string& dir, bool columnMajorEmission = false, int ldsSlots = 0, int rematConsumers = 2, int rematDepth = 3, int prefetch = 48,
             bool phaseByRow = false) {
    const auto& d = g.dims;
    const std::string name = d.name;
    std::ostringstream os;
    os << "// GENERATED by ungar_amd/csrc/codegen/codegen_main.cpp -- do not edit.\n"
       << "// Shooting-node model '" << name << "': straight-line value / value+Jacobian bodies lowered from the\n"
       << "// recorded tape; the surrounding kernels are hand-written (csrc/kernels/node_kernel.hpp).\n"
       << "#pragma once\n#include <hip/hip_runtime.h>\n\n"
       << "namespace ungar_amd::gen::" << name << " {\n\n"
       << "inline constexpr int kNx = " << d.nx << ", kNu = " << d.nu << ", kNw = " << d.nw << ", kNp = " << d.np << ";\n"
       << "inline constexpr int kJacRows = " << g.jac.rows << ", kJacCols = " << g.jac.cols << ", kJacNnz = " << g.jac.Nnz() << ";\n"
       << "inline constexpr int kJacMode = " << g.jacMode << ";  // 1 = forward, 2 = reverse accumulation\n";
    os << "inline constexpr int kJacRow[kJacNnz > 0 ? kJacNnz : 1] = {";
    for (std::size_t k = 0; k < g.jac.Nnz(); ++k) os << (k ? "," : "") << g.jac.row[k];
    os << (g.jac.Nnz() ? "" : "0") << "};\ninline constexpr int kJacCol[kJacNnz > 0 ? kJacNnz : 1] = {";
    for (std::size_t k = 0; k < g.jac.Nnz(); ++k) os << (k ? "," : "") << g.jac.col[k];
    os << (g.jac.Nnz() ? "" : "0") << "};\n\n";

    // Value only.
    {
        std::vector<tape::OutputSlot> slots;
        for (int i = 0; i < d.Ny(); ++i) slots.push_back({g.tape.outputs[static_cast<std::size_t>(i)], "io.f(" + std::to_string(i) + ", %s);"});
        tape::Emitter em{g.tape.graph, InputNames(d, false)};
        const std::string body = em.Emit(slots);
        os << "// " << em.Stats().statements << " statements, " << em.Stats().flops << " flops, " << em.Stats().transcendentals
           << " transcendentals, " << em.Stats().divisions << " divisions\n"
           << "template <class IO>\n__device__ __forceinline__ void Value(IO& io) {\n"
           << HipPrologue(d, UsedInputs(g.tape.graph, g.tape.outputs)) << body << "}\n\n";
    }
    // Value + Jacobian.  Jacobian sinks carry (k, row, col) so that an IO policy can store the
    // structural non-zeros (CSR value order) or scatter into the dense [A|B] block.
    {
        std::vector<tape::OutputSlot> slots;
        std::vector<tape::Id> roots = g.tape.outputs;
        auto jslot = [&](std::size_t k) {
            slots.push_back({g.jac.value[k],
                             "io.j(" + std::to_string(k) + ", " + std::to_string(g.jac.row[k]) + ", " + std::to_string(g.jac.col[k]) + ", %s);"});
            roots.push_back(g.jac.value[k]);
        };
        if (columnMajorEmission) {
            // values first (they need the primal solve), then one Jacobian column after the other: each
            // column is an independent solve + chain rule, so its temporaries die before the next starts
            for (int i = 0; i < d.Ny(); ++i) slots.push_back({g.tape.outputs[static_cast<std::size_t>(i)], "io.f(" + std::to_string(i) + ", %s);"});
            for (int c = 0; c < g.jac.cols; ++c)
                for (std::size_t k = 0; k < g.jac.Nnz(); ++k)
                    if (g.jac.col[k] == c) jslot(k);
        } else {
            std::size_t k = 0;
            for (int i = 0; i < d.Ny(); ++i) {
                slots.push_back({g.tape.outputs[static_cast<std::size_t>(i)], "io.f(" + std::to_string(i) + ", %s);"});
                for (; k < g.jac.Nnz() && g.jac.row[k] == i; ++k) jslot(k);
            }
        }
        tape::Emitter em{g.tape.graph, InputNames(d, false)};
        const std::string body = em.Emit(slots);
        os << "// " << em.Stats().statements << " statements, " << em.Stats().flops << " flops, " << em.Stats().transcendentals
           << " transcendentals, " << em.Stats().divisions << " divisions\n"
           << "inline constexpr int kJacStatements = " << em.Stats().statements << ", kJacFlops = " << em.Stats().flops << ";\n"
           << "template <class IO>\n__device__ __forceinline__ void ValueJacobian(IO& io) {\n"
           << HipPrologue(d, UsedInputs(g.tape.graph, roots)) << body << "}\n\n";
        std::fprintf(stderr, "[codegen] %-10s nnz(J)=%zu/%d  mode=%s  statements=%zu flops=%zu transc=%zu div=%zu\n", d.name, g.jac.Nnz(),
                     g.jac.rows * g.jac.cols, g.jacMode == 1 ? "forward" : g.jacMode == 2 ? "reverse" : "implicit", em.Stats().statements, em.Stats().flops,
                     em.Stats().transcendentals, em.Stats().divisions);
    }
    // Value + Jacobian, phased with an explicit per-lane LDS home for cross-phase values.
    if (ldsSlots > 0) {
        std::vector<std::string> names;
        for (int i = 0; i < d.nx; ++i) names.push_back("io.x(" + std::to_string(i) + ")");
        for (int i = 0; i < d.nu; ++i) names.push_back("io.u(" + std::to_string(i) + ")");
        for (int i = 0; i < d.nw; ++i) names.push_back("io.w(" + std::to_string(i) + ")");
        for (int i = 0; i < d.np; ++i) names.push_back("io.p(" + std::to_string(i) + ")");
        for (int i = 0; i < 64; ++i) names.push_back("aux_unused");
        std::vector<std::vector<tape::OutputSlot>> phases(1);
        for (int i = 0; i < d.Ny(); ++i) phases[0].push_back({g.tape.outputs[static_cast<std::size_t>(i)], "io.f(" + std::to_string(i) + ", %s);"});
        // Column order: columns that share sub-expressions are made neighbours so that the shared
        // values die quickly: base twist, pose, then (q_j, v_j, u_j) joint by joint.
        std::vector<int> colOrder;
        if (d.nx == 37 && d.nu == 12) {
            for (int c = 19; c < 25; ++c) colOrder.push_back(c);
            for (int c = 0; c < 7; ++c) colOrder.push_back(c);
            for (int j = 0; j < 12; ++j) {
                colOrder.push_back(7 + j);
                colOrder.push_back(25 + j);
                colOrder.push_back(37 + j);
            }
        } else {
            for (int c = 0; c < g.jac.cols; ++c) colOrder.push_back(c);
        }
        // Reverse-accumulated programs (one adjoint sweep per output row) are cut by ROW instead: the primal
        // values every sweep reads are the cross-phase set that gets the LDS home.
        if (phaseByRow) {
            colOrder.clear();
            for (int r = 0; r < g.jac.rows; ++r) colOrder.push_back(r);
        }
        for (int c : colOrder) {
            std::vector<tape::OutputSlot> ph;
            for (std::size_t k = 0; k < g.jac.Nnz(); ++k)
                if ((phaseByRow ? g.jac.row[k] : g.jac.col[k]) == c)
                    ph.push_back({g.jac.value[k],
                                  "io.j(" + std::to_string(k) + ", " + std::to_string(g.jac.row[k]) + ", " + std::to_string(g.jac.col[k]) + ", %s);"});
            if (!ph.empty()) phases.push_back(std::move(ph));
        }
        tape::Emitter em{g.tape.graph, names};
        int used = 0;
        const std::string body = em.EmitPhased(phases, ldsSlots, used, rematConsumers, rematDepth, prefetch);
        os << "// phased: " << phases.size() << " phases, " << used << " LDS slots per lane, " << em.Stats().statements << " statements\n"
           << "inline constexpr int kLdsSlots = " << used << ";\n"
           << "template <class IO>\n__device__ __forceinline__ void ValueJacobianPhased(IO& io) {\n"
           << body << "}\n\n";
        std::fprintf(stderr, "[codegen] %-10s phased: %zu phases, %d LDS slots/lane (cap %d), %zu statements\n", d.name, phases.size(), used, ldsSlots,
                     em.Stats().statements);
    } else {
        os << "inline constexpr int kLdsSlots = 0;\ntemplate <class IO>\n__device__ __forceinline__ void ValueJacobianPhased(IO&) {}\n";
    }
    os << "}  // namespace ungar_amd::gen::" << name << "\n";
    std::ofstream f(dir + "/" + name + "_gen.hpp");
    f << os.str();
}

/// Scalar stage-cost node: value, gradient w.r.t. (x, u) and the UPPER triangle of
/// the Hessian w.r.t. (x, u) (what Function::Hessian delivers, function.hpp:236-274) in one body.  Sinks:
/// io.f(0, v); io.j(k, 0, c, v) for the gradient (a 1 x (nx+nu) Jacobian); io.h(k, row, col, v) for the Hessian.
void EmitCostHip(const models::NodeDims& d, const std::function<void(const AD*, const AD*, const AD*, const AD*, AD*)>& fn, const std::string& dir) {
    const int nIn = d.nx + d.nu + d.nw + d.np, ncols = d.nx + d.nu;
    std::vector<AD> in = tape::Independent(nIn);
    AD y{0.0};
    fn(in.data(), in.data() + d.nx, in.data() + d.nx + d.nu, in.data() + d.nx + d.nu + d.nw, &y);
    tape::Tape t = tape::MakeTape(std::vector<AD>{y});
    tape::Differentiator diff{t};
    const tape::SparseEntries grad = diff.Jacobian(ncols, 0);
    const tape::SparseEntries hes = diff.Hessian(0, ncols);
    const std::string name = d.name;
    std::ostringstream os;
    os << "// GENERATED by ungar_amd/csrc/codegen/codegen_main.cpp -- do not edit.\n"
       << "// Scalar stage-cost node '" << name << "': value, gradient and upper-triangular Hessian w.r.t. (x, u).\n"
       << "#pragma once\n#include <hip/hip_runtime.h>\n\n"
       << "namespace ungar_amd::gen::" << name << " {\n\n"
       << "inline constexpr int kNx = " << d.nx << ", kNu = " << d.nu << ", kNw = " << d.nw << ", kNp = " << d.np << ";\n"
       << "inline constexpr int kJacRows = 1, kJacCols = " << ncols << ", kJacNnz = " << grad.Nnz() << ", kHesNnz = " << hes.Nnz() << ";\n";
    auto table = [&](const char* nm, const std::vector<int>& v, const char* size) {
        os << "inline constexpr int " << nm << "[" << size << "] = {";
        for (std::size_t k = 0; k < v.size(); ++k) os << (k ? "," : "") << v[k];
        os << "};\n";
    };
    table("kJacRow", grad.row, "kJacNnz");
    table("kJacCol", grad.col, "kJacNnz");
    table("kHesRow", hes.row, "kHesNnz");
    table("kHesCol", hes.col, "kHesNnz");
    std::vector<tape::OutputSlot> slots{{t.outputs[0], "io.f(0, %s);"}};
    std::vector<tape::Id> roots{t.outputs[0]};
    for (std::size_t k = 0; k < grad.Nnz(); ++k) {
        slots.push_back({grad.value[k], "io.j(" + std::to_string(k) + ", 0, " + std::to_string(grad.col[k]) + ", %s);"});
        roots.push_back(grad.value[k]);
    }
    for (std::size_t k = 0; k < hes.Nnz(); ++k) {
        slots.push_back({hes.value[k], "io.h(" + std::to_string(k) + ", " + std::to_string(hes.row[k]) + ", " + std::to_string(hes.col[k]) + ", %s);"});
        roots.push_back(hes.value[k]);
    }
    tape::Emitter em{t.graph, InputNames(d, false)};
    const std::string body = em.Emit(slots);
    os << "// " << em.Stats().statements << " statements, " << em.Stats().flops << " flops\n"
       << "template <class IO>\n__device__ __forceinline__ void ValueGradientHessian(IO& io) {\n"
       << HipPrologue(d, UsedInputs(t.graph, roots)) << body << "}\n\n"
       << "}  // namespace ungar_amd::gen::" << name << "\n";
    std::ofstream f(dir + "/" + name + "_gen.hpp");
    f << os.str();
    std::fprintf(stderr, "[codegen] %-14s scalar cost: grad nnz %zu/%d, upper Hessian nnz %zu, statements=%zu\n", d.name, grad.Nnz(), ncols, hes.Nnz(),
                 em.Stats().statements);
}

void EmitC(const Generated& g, const std::string& dir) {
    const auto& d = g.dims;
    const std::string name = d.name;
    std::ostringstream os;
    os << "/* GENERATED -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/README.md).\n"
       << " * Stand-in for the C that CppADCodeGen emits for the reference (function.hpp:468-503):\n"
       << " * one problem instance per call, straight-line code, sparse Jacobian in CSR value order. */\n"
       << "#include <math.h>\n\n";
    os << "const int " << name << "_dims[4] = {" << d.nx << ", " << d.nu << ", " << d.nw << ", " << d.np << "};\n"
       << "const int " << name << "_jac_nnz = " << g.jac.Nnz() << ";\n"
       << "const int " << name << "_jac_row[] = {";
    for (std::size_t k = 0; k < g.jac.Nnz(); ++k) os << (k ? "," : "") << g.jac.row[k];
    os << (g.jac.Nnz() ? "" : "0") << "};\nconst int " << name << "_jac_col[] = {";
    for (std::size_t k = 0; k < g.jac.Nnz(); ++k) os << (k ? "," : "") << g.jac.col[k];
    os << (g.jac.Nnz() ? "" : "0") << "};\nconst int " << name << "_ny = " << d.Ny() << ";\n\n";
    {
        std::vector<tape::OutputSlot> slots;
        for (int i = 0; i < d.Ny(); ++i) slots.push_back({g.tape.outputs[static_cast<std::size_t>(i)], "f[" + std::to_string(i) + "] = %s;"});
        tape::Emitter em{g.tape.graph, InputNames(d, true)};
        os << "void " << name << "_forward_zero(const double* x, const double* u, const double* w, const double* p, double* f) {\n"
           << "    (void)x; (void)u; (void)w; (void)p;\n"
           << em.Emit(slots) << "}\n\n";
    }
    {
        std::vector<tape::OutputSlot> slots;
        for (int i = 0; i < d.Ny(); ++i) slots.push_back({g.tape.outputs[static_cast<std::size_t>(i)], "f[" + std::to_string(i) + "] = %s;"});
        for (std::size_t k = 0; k < g.jac.Nnz(); ++k) slots.push_back({g.jac.value[k], "jac[" + std::to_string(k) + "] = %s;"});
        tape::Emitter em{g.tape.graph, InputNames(d, true)};
        os << "void " << name
           << "_sparse_jacobian(const double* x, const double* u, const double* w, const double* p, double* f, double* jac) {\n"
           << "    (void)x; (void)u; (void)w; (void)p;\n"
           << em.Emit(slots) << "}\n";
    }
    std::ofstream f(dir + "/" + name + "_cg.c");
    f << os.str();
}

}  // namespace

/// The lane-per-leg program split over the two wavefronts of a workgroup (quad_leg_program.hpp: QuadRole): two function templates in
/// gen/anymal_split_gen.hpp, ProducerQuad (right-hand sides of the columns that need the RNEA tangents) and ConsumerQuad (everything else).
static void EmitSplitQuad(const rbd::Model& anymal, const tape::SparseEntries& pattern, const std::string& outDir, int producerSlots, int producerUniform,
                          int consumerSlots, int consumerUniform, int rematConsumers, int rematDepth, int prefetch, bool pairStores) {
    std::ostringstream so;
    so << "// GENERATED by ungar_amd/csrc/codegen (quad_leg_program.hpp, split program) -- do not edit.\n"
       << "#pragma once\n#ifndef __host__\n#define __host__\n#endif\n#ifndef __device__\n#define __device__\n#endif\n\n"
       << "namespace ungar_amd::gen::anymal_split {\n\n"
       << "inline constexpr int kMessages = " << codegen::kQuadMessages << ", kMessageItems = " << codegen::kQuadMessageItems << ";\n";
    for (const codegen::QuadRole role : {codegen::QuadRole::Producer, codegen::QuadRole::Consumer}) {
        const bool producer = role == codegen::QuadRole::Producer;
        const codegen::QuadProgram qp = codegen::RecordQuadLegProgram(anymal, pattern, 1, false, role, pairStores);
        tape::EmitStats qs;
        int lds = 0, uniformUsed = 0;
        const std::string fn = codegen::EmitQuadProgram(qp, producer ? "ProducerQuad" : "ConsumerQuad", &qs, true, producer ? producerSlots : consumerSlots, &lds, rematConsumers,
                                                           rematDepth, prefetch, producer ? producerUniform : consumerUniform, &uniformUsed, false);
        so << "// " << (producer ? "producer" : "consumer") << ": " << qs.statements << " statements, " << qs.flops << " flops per lane\n"
           << "inline constexpr int k" << (producer ? "Producer" : "Consumer") << "LdsSlots = " << lds << ", k" << (producer ? "Producer" : "Consumer")
           << "LdsUniformSlots = " << uniformUsed << ";\n"
           << fn << "\n";
        std::fprintf(stderr, "[codegen] anymal_split %s: %zu statements, %zu flops per lane, LDS home %d + %d slots\n", producer ? "producer" : "consumer", qs.statements,
                     qs.flops, lds, uniformUsed);
    }
    so << "}  // namespace ungar_amd::gen::anymal_split\n";
    std::ofstream sf(outDir + "/anymal_split_gen.hpp");
    sf << so.str();
}

int main(int argc, char** argv) {
    std::string outDir, cDir, robot;
    int jacMode = 0, structuredDMode = 1, ldsSlots = 320;
    int rematConsumers = 2, rematDepth = 3, prefetch = 48;
    int quadColumnsPerPhase = 1, quadRematConsumers = 4, quadRematDepth = 4;  // tools/sweep_quad.sh on MI355X
    bool quadPairStores = false;      // fused program: sinks carry two entries of a column -> one 16-byte store in kernels with paired stores (quad_kernel.hpp: PAIR; tools/quad_split_bench.hip)
    bool quadMergeShared = false;     // four base-row entries of a shared column per store instruction (measured: no gain, 0.347 vs 0.343 ms)
    int quadPrefetch = 48;            // LDS loads are hoisted this many statements ahead of their first use ...
    bool quadPrefetchAcross = false;  // ... and may cross into the tail of the previous phase
    int quadUniformSlots = 80;  // compact (one copy per quad) LDS slots; budget: quadLdsSlots + quadUniformSlots / 4 <= 80
    int crbaQuadLdsSlots = 40;  // LDS home of the lane-per-leg inertia-matrix program
    bool rneaQuadReverse = true;  // partials of the lane-local function by reverse accumulation, one phase per row (measured: 11 % faster than forward / per column)
    int rneaQuadLdsSlots = 60, rneaQuadUniformSlots = 80;  // LDS home of the lane-per-leg joint-torque program (same budget)
    // --quad-explicit-fma 1: multiply-adds of the lane-per-leg programs contracted by the generator (Op::Fma, tape::FuseMultiplyAdd) and compiled with contraction off, so
    // that every kernel a program is instantiated into rounds identically whatever its store code looks like to the compiler's heuristics (with the compiler's own
    // contraction the instantiations differ in the last bit of ~1e-4 of the entries: `x y + z w` may become fma(x, y, z w) or fma(z, w, x y)).  OFF by default: the
    // compiler finds ~100 contractions more than the generator's rule and the kernel is 2.5 % faster with them (0.2259-0.2281 against 0.2326 ms,
    // profiles/r06c_explicit_fma.log); the tests hold kernel variants to 1e-14 of the block scale instead of to the bit.
    bool quadExplicitFma = false;
    // unit-fastest program: phases in this order (its paired sinks never cross a phase: any order of the column phases is valid).  As for the tile program below, the
    // order decides how evenly the store traffic of the lock-stepped wavefronts is spread: the literal position columns (21-23: stores without arithmetic) first, then the
    // tile program's order.  0.2415-0.2428 ms per 81 920 nodes against 0.2531 as recorded (profiles/r06b_quad_phase_order.log).
    std::vector<int> quadPhaseOrder{0, 21, 22, 23, 1, 8, 16, 9, 15, 10, 14, 5, 13, 6, 12, 7, 11, 2, 17, 3, 18, 4, 19, 20};
    // tile program: its (self-contained) phases in this order (empty = as recorded: 0 kinematics / CRBA / factorisation, 1 value, 2-4 q_L, 5-7 v_L, 8-10 u_L columns of the
    // four legs, 11-16 base twist, 17-20 quaternion columns).  With one wavefront per SIMD every resident wavefront is in the same phase at the same time, so the ORDER
    // decides how evenly the chip's store traffic is spread: the densest phases (u_L: 19 KiB of results per 2.9 k cycles of arithmetic) right after the store-free
    // prefix, each followed by one of the arithmetic-heavy twist columns, the q_L columns (heaviest arithmetic) last beside the quaternion columns.  Measured over
    // fifteen orders (tools/quad_tile_bench.hip, profiles/r06a_tile_variants5-7.log): 0.2237 ms per 81 920 nodes against 0.238-0.244 as recorded.
    std::vector<int> tilePhaseOrder{0, 1, 8, 16, 9, 15, 10, 14, 5, 13, 6, 12, 7, 11, 2, 17, 3, 18, 4, 19, 20};
    bool tileInterleave = true;  // tile program: each store statement right behind the statements that produce its values (stores spread over the phase)
    int tileLdsSlots = -1, tileUniformSlots = -1;  // tile program: LDS home of its own (-1: as the quad program)
    int quadLdsSlots = 60;  // 64-lane workgroups, four per CU: 160 KiB / 4 / 64 lanes / 8 B
    // split program (producer / consumer wavefront of a 128-lane workgroup, two wavefronts per SIMD): LDS homes of the two halves
    int splitProducerSlots = 24, splitProducerUniform = 12, splitConsumerSlots = 4, splitConsumerUniform = 80;
    int rbdLdsSlots = -1;  // >= 0 overrides the per-lane LDS home of the phased rigid-body quantity Jacobians (0 = plain bodies only)
    std::vector<std::string> only;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--out" && i + 1 < argc) outDir = argv[++i];
        else if (a == "--c-oracle" && i + 1 < argc) cDir = argv[++i];
        else if (a == "--anymal-robot" && i + 1 < argc) robot = argv[++i];
        else if (a == "--jac-mode" && i + 1 < argc) jacMode = std::atoi(argv[++i]);
        else if (a == "--structured-dmode" && i + 1 < argc) structuredDMode = std::atoi(argv[++i]);
        else if (a == "--lds-slots" && i + 1 < argc) ldsSlots = std::atoi(argv[++i]);
        else if (a == "--remat-consumers" && i + 1 < argc) rematConsumers = std::atoi(argv[++i]);
        else if (a == "--remat-depth" && i + 1 < argc) rematDepth = std::atoi(argv[++i]);
        else if (a == "--prefetch" && i + 1 < argc) prefetch = std::atoi(argv[++i]);
        else if (a == "--quad-lds-slots" && i + 1 < argc) quadLdsSlots = std::atoi(argv[++i]);
        else if (a == "--rnea-quad-reverse" && i + 1 < argc) rneaQuadReverse = std::atoi(argv[++i]) != 0;
        else if (a == "--rnea-quad-slots" && i + 2 < argc) {
            rneaQuadLdsSlots = std::atoi(argv[++i]);
            rneaQuadUniformSlots = std::atoi(argv[++i]);
        }
        else if (a == "--split-slots" && i + 4 < argc) {
            splitProducerSlots = std::atoi(argv[++i]);
            splitProducerUniform = std::atoi(argv[++i]);
            splitConsumerSlots = std::atoi(argv[++i]);
            splitConsumerUniform = std::atoi(argv[++i]);
        }
        else if (a == "--quad-pair-stores" && i + 1 < argc) quadPairStores = std::atoi(argv[++i]) != 0;
        else if (a == "--quad-explicit-fma" && i + 1 < argc) quadExplicitFma = std::atoi(argv[++i]) != 0;
        else if (a == "--tile-interleave" && i + 1 < argc) tileInterleave = std::atoi(argv[++i]) != 0;
        else if ((a == "--tile-phase-order" || a == "--quad-phase-order") && i + 1 < argc) {
            std::vector<int>& order = a == "--tile-phase-order" ? tilePhaseOrder : quadPhaseOrder;
            order.clear();
            for (const char* c = argv[++i]; *c;) {
                order.push_back(std::atoi(c));
                while (*c && *c != ',') ++c;
                if (*c == ',') ++c;
            }
        }
        else if (a == "--tile-lds-slots" && i + 2 < argc) {
            tileLdsSlots = std::atoi(argv[++i]);
            tileUniformSlots = std::atoi(argv[++i]);
        }
        else if (a == "--quad-merge-shared" && i + 1 < argc) quadMergeShared = std::atoi(argv[++i]) != 0;
        else if (a == "--quad-uniform-slots" && i + 1 < argc) quadUniformSlots = std::atoi(argv[++i]);
        else if (a == "--quad-prefetch" && i + 2 < argc) {
            quadPrefetch = std::atoi(argv[++i]);
            quadPrefetchAcross = std::atoi(argv[++i]) != 0;
        }
        else if (a == "--quad-columns-per-phase" && i + 1 < argc) quadColumnsPerPhase = std::atoi(argv[++i]);
        else if (a == "--quad-remat" && i + 2 < argc) {
            quadRematConsumers = std::atoi(argv[++i]);
            quadRematDepth = std::atoi(argv[++i]);
        }
        else if (a == "--rbd-lds-slots" && i + 1 < argc) rbdLdsSlots = std::atoi(argv[++i]);
        else if (a == "--model" && i + 1 < argc) only.push_back(argv[++i]);
        else {
            std::fprintf(stderr, "usage: %s --out DIR [--c-oracle DIR] [--anymal-robot FILE] [--jac-mode 0|1|2] [--model NAME]...\n", argv[0]);
            return 2;
        }
    }
    if (outDir.empty()) {
        std::fprintf(stderr, "--out is required\n");
        return 2;
    }
    std::vector<NodeSpec> specs;
    specs.push_back({models::kQuadrotorDims, [](auto... a) { models::QuadrotorNode<AD>(a...); }});
    specs.push_back({models::kRcCarDims, [](auto... a) { models::RcCarNode<AD>(a...); }});
    specs.push_back({models::kSrbdDims, [](auto... a) { models::SrbdNode<AD>(a...); }});
    specs.push_back({models::kSrbdIneqDims, [](auto... a) { models::SrbdIneqNode<AD>(a...); }});
    specs.push_back({models::kQuadrotorIneqDims, [](auto... a) { models::QuadrotorIneqNode<AD>(a...); }});
    specs.push_back({models::kRcCarIneqDims, [](auto... a) { models::RcCarIneqNode<AD>(a...); }});
    specs.push_back({models::kSrbdFeetDims, [](auto... a) { models::SrbdFeetNode<AD>(a...); }});
    rbd::Model anymal;
    if (!robot.empty()) {
        anymal = rbd::BuildModel(rbd::ReadRobotDescription(robot));
        if (anymal.nq != 19 || anymal.nv != 18) {
            std::fprintf(stderr, "ANYmal model has nq=%d nv=%d, expected 19/18 (test/rbd/robot.test.cpp:103-106)\n", anymal.nq, anymal.nv);
            return 1;
        }
        std::fprintf(stderr, "[codegen] anymal: %d joints, nq=%d nv=%d, mass=%.6f kg\n", anymal.NumJoints() - 1, anymal.nq, anymal.nv,
                     anymal.TotalMass());
        specs.push_back({models::kAnymalDims, [&anymal](const AD* x, const AD* u, const AD* w, const AD* p, AD* xn) {
                             models::FloatingBaseNode<AD>(anymal, x, u, w, p, xn);
                         }});
        // rigid-body quantities as node models
        specs.push_back({models::kAnymalRneaDims, [&anymal](const AD* x, const AD* u, const AD*, const AD*, AD* y) { models::JointTorquesNode<AD>(anymal, x, u, y); }, false, 160, 1});
        specs.push_back({models::kAnymalCrbaDims, [&anymal](const AD* x, const AD*, const AD*, const AD*, AD* y) { models::InertiaMatrixNode<AD>(anymal, x, y); }, false, 320, 1});
        specs.push_back({models::kAnymalMinvDims, [&anymal](const AD* x, const AD*, const AD*, const AD*, AD* y) { models::InertiaInverseNode<AD>(anymal, x, y); }, true});
        specs.push_back({models::kAnymalFeetDims, [&anymal](const AD* x, const AD*, const AD*, const AD*, AD* y) { models::FootFramesNode<AD>(anymal, x, y); }});
        specs.push_back({models::kAnymalCentroidalDims, [&anymal](const AD* x, const AD*, const AD*, const AD*, AD* y) { models::CentroidalMomentumNode<AD>(anymal, x, y); }, false, 160, 2});
    }
    auto wanted = [&](const char* nm) {
        if (only.empty()) return true;
        for (const auto& o : only)
            if (o == nm) return true;
        return false;
    };
    for (const NodeSpec& s : specs) {
        if (std::string(s.dims.name) == "anymal") {
            // Three kernels for the same node function and the same sparsity pattern:
            //   "anymal"     structured implicit differentiation, phased body with an LDS home (production)
            //   "anymal_reg" the same structured program as one plain straight-line body
            //   "anymal_ad"  derivatives by taping ABA (what the reference does, robot.test.cpp:124-135)
            if (!wanted("anymal") && !wanted("anymal_ad") && !wanted("anymal_reg")) continue;
            Generated adv = Record(s, jacMode);
            Generated st = RecordFloatingBaseStructured(anymal, adv, "anymal", structuredDMode);
            adv.dims.name = "anymal_ad";
            if (wanted("anymal_ad")) {
                EmitHip(adv, outDir, false, 0);
                if (!cDir.empty()) EmitC(adv, cDir);
            }
            if (wanted("anymal")) {
                EmitHip(st, outDir, true, ldsSlots, rematConsumers, rematDepth, prefetch);
                if (!cDir.empty()) EmitC(st, cDir);
            }
            if (wanted("anymal_reg")) {
                st.dims.name = "anymal_reg";
                EmitHip(st, outDir, true, 0);
            }
            if (wanted("anymal")) {  // lane-per-leg SPMD program (dense Jacobian path of the 'anymal' model)
                const codegen::QuadProgram qp = codegen::RecordQuadLegProgram(anymal, adv.jac, quadColumnsPerPhase, quadMergeShared, codegen::QuadRole::Fused, quadPairStores && !quadMergeShared);
                tape::EmitStats qs;
                int quadLds = 0, quadUniformUsed = 0;
                const std::string fn = codegen::EmitQuadProgram(qp, "ValueJacobianQuad", &qs, true, quadLdsSlots, &quadLds, quadRematConsumers, quadRematDepth, quadPrefetch,
                                                                   quadUniformSlots, &quadUniformUsed, quadPrefetchAcross, false, quadPhaseOrder.empty() || quadColumnsPerPhase != 1 ? nullptr : &quadPhaseOrder, quadExplicitFma);
                std::ostringstream qo;
                qo << "// GENERATED by ungar_amd/csrc/codegen (quad_leg_program.hpp) -- do not edit.\n"
                   << "// ANYmal B shooting node, one lane per leg: " << qs.statements << " statements, " << qs.flops << " flops, "
                   << qs.transcendentals << " transcendentals, " << qs.divisions << " divisions per lane.\n"
                   << "#pragma once\n#ifndef __host__\n#define __host__\n#endif\n#ifndef __device__\n#define __device__\n#endif\n\n"
                   << "namespace ungar_amd::gen::anymal_quad {\n\n"
                   << "inline constexpr int kNumConstants = " << qp.constants.size() << ";\n"
                   << "inline constexpr int kLdsSlots = " << quadLds << ";  // per-lane LDS slots of the phased body\n"
                   << "inline constexpr int kLdsUniformSlots = " << quadUniformUsed << ";  // per-quad (lane-uniform) LDS slots\n"
                   << "inline constexpr int kJacNnz = " << adv.jac.Nnz() << ";  // entries of the sparse (CSR) output\n"
                   << "// per-leg CSR index patterns (k_L - k_0) of the per-lane Jacobian sinks: one per-lane base pointer each\n"
                   << "struct SparsePlan {\n    static constexpr int kCount = " << qp.sparseDeltas.size() << ";\n    static constexpr int kDeltas["
                   << std::max<std::size_t>(1, qp.sparseDeltas.size()) << "][4] = {";
                for (const auto& dl : qp.sparseDeltas) qo << "{" << dl[0] << ", " << dl[1] << ", " << dl[2] << ", " << dl[3] << "}, ";
                if (qp.sparseDeltas.empty()) qo << "{0, 0, 0, 0}";
                qo << "};\n};\n"
                   << "// leg constants that differ between legs, [k][leg]; legs in model order (LF, LH, RF, RH)\n"
                   << "inline constexpr double kLegConstants[" << std::max<std::size_t>(1, qp.constants.size()) << "][4] = {\n";
                for (const auto& c : qp.constants) {
                    char buf[256];
                    std::snprintf(buf, sizeof buf, "    {%.17g, %.17g, %.17g, %.17g},\n", c[0], c[1], c[2], c[3]);
                    qo << buf;
                }
                qo << "};\n#ifdef __HIPCC__\n// device copy of the table (indexed by the lane's leg at run time; one per translation unit)\nstatic __device__ __constant__ double kLegConstantsDev["
                   << std::max<std::size_t>(1, qp.constants.size()) << "][4] = {\n";
                for (const auto& c : qp.constants) {
                    char buf[256];
                    std::snprintf(buf, sizeof buf, "    {%.17g, %.17g, %.17g, %.17g},\n", c[0], c[1], c[2], c[3]);
                    qo << buf;
                }
                // value only: the same recorded program with the value sinks alone (everything the Jacobian columns need is unreachable
                // from them and is not emitted): what forward_zero and the SQP's stacked line search launch for this model
                codegen::QuadProgram qv = qp;
                qv.slots.clear();
                qv.phaseStarts.clear();
                for (const auto& sl : qp.slots)
                    if (sl.sink.rfind("io.f_base(", 0) == 0 || sl.sink.rfind("io.f_leg(", 0) == 0) qv.slots.push_back(sl);
                tape::EmitStats vs;
                const std::string fnValue = codegen::EmitQuadProgram(qv, "ValueQuad", &vs, false, 0, nullptr, 0, 0, 0, 0, nullptr, false, false, nullptr, quadExplicitFma);
                qo << "};\n#endif\n\n" << fn << "\n// value only: " << vs.statements << " statements, " << vs.flops << " flops per lane\n" << fnValue
                   << "\n}  // namespace ungar_amd::gen::anymal_quad\n";
                std::fprintf(stderr, "[codegen] anymal_quad value only: %zu statements, %zu flops per lane\n", vs.statements, vs.flops);
                std::ofstream qf(outDir + "/anymal_quad_gen.hpp");
                qf << qo.str();
                {  // tile program: the same node program with the Jacobian leaving as register images of the wavefront (quad_leg_program.hpp: tileStores)
                    const codegen::QuadProgram tp = codegen::RecordQuadLegProgram(anymal, adv.jac, quadColumnsPerPhase, false, codegen::QuadRole::Fused, false, true);
                    tape::EmitStats ts;
                    int tileLds = 0, tileUniformUsed = 0;
                    const std::string tfn = codegen::EmitQuadProgram(tp, "ValueJacobianQuadTiles", &ts, true, tileLdsSlots >= 0 ? tileLdsSlots : quadLdsSlots, &tileLds, quadRematConsumers,
                                                                        quadRematDepth, quadPrefetch, tileUniformSlots >= 0 ? tileUniformSlots : quadUniformSlots, &tileUniformUsed,
                                                                        quadPrefetchAcross, tileInterleave, tilePhaseOrder.empty() ? nullptr : &tilePhaseOrder, quadExplicitFma);
                    std::ostringstream to;
                    to << "// GENERATED by ungar_amd/csrc/codegen (quad_leg_program.hpp, tile stores) -- do not edit.\n"
                       << "// ANYmal B shooting node, one lane per leg, Jacobian stored as register images: " << ts.statements << " statements, " << ts.flops
                       << " flops per lane.\n"
                       << "#pragma once\n#ifndef __host__\n#define __host__\n#endif\n#ifndef __device__\n#define __device__\n#endif\n\n"
                       << "namespace ungar_amd::gen::anymal_tiles {\n\n"
                       << "inline constexpr int kLdsSlots = " << tileLds << ";\n"
                       << "inline constexpr int kLdsUniformSlots = " << tileUniformUsed << ";\n"
                       << "inline constexpr int kImages = " << tp.tileEntries.size() / 4 << ";  // store images per wavefront (two per 16-byte store instruction)\n"
                       << "// entry (row * 49 + col; -1: padding) held by the lane of leg q in image s: kEntryOfSlot[4 * s + q]\n"
                       << "inline constexpr short kEntryOfSlot[" << tp.tileEntries.size() << "] = {";
                    for (std::size_t i = 0; i < tp.tileEntries.size(); ++i) to << (i % 16 ? " " : "\n    ") << tp.tileEntries[i] << ",";
                    to << "\n};\n#ifdef __HIPCC__\n// device copy of the table (one per translation unit)\nstatic __device__ __constant__ short kEntryOfSlotDev[" << tp.tileEntries.size() << "] = {";
                    for (std::size_t i = 0; i < tp.tileEntries.size(); ++i) to << (i % 16 ? " " : "\n    ") << tp.tileEntries[i] << ",";
                    to << "\n};\n#endif\n\n" << tfn << "\n}  // namespace ungar_amd::gen::anymal_tiles\n";
                    std::ofstream tf(outDir + "/anymal_tiles_gen.hpp");
                    tf << to.str();
                    std::fprintf(stderr, "[codegen] anymal_tiles (lane per leg, tile stores): %zu statements, %zu flops per lane, %zu images\n", ts.statements, ts.flops,
                                 tp.tileEntries.size() / 4);
                }
                EmitSplitQuad(anymal, adv.jac, outDir, splitProducerSlots, splitProducerUniform, splitConsumerSlots, splitConsumerUniform, quadRematConsumers, quadRematDepth,
                              quadPrefetch, true);
                std::fprintf(stderr, "[codegen] anymal_quad (lane per leg): %zu statements, %zu flops per lane, %zu table constants\n", qs.statements, qs.flops,
                             qp.constants.size());
            }
            continue;
        }
        if (!only.empty()) {
            bool keep = false;
            for (const auto& o : only) keep = keep || o == s.dims.name;
            if (!keep) continue;
        }
        const Generated g = Record(s, jacMode ? jacMode : s.jacMode);
        const int slots = rbdLdsSlots >= 0 && s.phasedLdsSlots > 0 ? rbdLdsSlots : s.phasedLdsSlots;
        EmitHip(g, outDir, false, slots, slots > 0 ? 4 : rematConsumers, slots > 0 ? 4 : rematDepth, prefetch, g.jacMode == 2);
        if (!cDir.empty()) EmitC(g, cDir);
        const std::string sname = s.dims.name;
        if (sname == "anymal_rnea" || sname == "anymal_crba" || sname == "anymal_centroidal") {  // lane-per-leg SPMD programs: what the Jacobian modes of these models launch
            const bool rnea = sname == "anymal_rnea", crba = sname == "anymal_crba";
            const codegen::QuadProgram qp = rnea   ? codegen::RecordQuadRneaProgram(anymal, g.jac, rneaQuadReverse)
                                            : crba ? codegen::RecordQuadCrbaProgram(anymal, g.jac)
                                                   : codegen::RecordQuadCentroidalProgram(anymal, g.jac);
            tape::EmitStats qs;
            int quadLds = 0, quadUniformUsed = 0;
            const std::string fn = crba ? codegen::EmitQuadProgram(qp, "ValueJacobianQuad", &qs, true, crbaQuadLdsSlots, &quadLds, quadRematConsumers, quadRematDepth, quadPrefetch, 0, &quadUniformUsed, false)
                                        : codegen::EmitQuadProgram(qp, "ValueJacobianQuad", &qs, true, rneaQuadLdsSlots, &quadLds, quadRematConsumers, quadRematDepth, quadPrefetch, rneaQuadUniformSlots,
                                                                   &quadUniformUsed, false);
            const std::string ns = sname + "_quad";
            std::ostringstream qo;
            qo << "// GENERATED by ungar_amd/csrc/codegen (" << (rnea ? "quad_rnea_program.hpp" : crba ? "quad_crba_program.hpp" : "quad_centroidal_program.hpp") << ") -- do not edit.\n"
               << "// ANYmal B " << (rnea ? "joint torques" : crba ? "joint-space inertia matrix" : "centroidal momentum") << " and derivatives, one lane per leg: " << qs.statements << " statements, " << qs.flops << " flops, "
               << qs.transcendentals << " transcendentals, " << qs.divisions << " divisions per lane.\n"
               << "#pragma once\n#ifndef __host__\n#define __host__\n#endif\n#ifndef __device__\n#define __device__\n#endif\n\n"
               << "namespace ungar_amd::gen::" << ns << " {\n\n"
               << "inline constexpr int kNumConstants = " << qp.constants.size() << ";\n"
               << "inline constexpr int kLdsSlots = " << quadLds << ", kLdsUniformSlots = " << quadUniformUsed << ";\n"
               << "inline constexpr int kJacNnz = " << g.jac.Nnz() << ";  // entries of the sparse (CSR) output\n"
               << "// per-leg CSR index patterns (k_L - k_0) of the per-lane Jacobian sinks: one per-lane base pointer each\n"
               << "struct SparsePlan {\n    static constexpr int kCount = " << qp.sparseDeltas.size() << ";\n    static constexpr int kDeltas[" << std::max<std::size_t>(1, qp.sparseDeltas.size())
               << "][4] = {";
            for (const auto& dl : qp.sparseDeltas) qo << "{" << dl[0] << ", " << dl[1] << ", " << dl[2] << ", " << dl[3] << "}, ";
            if (qp.sparseDeltas.empty()) qo << "{0, 0, 0, 0}";
            qo << "};\n};\n"
               << "// leg constants that differ between legs, [k][leg]; legs in model order (LF, LH, RF, RH)\n";
            for (int dev = 0; dev < 2; ++dev) {
                qo << (dev ? "#ifdef __HIPCC__\nstatic __device__ __constant__ double kLegConstantsDev[" : "inline constexpr double kLegConstants[") << std::max<std::size_t>(1, qp.constants.size())
                   << "][4] = {\n";
                for (const auto& c : qp.constants) {
                    char buf[256];
                    std::snprintf(buf, sizeof buf, "    {%.17g, %.17g, %.17g, %.17g},\n", c[0], c[1], c[2], c[3]);
                    qo << buf;
                }
                qo << (dev ? "};\n#endif\n\n" : "};\n");
            }
            qo << fn << "\n}  // namespace ungar_amd::gen::" << ns << "\n";
            std::ofstream qf(outDir + "/" + ns + "_gen.hpp");
            qf << qo.str();
            std::fprintf(stderr, "[codegen] %s (lane per leg): %zu statements, %zu flops per lane, %zu table constants, %zu index patterns\n", ns.c_str(), qs.statements, qs.flops, qp.constants.size(),
                         qp.sparseDeltas.size());
        }
    }
    // scalar stage-cost nodes (value + gradient + upper Hessian)
    {
        bool keep = only.empty();
        for (const auto& o : only) keep = keep || o == models::kQuadrotorCostDims.name;
        if (keep) EmitCostHip(models::kQuadrotorCostDims, models::QuadrotorCostNode<AD>, outDir);
        keep = only.empty();
        for (const auto& o : only) keep = keep || o == models::kSrbdCostDims.name;
        if (keep) EmitCostHip(models::kSrbdCostDims, models::SrbdCostNode<AD>, outDir);
        keep = only.empty();
        for (const auto& o : only) keep = keep || o == models::kRcCarCostDims.name;
        if (keep) EmitCostHip(models::kRcCarCostDims, models::RcCarCostNode<AD>, outDir);
        keep = only.empty();
        for (const auto& o : only) keep = keep || o == models::kAnymalCostDims.name;
        if (keep) EmitCostHip(models::kAnymalCostDims, models::AnymalCostNode<AD>, outDir);
    }
    return 0;
}
