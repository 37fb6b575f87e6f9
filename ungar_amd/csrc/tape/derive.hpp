// ungar_amd :: derivative transforms on the expression tape (the CppADCodeGen "sparse Jacobian /
// sparse Hessian model" half of the replacement).
//
// Reference behaviour reproduced:
//   * sparse Jacobian restricted to the first n (decision-variable) columns, parameters trimmed
//       include/ungar/autodiff/function.hpp:529-550
//   * sparse *upper-triangular* Hessian of dependent variable 0 over decision variables only
//       include/ungar/autodiff/function.hpp:552-574, 232-235
//   * sparsity by dependency propagation through the operation sequence (CppAD's notion), CSR
//     with rows ascending; columns ascending inside a row (canonical -- the reference's own
//     within-row order is generator-defined and "cannot be used reliably", function.hpp:367-374)
//
// Both transforms are source-to-source: derivative expressions are appended to the same
// hash-consed DAG, so partials shared between directions (cos(q), 1/m, R(q) ...) are emitted once.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <unordered_set>
#include <utility>
#include <vector>

#include "graph.hpp"
#include "scalar.hpp"

namespace ungar_amd::tape {

/// A recorded function y = f([x; p]) living in its own compact graph.
struct Tape {
    Graph graph;
    std::vector<Id> inputs;   // size n + p, inputs[i] is the node of independent variable i
    std::vector<Id> outputs;  // size m
};

/// Start recording: clears the thread's graph and returns n fresh independent variables
/// (CppAD::Independent, function.hpp:456-458).
inline std::vector<AD> Independent(int n) {
    Graph& g = CurrentGraph();
    g.Clear();
    std::vector<AD> x;
    x.reserve(static_cast<std::size_t>(n));
    for (int i = 0; i < n; ++i) x.push_back(AD::FromId(g.Input()));
    return x;
}

namespace detail {

/// Copies the sub-DAG reachable from `roots` (plus every input) of `src` into `dst`, preserving
/// relative order.  Returns old-id -> new-id.
inline std::vector<Id> CopyReachable(const Graph& src, const std::vector<Id>& roots, Graph& dst) {
    const std::size_t n = src.Size();
    std::vector<std::uint8_t> live(n, 0);
    for (Id r : roots) live[static_cast<std::size_t>(r)] = 1;
    for (std::size_t i = n; i-- > 0;) {
        if (!live[i]) continue;
        const Node& nd = src.At(static_cast<Id>(i));
        if (nd.op == Op::Input || nd.op == Op::Const) continue;
        for (Id o : {nd.a, nd.b, nd.c, nd.d})
            if (o != kNoId) live[static_cast<std::size_t>(o)] = 1;
    }
    std::vector<Id> map(n, kNoId);
    for (std::size_t i = 0; i < n; ++i) {
        const Node& nd = src.At(static_cast<Id>(i));
        if (nd.op == Op::Input) {
            map[i] = dst.Input();
            continue;
        }
        if (!live[i]) continue;
        switch (Arity(nd.op)) {
            case 0: map[i] = dst.Constant(nd.value); break;
            case 1: map[i] = dst.Unary(nd.op, map[static_cast<std::size_t>(nd.a)]); break;
            case 2:
                map[i] = dst.Binary(nd.op, map[static_cast<std::size_t>(nd.a)], map[static_cast<std::size_t>(nd.b)]);
                break;
            default:
                map[i] = dst.Cond(nd.op,
                                  map[static_cast<std::size_t>(nd.a)],
                                  map[static_cast<std::size_t>(nd.b)],
                                  map[static_cast<std::size_t>(nd.c)],
                                  map[static_cast<std::size_t>(nd.d)]);
        }
    }
    return map;
}

}  // namespace detail

/// Stop recording (CppAD::ADFun(xp, y) + optimize, function.hpp:465-466): dead code is dropped,
/// everything else was optimised while it was being recorded.
inline Tape MakeTape(const std::vector<AD>& y) {
    Graph& g = CurrentGraph();
    std::vector<Id> roots;
    roots.reserve(y.size());
    for (const AD& v : y) roots.push_back(v.Node());
    Tape t;
    const std::vector<Id> map = detail::CopyReachable(g, roots, t.graph);
    for (std::size_t i = 0; i < g.Size(); ++i)
        if (g.At(static_cast<Id>(i)).op == Op::Input) t.inputs.push_back(map[i]);
    for (Id r : roots) t.outputs.push_back(map[static_cast<std::size_t>(r)]);
    return t;
}

/// Row-major coordinate list with ids of the value expressions.
struct SparseEntries {
    int rows = 0, cols = 0;
    std::vector<int> row, col;
    std::vector<Id> value;
    std::size_t Nnz() const {
        return value.size();
    }
    /// CSR row starts (size rows+1), the array Function's ctor derives at function.hpp:112-124.
    std::vector<int> RowStarts() const {
        std::vector<int> s(static_cast<std::size_t>(rows) + 1, 0);
        for (int r : row) ++s[static_cast<std::size_t>(r) + 1];
        for (int i = 0; i < rows; ++i) s[static_cast<std::size_t>(i) + 1] += s[static_cast<std::size_t>(i)];
        return s;
    }
};

class Differentiator {
  public:
    explicit Differentiator(Tape& tape) : g_{tape.graph}, t_{tape.inputs, tape.outputs} {
    }
    /// Operate directly on a graph (e.g. the thread's recording graph) with the given input nodes.
    Differentiator(Graph& graph, std::vector<Id> inputs, std::vector<Id> outputs = {}) : g_{graph}, t_{std::move(inputs), std::move(outputs)} {
    }

    /// Differentiable-dependency bitsets for nodes [0, limit): bit j of dep(node) is set iff the
    /// node's value can vary with input j through differentiable operations.  Comparison
    /// operands of CondExp and the argument of Sign carry no dependency (as in CppAD).
    void ComputeDependencies(Id limit) {
        const std::size_t nIn = t_.inputs.size();
        words_ = (nIn + 63) / 64;
        dep_.assign(static_cast<std::size_t>(limit) * words_, 0);
        for (Id id = 0; id < limit; ++id) {
            const Node& nd = g_.At(id);
            std::uint64_t* d = Dep(id);
            if (nd.op == Op::Input) {
                d[static_cast<std::size_t>(nd.a) / 64] |= (1ULL << (static_cast<std::size_t>(nd.a) % 64));
            } else if (nd.op == Op::Const || nd.op == Op::Sign) {
            } else if (IsCond(nd.op)) {
                Or(d, Dep(nd.c));
                Or(d, Dep(nd.d));
            } else {
                Or(d, Dep(nd.a));
                if (nd.b != kNoId) Or(d, Dep(nd.b));
            }
        }
        depLimit_ = limit;
    }

    bool DependsOn(Id node, int input) const {
        return (Dep(node)[static_cast<std::size_t>(input) / 64] >> (static_cast<std::size_t>(input) % 64)) & 1ULL;
    }

    /// Sparse Jacobian d outputs / d inputs[0, nCols).  Picks forward or reverse accumulation by
    /// counting which creates fewer new nodes (both are exact; CppADCodeGen makes the same kind of
    /// per-model choice).  mode: 0 = auto, 1 = forward, 2 = reverse.
    SparseEntries Jacobian(int nCols, int mode = 0) {
        std::vector<int> cols(static_cast<std::size_t>(nCols));
        for (int j = 0; j < nCols; ++j) cols[static_cast<std::size_t>(j)] = j;
        return Jacobian(t_.outputs, cols, mode);
    }

    /// General form: d outs[i] / d input(cols[j]); entry (i, j) indexes into `outs` / `cols`.
    SparseEntries Jacobian(const std::vector<Id>& outs, const std::vector<int>& cols, int mode = 0) {
        const Id limit = static_cast<Id>(g_.Size());
        ComputeDependencies(limit);
        SparseEntries e;
        e.rows = static_cast<int>(outs.size());
        e.cols = static_cast<int>(cols.size());
        for (int i = 0; i < e.rows; ++i)
            for (int j = 0; j < e.cols; ++j)
                if (DependsOn(outs[static_cast<std::size_t>(i)], cols[static_cast<std::size_t>(j)])) {
                    e.row.push_back(i);
                    e.col.push_back(j);
                }
        e.value.assign(e.row.size(), kNoId);
        // restrict all work to the sub-DAG the requested outputs can reach
        active_.assign(static_cast<std::size_t>(limit), 0);
        for (Id o : outs) active_[static_cast<std::size_t>(o)] = 1;
        for (Id i = limit; i-- > 0;) {
            if (!active_[static_cast<std::size_t>(i)]) continue;
            const Node& nd = g_.At(i);
            if (Arity(nd.op) == 0) continue;
            for (Id o : {nd.a, nd.b, nd.c, nd.d})
                if (o != kNoId) active_[static_cast<std::size_t>(o)] = 1;
        }
        if (mode == 0) mode = EstimateForwardCost(limit, cols) <= EstimateReverseCost(limit, outs) ? 1 : 2;
        lastMode_ = mode;
        if (mode == 1) {
            std::vector<Id> d;
            for (int j = 0; j < e.cols; ++j) {
                bool any = false;
                for (std::size_t k = 0; k < e.col.size() && !any; ++k) any = e.col[k] == j;
                if (!any) continue;
                ForwardSweep(limit, cols[static_cast<std::size_t>(j)], d);
                for (std::size_t k = 0; k < e.col.size(); ++k)
                    if (e.col[k] == j) e.value[k] = d[static_cast<std::size_t>(outs[static_cast<std::size_t>(e.row[k])])];
            }
        } else {
            std::vector<Id> adj;
            std::size_t k = 0;
            for (int i = 0; i < e.rows; ++i) {
                if (k >= e.row.size() || e.row[k] != i) continue;
                ReverseSweep(limit, outs[static_cast<std::size_t>(i)], adj);
                for (; k < e.row.size() && e.row[k] == i; ++k)
                    e.value[k] = adj[static_cast<std::size_t>(t_.inputs[static_cast<std::size_t>(cols[static_cast<std::size_t>(e.col[k])])])];
            }
        }
        active_.clear();
        return e;
    }

    /// Replaces input nodes by expressions (inputIndex -> expression id) in the sub-DAGs under
    /// `roots`; returns the rewritten roots.  Used to compose separately differentiated stages.
    std::vector<Id> Substitute(const std::vector<Id>& roots, const std::vector<std::pair<int, Id>>& inputToExpr) {
        const Id limit = static_cast<Id>(g_.Size());
        std::vector<char> live(static_cast<std::size_t>(limit), 0);
        for (Id r : roots) live[static_cast<std::size_t>(r)] = 1;
        for (Id i = limit; i-- > 0;) {
            if (!live[static_cast<std::size_t>(i)]) continue;
            const Node& nd = g_.At(i);
            if (Arity(nd.op) == 0) continue;
            for (Id o : {nd.a, nd.b, nd.c, nd.d})
                if (o != kNoId) live[static_cast<std::size_t>(o)] = 1;
        }
        std::vector<Id> map(static_cast<std::size_t>(limit), kNoId);
        for (Id i = 0; i < limit; ++i) {
            if (!live[static_cast<std::size_t>(i)]) continue;
            const Node nd = g_.At(i);
            Id r = i;
            if (nd.op == Op::Input) {
                for (const auto& [idx, expr] : inputToExpr)
                    if (idx == nd.a) r = expr;
            } else if (nd.op != Op::Const) {
                auto m = [&](Id o) { return o == kNoId ? kNoId : map[static_cast<std::size_t>(o)]; };
                const Id a = m(nd.a), b = m(nd.b), c = m(nd.c), d = m(nd.d);
                if (a != nd.a || b != nd.b || c != nd.c || d != nd.d) {
                    switch (Arity(nd.op)) {
                        case 1: r = g_.Unary(nd.op, a); break;
                        case 2: r = g_.Binary(nd.op, a, b); break;
                        default: r = g_.Cond(nd.op, a, b, c, d);
                    }
                }
            }
            map[static_cast<std::size_t>(i)] = r;
        }
        std::vector<Id> out;
        out.reserve(roots.size());
        for (Id r : roots) out.push_back(map[static_cast<std::size_t>(r)]);
        return out;
    }

    /// Sparse upper-triangular Hessian of outputs[outputIndex] over inputs[0, nCols):
    /// reverse sweep for the gradient, then one forward sweep per column over the gradient DAG.
    SparseEntries Hessian(int outputIndex, int nCols) {
        const Id primalLimit = static_cast<Id>(g_.Size());
        ComputeDependencies(primalLimit);
        std::vector<Id> adj;
        ReverseSweep(primalLimit, t_.outputs[static_cast<std::size_t>(outputIndex)], adj);
        std::vector<Id> grad(static_cast<std::size_t>(nCols));
        for (int j = 0; j < nCols; ++j) grad[static_cast<std::size_t>(j)] = adj[static_cast<std::size_t>(t_.inputs[static_cast<std::size_t>(j)])];
        const Id limit = static_cast<Id>(g_.Size());
        ComputeDependencies(limit);
        SparseEntries e;
        e.rows = e.cols = nCols;
        for (int r = 0; r < nCols; ++r)
            for (int c = r; c < nCols; ++c)
                if (DependsOn(grad[static_cast<std::size_t>(r)], c)) {
                    e.row.push_back(r);
                    e.col.push_back(c);
                }
        e.value.assign(e.row.size(), kNoId);
        std::vector<Id> d;
        for (int c = 0; c < nCols; ++c) {
            bool any = false;
            for (std::size_t k = 0; k < e.col.size() && !any; ++k) any = e.col[k] == c;
            if (!any) continue;
            ForwardSweep(limit, c, d);
            for (std::size_t k = 0; k < e.col.size(); ++k)
                if (e.col[k] == c) e.value[k] = d[static_cast<std::size_t>(grad[static_cast<std::size_t>(e.row[k])])];
        }
        return e;
    }

    int LastMode() const {
        return lastMode_;
    }

  private:
    std::uint64_t* Dep(Id id) {
        return dep_.data() + static_cast<std::size_t>(id) * words_;
    }
    const std::uint64_t* Dep(Id id) const {
        return dep_.data() + static_cast<std::size_t>(id) * words_;
    }
    void Or(std::uint64_t* dst, const std::uint64_t* src) const {
        for (std::size_t w = 0; w < words_; ++w) dst[w] |= src[w];
    }

    double EstimateForwardCost(Id limit, const std::vector<int>& cols) const {
        std::vector<std::uint64_t> mask(words_, 0);
        for (int c : cols) mask[static_cast<std::size_t>(c) / 64] |= 1ULL << (static_cast<std::size_t>(c) % 64);
        // only nodes that some requested output can reach matter; approximate with all nodes
        double c = 0;
        for (Id id = 0; id < limit; ++id) {
            if (Arity(g_.At(id).op) == 0 || (!active_.empty() && !active_[static_cast<std::size_t>(id)])) continue;
            for (std::size_t w = 0; w < words_; ++w) c += static_cast<double>(__builtin_popcountll(Dep(id)[w] & mask[w]));
        }
        return c;
    }

    double EstimateReverseCost(Id limit, const std::vector<Id>& outs) const {
        // #outputs reaching each node, by reverse propagation of output-membership bitsets.
        const std::size_t m = outs.size();
        const std::size_t w = (m + 63) / 64;
        std::vector<std::uint64_t> reach(static_cast<std::size_t>(limit) * w, 0);
        for (std::size_t i = 0; i < m; ++i)
            reach[static_cast<std::size_t>(outs[i]) * w + i / 64] |= 1ULL << (i % 64);
        double c = 0;
        for (Id id = limit; id-- > 0;) {
            const Node& nd = g_.At(id);
            if (Arity(nd.op) == 0) continue;
            const std::uint64_t* r = reach.data() + static_cast<std::size_t>(id) * w;
            std::size_t cnt = 0;
            for (std::size_t k = 0; k < w; ++k) cnt += static_cast<std::size_t>(__builtin_popcountll(r[k]));
            if (!cnt) continue;
            c += static_cast<double>(cnt);
            auto push = [&](Id o) {
                if (o == kNoId) return;
                std::uint64_t* ro = reach.data() + static_cast<std::size_t>(o) * w;
                for (std::size_t k = 0; k < w; ++k) ro[k] |= r[k];
            };
            if (IsCond(nd.op)) {
                push(nd.c);
                push(nd.d);
            } else if (nd.op != Op::Sign) {
                push(nd.a);
                push(nd.b);
            }
        }
        return c;
    }

    /// d(node)/d(operand) for unary/binary nodes, as expression ids (cached per node).
    const std::array<Id, 2>& Partials(Id id) {
        if (partials_.size() < g_.Size()) partials_.resize(g_.Size(), {kNoId, kNoId});
        if (partials_[static_cast<std::size_t>(id)][0] != kNoId || partials_[static_cast<std::size_t>(id)][1] != kNoId)
            return partials_[static_cast<std::size_t>(id)];
        const Node nd = g_.At(id);
        const Id one = g_.Constant(1.0);
        Id pa = kNoId, pb = kNoId;
        switch (nd.op) {
            case Op::Add: pa = one; pb = one; break;
            case Op::Sub: pa = one; pb = g_.Constant(-1.0); break;
            case Op::Mul: pa = nd.b; pb = nd.a; break;
            case Op::Div: {
                const Id inv = g_.Div(one, nd.b);
                pa = inv;
                pb = g_.Neg(g_.Mul(id, inv));
                break;
            }
            case Op::Neg: pa = g_.Constant(-1.0); break;
            case Op::Sin: pa = g_.Unary(Op::Cos, nd.a); break;
            case Op::Cos: pa = g_.Neg(g_.Unary(Op::Sin, nd.a)); break;
            case Op::Tan: pa = g_.Add(one, g_.Mul(id, id)); break;
            case Op::Asin: pa = g_.Div(one, g_.Unary(Op::Sqrt, g_.Sub(one, g_.Mul(nd.a, nd.a)))); break;
            case Op::Acos: pa = g_.Neg(g_.Div(one, g_.Unary(Op::Sqrt, g_.Sub(one, g_.Mul(nd.a, nd.a))))); break;
            case Op::Atan: pa = g_.Div(one, g_.Add(one, g_.Mul(nd.a, nd.a))); break;
            case Op::Exp: pa = id; break;
            case Op::Log: pa = g_.Div(one, nd.a); break;
            case Op::Sqrt: pa = g_.Div(g_.Constant(0.5), id); break;
            case Op::Abs: pa = g_.Unary(Op::Sign, nd.a); break;
            case Op::Sign: break;
            case Op::QuadSum:
            case Op::QuadRot1:
            case Op::QuadRot2:
            case Op::QuadRot3:
                throw std::logic_error("tape: cannot differentiate through a quad communication op; differentiate the lane-local stage and substitute");
            case Op::Pow: {
                if (g_.IsConst(nd.b)) {
                    const double e = g_.ConstValue(nd.b);
                    pa = g_.Mul(nd.b, g_.Binary(Op::Pow, nd.a, g_.Constant(e - 1.0)));
                } else {
                    pa = g_.Mul(nd.b, g_.Binary(Op::Pow, nd.a, g_.Sub(nd.b, one)));
                    pb = g_.Mul(id, g_.Unary(Op::Log, nd.a));
                }
                break;
            }
            case Op::Atan2: {
                const Id den = g_.Add(g_.Mul(nd.a, nd.a), g_.Mul(nd.b, nd.b));
                const Id inv = g_.Div(one, den);
                pa = g_.Mul(nd.b, inv);
                pb = g_.Neg(g_.Mul(nd.a, inv));
                break;
            }
            default: break;
        }
        if (partials_.size() < g_.Size()) partials_.resize(g_.Size(), {kNoId, kNoId});
        partials_[static_cast<std::size_t>(id)] = {pa, pb};
        return partials_[static_cast<std::size_t>(id)];
    }

    /// Tangent of every node in [0, limit) along input `j`; d[node] is the id of the tangent
    /// expression (constant 0 where the node does not depend on j).
    void ForwardSweep(Id limit, int j, std::vector<Id>& d) {
        const Id zero = g_.Constant(0.0);
        d.assign(static_cast<std::size_t>(limit), zero);
        for (Id id = 0; id < limit; ++id) {
            if (id < depLimit_ && !DependsOn(id, j)) continue;
            if (!active_.empty() && !active_[static_cast<std::size_t>(id)]) continue;
            const Node nd = g_.At(id);
            if (nd.op == Op::Input) {
                if (nd.a == j) d[static_cast<std::size_t>(id)] = g_.Constant(1.0);
                continue;
            }
            if (nd.op == Op::Const || nd.op == Op::Sign) continue;
            if (IsCond(nd.op)) {
                d[static_cast<std::size_t>(id)] =
                    g_.Cond(nd.op, nd.a, nd.b, d[static_cast<std::size_t>(nd.c)], d[static_cast<std::size_t>(nd.d)]);
                continue;
            }
            const std::array<Id, 2> p = Partials(id);
            Id acc = zero;
            if (p[0] != kNoId && d[static_cast<std::size_t>(nd.a)] != zero) acc = g_.Mul(p[0], d[static_cast<std::size_t>(nd.a)]);
            if (nd.b != kNoId && p[1] != kNoId && d[static_cast<std::size_t>(nd.b)] != zero)
                acc = g_.Add(acc, g_.Mul(p[1], d[static_cast<std::size_t>(nd.b)]));
            d[static_cast<std::size_t>(id)] = acc;
        }
    }

    /// w * p with CppAD's "absolute zero" semantics for the adjoint w (azmul in CppAD's reverse sweeps): when w is
    /// exactly zero the product is zero whatever p is.  Needed only below a conditional: the adjoint of the
    /// branch that was NOT selected is an exact zero, while that branch's partials may be NaN/Inf at the guarded
    /// point (y = x / CondExpGt(z, 0, sqrt(z), 1) at z <= 0 -- the guard pattern of
    /// autodiff/support/quaternion.hpp:38-60 and utils.hpp:731-736).  guardedExpr_ holds the adjoint expressions
    /// that can be such a zero; the result is registered there too.
    Id AdjointTimesPartial(Id p, Id w) {
        if (!guardedExpr_.count(w) || g_.IsConst(p)) {
            const Id r = g_.Mul(p, w);
            if (guardedExpr_.count(w) && !g_.IsConst(r)) guardedExpr_.insert(r);
            return r;
        }
        const Id zero = g_.Constant(0.0);
        const Node nw = g_.At(w);
        Id r;
        if (IsCond(nw.op) && (nw.c == zero || nw.d == zero))  // w = (a cmp b) ? c : 0: multiply inside the selection (no extra compare)
            r = g_.Cond(nw.op, nw.a, nw.b, nw.c == zero ? zero : AdjointTimesPartial(p, nw.c), nw.d == zero ? zero : AdjointTimesPartial(p, nw.d));
        else
            r = g_.Cond(Op::CondEq, w, zero, zero, g_.Mul(p, w));
        if (!g_.IsConst(r)) guardedExpr_.insert(r);
        return r;
    }
    void Accumulate(std::vector<Id>& adj, Id operand, Id contribution) {
        Id& slot = adj[static_cast<std::size_t>(operand)];
        const bool g = guardedExpr_.count(slot) || guardedExpr_.count(contribution);
        slot = g_.Add(slot, contribution);
        if (g && !g_.IsConst(slot)) guardedExpr_.insert(slot);
    }

    /// Adjoint of every node in [0, limit) with respect to `output`.
    void ReverseSweep(Id limit, Id output, std::vector<Id>& adj) {
        const Id zero = g_.Constant(0.0);
        adj.assign(static_cast<std::size_t>(limit), zero);
        guardedExpr_.clear();
        adj[static_cast<std::size_t>(output)] = g_.Constant(1.0);
        for (Id id = output + 1; id-- > 0;) {
            const Id w = adj[static_cast<std::size_t>(id)];
            if (w == zero) continue;
            const Node nd = g_.At(id);
            if (Arity(nd.op) == 0 || nd.op == Op::Sign) continue;
            if (IsCond(nd.op)) {
                const Id wc = g_.Cond(nd.op, nd.a, nd.b, w, zero), wd = g_.Cond(nd.op, nd.a, nd.b, zero, w);
                if (!g_.IsConst(wc)) guardedExpr_.insert(wc);
                if (!g_.IsConst(wd)) guardedExpr_.insert(wd);
                Accumulate(adj, nd.c, wc);
                Accumulate(adj, nd.d, wd);
                continue;
            }
            const std::array<Id, 2> p = Partials(id);
            if (p[0] != kNoId) Accumulate(adj, nd.a, AdjointTimesPartial(p[0], w));
            if (nd.b != kNoId && p[1] != kNoId) Accumulate(adj, nd.b, AdjointTimesPartial(p[1], w));
        }
    }

    struct Io {
        std::vector<Id> inputs, outputs;
    };
    Graph& g_;
    Io t_;
    std::vector<std::uint64_t> dep_;
    std::size_t words_ = 1;
    Id depLimit_ = 0;
    std::vector<std::array<Id, 2>> partials_;
    std::unordered_set<Id> guardedExpr_;  // reverse sweep: adjoint expressions that may be the exact zero of a non-selected branch
    std::vector<char> active_;  // when non-empty: nodes the current Jacobian request can reach
    int lastMode_ = 0;
};

}  // namespace ungar_amd::tape
