// ungar_amd :: direct solver for the equality-constrained QP of one SQP iteration
//
//     min_d  1/2 d^T H d + g^T d    s.t.  A d = b          (H symmetric positive definite, upper triangle given)
//
// replacing the reference's OSQP instance (soft_sqp.hpp:143-158, 186-232: lower == upper bounds, i.e. the
// QP is purely equality constrained).  The KKT system  [H A^T; A -delta I] [d; lambda] = [-g; b]  is
// quasi-definite, so a sparse L D L^T without pivoting exists for ANY symmetric ordering:
//   * ordering: reverse Cuthill-McKee on the KKT graph (an optimal-control problem's KKT matrix becomes
//     banded with bandwidth ~ 2 (nx + nu) without being told about stages);
//   * symbolic analysis once per sparsity pattern (elimination tree, column counts);
//   * numeric up-looking L D L^T + iterative refinement against the delta = 0 system until the correction stalls.
// Everything is plain C++ on the host: this is the control loop AROUND the device hot path (per-node
// derivatives are what is evaluated on the GPU), sized for one instance.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <queue>
#include <stdexcept>
#include <vector>

#include "../data_types.hpp"

namespace Ungar {

class KktSolver {
  public:
    /// H: n x n, upper triangle in CSR (entries below the diagonal are ignored); A: m x n CSR.
    /// The patterns may change between calls (the symbolic phase is redone when they do).
    void Solve(index_t n, const std::vector<int>& hStarts, const std::vector<int>& hCols, const std::vector<real_t>& hValues, const real_t* g,
               index_t m, const int* aStarts, const int* aCols, const real_t* aValues, const real_t* b, std::vector<real_t>& d,
               std::vector<real_t>& lambda) {
        const index_t N = n + m;
        // ---- upper triangle of K in coordinate form (row <= col), columns of A^T placed at n + constraint
        std::vector<int> rows, cols;
        std::vector<real_t> vals;
        for (index_t r = 0; r < n; ++r)
            for (int k = hStarts[static_cast<std::size_t>(r)]; k < hStarts[static_cast<std::size_t>(r) + 1]; ++k)
                if (hCols[static_cast<std::size_t>(k)] >= r) {
                    rows.push_back(static_cast<int>(r));
                    cols.push_back(hCols[static_cast<std::size_t>(k)]);
                    vals.push_back(hValues[static_cast<std::size_t>(k)]);
                }
        for (index_t c = 0; c < m; ++c) {
            for (int k = aStarts[c]; k < aStarts[c + 1]; ++k) {
                rows.push_back(aCols[k]);
                cols.push_back(static_cast<int>(n + c));
                vals.push_back(aValues[k]);
            }
            rows.push_back(static_cast<int>(n + c));
            cols.push_back(static_cast<int>(n + c));
            vals.push_back(-kDelta);
        }
        if (N != _N || rows != _rows || cols != _cols) Analyse(N, rows, cols);
        Factorise(vals);

        std::vector<real_t> rhs(static_cast<std::size_t>(N)), x(static_cast<std::size_t>(N), 0.0), res(static_cast<std::size_t>(N)), corr;
        for (index_t i = 0; i < n; ++i) rhs[static_cast<std::size_t>(i)] = -g[i];
        for (index_t i = 0; i < m; ++i) rhs[static_cast<std::size_t>(n + i)] = b[i];
        SolveFactored(rhs, x);
        // Refinement against the UNREGULARISED system until the correction stalls.  Two fixed steps were not enough for the
        // reference's quadruped OCP with all feet in stance (360 active contact rows, force curvature 1e-6 next to foothold
        // curvature 2, cond(K) ~ 2e8): the step kept a relative error of 3e-7 in the force directions; with the loop below it reaches
        // 1e-12 (tools/qp_accuracy.py measures both QP solvers against an extended-precision solution).
        real_t previous = 0.0;
        for (int it = 0; it < kMaxRefinements; ++it) {
            res = rhs;
            for (std::size_t e = 0; e < vals.size(); ++e) {
                const std::size_t r = static_cast<std::size_t>(_rows[e]), c = static_cast<std::size_t>(_cols[e]);
                const real_t v = (r == c && r >= static_cast<std::size_t>(n)) ? 0.0 : vals[e];
                res[r] -= v * x[c];
                if (r != c) res[c] -= v * x[r];
            }
            SolveFactored(res, corr);
            real_t size = 0.0, scale = 0.0;
            for (std::size_t i = 0; i < x.size(); ++i) {
                size = std::max(size, std::abs(corr[i]));
                scale = std::max(scale, std::abs(x[i]));
            }
            if (!std::isfinite(size) || (it > 0 && size > 0.5 * previous)) break;  // no longer contracting: keep the iterate
            for (std::size_t i = 0; i < x.size(); ++i) x[i] += corr[i];
            if (size <= 1e-16 * scale) break;
            previous = size;
        }
        d.assign(x.begin(), x.begin() + n);
        lambda.assign(x.begin() + n, x.end());
    }

    index_t FactorNonZeros() const {
        return static_cast<index_t>(_Li.size());
    }

  private:
    static constexpr real_t kDelta = 1e-9;
    static constexpr int kMaxRefinements = 12;

    void Analyse(index_t N, const std::vector<int>& rows, const std::vector<int>& cols) {
        _N = N;
        _rows = rows;
        _cols = cols;
        const std::size_t n = static_cast<std::size_t>(N);
        // adjacency of the symmetric graph
        std::vector<std::vector<int>> adj(n);
        for (std::size_t e = 0; e < rows.size(); ++e)
            if (rows[e] != cols[e]) {
                adj[static_cast<std::size_t>(rows[e])].push_back(cols[e]);
                adj[static_cast<std::size_t>(cols[e])].push_back(rows[e]);
            }
        for (auto& a : adj) {
            std::sort(a.begin(), a.end());
            a.erase(std::unique(a.begin(), a.end()), a.end());
        }
        // reverse Cuthill-McKee, every connected component started from a minimum-degree vertex
        std::vector<int> order;
        order.reserve(n);
        std::vector<char> seen(n, 0);
        std::vector<int> byDegree(n);
        std::iota(byDegree.begin(), byDegree.end(), 0);
        std::stable_sort(byDegree.begin(), byDegree.end(), [&](int a, int b) { return adj[static_cast<std::size_t>(a)].size() < adj[static_cast<std::size_t>(b)].size(); });
        for (int start : byDegree) {
            if (seen[static_cast<std::size_t>(start)]) continue;
            std::queue<int> q;
            q.push(start);
            seen[static_cast<std::size_t>(start)] = 1;
            while (!q.empty()) {
                const int v = q.front();
                q.pop();
                order.push_back(v);
                std::vector<int> next;
                for (int w : adj[static_cast<std::size_t>(v)])
                    if (!seen[static_cast<std::size_t>(w)]) {
                        seen[static_cast<std::size_t>(w)] = 1;
                        next.push_back(w);
                    }
                std::stable_sort(next.begin(), next.end(), [&](int a, int b) { return adj[static_cast<std::size_t>(a)].size() < adj[static_cast<std::size_t>(b)].size(); });
                for (int w : next) q.push(w);
            }
        }
        std::reverse(order.begin(), order.end());
        _perm = order;  // new index k holds old index _perm[k]
        _inv.assign(n, 0);
        for (std::size_t k = 0; k < n; ++k) _inv[static_cast<std::size_t>(_perm[k])] = static_cast<int>(k);

        // permuted upper triangle in compressed-column form; _slot[e] = where entry e lands
        std::vector<int> count(n + 1, 0);
        _slot.assign(rows.size(), 0);
        auto place = [&](std::size_t e) {
            int r = _inv[static_cast<std::size_t>(rows[e])], c = _inv[static_cast<std::size_t>(cols[e])];
            if (r > c) std::swap(r, c);
            return std::pair<int, int>{r, c};
        };
        for (std::size_t e = 0; e < rows.size(); ++e) ++count[static_cast<std::size_t>(place(e).second) + 1];
        _Ap.assign(n + 1, 0);
        for (std::size_t c = 0; c < n; ++c) _Ap[c + 1] = _Ap[c] + count[c + 1];
        _Ai.assign(rows.size(), 0);
        std::vector<int> fill(_Ap.begin(), _Ap.end() - 1);
        for (std::size_t e = 0; e < rows.size(); ++e) {
            const auto [r, c] = place(e);
            const int at = fill[static_cast<std::size_t>(c)]++;
            _Ai[static_cast<std::size_t>(at)] = r;
            _slot[e] = at;
        }
        // elimination tree and column counts of L (row-subtree traversal with path marking)
        _parent.assign(n, -1);
        std::vector<int> mark(n), lnz(n, 0);
        for (std::size_t k = 0; k < n; ++k) {
            mark[k] = static_cast<int>(k);
            for (int p = _Ap[k]; p < _Ap[k + 1]; ++p)
                for (int i = _Ai[static_cast<std::size_t>(p)]; i < static_cast<int>(k) && mark[static_cast<std::size_t>(i)] != static_cast<int>(k);
                     i = _parent[static_cast<std::size_t>(i)]) {
                    if (_parent[static_cast<std::size_t>(i)] < 0) _parent[static_cast<std::size_t>(i)] = static_cast<int>(k);
                    ++lnz[static_cast<std::size_t>(i)];
                    mark[static_cast<std::size_t>(i)] = static_cast<int>(k);
                }
        }
        _Lp.assign(n + 1, 0);
        for (std::size_t k = 0; k < n; ++k) _Lp[k + 1] = _Lp[k] + lnz[k];
        _Li.assign(static_cast<std::size_t>(_Lp[n]), 0);
        _Lx.assign(static_cast<std::size_t>(_Lp[n]), 0.0);
        _D.assign(n, 0.0);
    }

    void Factorise(const std::vector<real_t>& vals) {
        const std::size_t n = static_cast<std::size_t>(_N);
        std::vector<real_t> Ax(vals.size(), 0.0);
        for (std::size_t e = 0; e < vals.size(); ++e) Ax[static_cast<std::size_t>(_slot[e])] += vals[e];
        std::vector<real_t> y(n, 0.0);
        std::vector<int> mark(n), pattern(n), used(n, 0);
        for (std::size_t k = 0; k < n; ++k) {
            // non-zero pattern of row k of L = nodes reached from the entries of column k in the elimination tree
            std::size_t top = n;
            mark[k] = static_cast<int>(k);
            for (int p = _Ap[k]; p < _Ap[k + 1]; ++p) {
                int i = _Ai[static_cast<std::size_t>(p)];
                y[static_cast<std::size_t>(i)] += Ax[static_cast<std::size_t>(p)];
                std::size_t len = 0;
                for (; i < static_cast<int>(k) && mark[static_cast<std::size_t>(i)] != static_cast<int>(k); i = _parent[static_cast<std::size_t>(i)]) {
                    pattern[len++] = i;
                    mark[static_cast<std::size_t>(i)] = static_cast<int>(k);
                }
                while (len > 0) pattern[--top] = pattern[--len];
            }
            real_t dk = y[k];
            y[k] = 0.0;
            for (; top < n; ++top) {  // sparse triangular solve in topological order
                const std::size_t i = static_cast<std::size_t>(pattern[top]);
                const real_t yi = y[i];
                y[i] = 0.0;
                const int begin = _Lp[i], end = begin + used[i];
                for (int p = begin; p < end; ++p) y[static_cast<std::size_t>(_Li[static_cast<std::size_t>(p)])] -= _Lx[static_cast<std::size_t>(p)] * yi;
                const real_t lki = yi / _D[i];
                dk -= lki * yi;
                _Li[static_cast<std::size_t>(end)] = static_cast<int>(k);
                _Lx[static_cast<std::size_t>(end)] = lki;
                ++used[i];
            }
            if (dk == 0.0 || !std::isfinite(dk)) throw std::runtime_error("KktSolver: zero or non-finite pivot (is H positive definite?)");
            _D[k] = dk;
        }
    }

    /// x = P^T (L D L^T)^-1 P rhs
    void SolveFactored(const std::vector<real_t>& rhs, std::vector<real_t>& x) const {
        const std::size_t n = static_cast<std::size_t>(_N);
        std::vector<real_t> z(n);
        for (std::size_t k = 0; k < n; ++k) z[k] = rhs[static_cast<std::size_t>(_perm[k])];
        for (std::size_t j = 0; j < n; ++j)
            for (int p = _Lp[j]; p < _Lp[j + 1]; ++p) z[static_cast<std::size_t>(_Li[static_cast<std::size_t>(p)])] -= _Lx[static_cast<std::size_t>(p)] * z[j];
        for (std::size_t j = 0; j < n; ++j) z[j] /= _D[j];
        for (std::size_t j = n; j-- > 0;)
            for (int p = _Lp[j]; p < _Lp[j + 1]; ++p) z[j] -= _Lx[static_cast<std::size_t>(p)] * z[static_cast<std::size_t>(_Li[static_cast<std::size_t>(p)])];
        x.assign(n, 0.0);
        for (std::size_t k = 0; k < n; ++k) x[static_cast<std::size_t>(_perm[k])] = z[k];
    }

    index_t _N = -1;
    std::vector<int> _rows, _cols, _perm, _inv, _slot, _Ap, _Ai, _parent, _Lp, _Li;
    std::vector<real_t> _Lx, _D;
};

}  // namespace Ungar
