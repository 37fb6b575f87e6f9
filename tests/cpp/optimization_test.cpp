// Host-only checks of the optimisation layer (ungar/optimization/*.hpp); no GPU needed.
//   * relaxed barriers: values on a grid (printed for the Python side, which restates the reference's
//     formulas of soft_inequality_constraint.hpp:77-205 independently), C2 continuity at the joints,
//     closed-form derivatives against central differences, tape recording = real evaluation;
//   * KktSolver: random sparse equality-constrained QPs against their KKT residuals and a dense solve
//     done by the Python side (the matrices are printed);
//   * BacktrackingLineSearch: the three acceptance branches (backtracking_line_search.hpp:120-147).
#include <cmath>
#include <cstdio>
#include <random>

#include "ungar/optimization/backtracking_line_search.hpp"
#include "ungar/optimization/kkt_solver.hpp"
#include "ungar/optimization/soft_equality_constraint.hpp"
#include "ungar/optimization/soft_inequality_constraint.hpp"

using namespace Ungar;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                               \
    do {                                                                \
        if (!(cond)) {                                                  \
            ++g_failures;                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
        }                                                               \
    } while (0)

template <class B>
static void CheckBarrier(const char* tag, const B& barrier, real_t lo, real_t hi, real_t h) {
    for (int i = 0; i <= 40; ++i) {
        const real_t x = lo + (hi - lo) * (i + 0.37) / 41.0;  // never on a joint of the piecewise definition (third derivative jumps there)
        const real_t value = barrier.Evaluate(x);
        std::printf("%s %.17g %.17g\n", tag, x, value);
        const real_t fd1 = (barrier.Evaluate(x + h) - barrier.Evaluate(x - h)) / (2 * h);
        const real_t fd2 = (barrier.FirstDerivative(x + h) - barrier.FirstDerivative(x - h)) / (2 * h);
        EXPECT_TRUE(std::fabs(fd1 - barrier.FirstDerivative(x)) <= 1e-6 * (1.0 + std::fabs(fd1)));
        EXPECT_TRUE(std::fabs(fd2 - barrier.SecondDerivative(x)) <= 1e-5 * (1.0 + std::fabs(fd2)));
        // recorded on a tape and read back (literal arithmetic): same number
        const ad_scalar_t onTape = barrier.Evaluate(ad_scalar_t{x});
        EXPECT_TRUE(std::fabs(::ungar_amd::tape::Value(onTape) - value) <= 1e-15 * (1.0 + std::fabs(value)));
    }
}

int main() {
    // ---- barriers ---------------------------------------------------------------------------------
    const RelaxedPolyBarrierFunction poly{0.5, 100.0, 2e-2};
    CheckBarrier("POLY", poly, 0.4, 0.6, 1e-6);
    EXPECT_TRUE(poly.Evaluate(0.5 + 2e-2) == 0.0 && poly.Evaluate(1.0) == 0.0);
    EXPECT_TRUE(std::fabs(poly.Evaluate(0.5 - 1e-12) - poly.Evaluate(0.5 + 1e-12)) < 1e-9);  // continuity at x = 0
    const RelaxedLogBarrierFunction logb{-1.0, 1e-2, 0.5};
    CheckBarrier("LOG", logb, -1.5, 1.0, 1e-6);
    EXPECT_TRUE(std::fabs(logb.Evaluate(-0.5 - 1e-12) - logb.Evaluate(-0.5 + 1e-12)) < 1e-9);  // continuity at x = epsilon
    const SoftBoundConstraint bound{-1.0, 2.0, 10.0};
    for (int i = 0; i <= 40; ++i) {
        const real_t x = -1.5 + 4.0 * i / 40.0;
        std::printf("BOUND %.17g %.17g\n", x, bound.Evaluate(x));
    }
    EXPECT_TRUE(bound.Evaluate(0.5) == 0.0 && bound.Evaluate(-1.2) > 0.0 && bound.Evaluate(2.2) > 0.0);
    EXPECT_TRUE((SoftEqualityConstraint{1.0, 4.0}.Evaluate(1.5) == 0.5));
    VectorXr three{3};
    three[0] = 0.45, three[1] = 0.5, three[2] = 0.7;
    EXPECT_TRUE(std::fabs(poly.Evaluate(three) - (poly.Evaluate(0.45) + poly.Evaluate(0.5) + poly.Evaluate(0.7))) < 1e-15);
    EXPECT_TRUE(LogisticFunction::SmoothGreaterThan(5.0, 0.0, 10.0) > 0.999 && LogisticFunction::SmoothLessThan(5.0, 0.0, 10.0) < 1e-3);
    // reference defaults (soft_inequality_constraint.hpp:36-59): midpoint 0, steepness 1 (constructor) / 1024 (smooth comparisons)
    EXPECT_TRUE(std::fabs(LogisticFunction{}.Evaluate(0.3) - 1.0 / (1.0 + std::exp(-0.3))) < 1e-15);
    EXPECT_TRUE(std::fabs(LogisticFunction::SmoothGreaterThan(1e-3, 0.0) - 1.0 / (1.0 + std::exp(-1.024))) < 1e-15);
    EXPECT_TRUE(std::fabs(LogisticFunction::SmoothLessThan(1e-3, 0.0) - (1.0 - 1.0 / (1.0 + std::exp(-1.024)))) < 1e-15);
    {  // the reference's explicit-scalar vector form, barrier.Evaluate<Scalar>(vector) (:90-95, :157-168, :225-228)
        VectorXr v{3};
        v << 0.5, -0.25, 1e-5;
        const RelaxedPolyBarrierFunction poly{0.0, 10.0, 1e-2};
        const RelaxedLogBarrierFunction logb{0.0, 1e-2, 0.1};
        const SoftBoundConstraint bound{-1.0, 1.0, 5.0};
        EXPECT_TRUE(poly.Evaluate<real_t>(v) == poly.Evaluate(v[0]) + poly.Evaluate(v[1]) + poly.Evaluate(v[2]));
        EXPECT_TRUE((poly.Evaluate<real_t, true>(v)) == (poly.Evaluate<real_t, true>(v[0]) + poly.Evaluate<real_t, true>(v[1]) + poly.Evaluate<real_t, true>(v[2])));
        EXPECT_TRUE(logb.Evaluate<real_t>(v) == logb.Evaluate(v[0]) + logb.Evaluate(v[1]) + logb.Evaluate(v[2]));
        EXPECT_TRUE(bound.Evaluate<real_t>(v) == bound.Evaluate(v[0]) + bound.Evaluate(v[1]) + bound.Evaluate(v[2]));
        VectorXad va = v.cast<ad_scalar_t>();
        EXPECT_TRUE(std::fabs(::ungar_amd::tape::Value(poly.Evaluate<ad_scalar_t>(va)) - poly.Evaluate<real_t>(v)) < 1e-15);
        EXPECT_TRUE(std::fabs(::ungar_amd::tape::Value(bound.Evaluate<ad_scalar_t>(va)) - bound.Evaluate<real_t>(v)) < 1e-15);
    }

    // ---- KKT solver -------------------------------------------------------------------------------
    std::mt19937 gen{7};
    std::uniform_real_distribution<real_t> U{-1.0, 1.0};
    for (int trial = 0; trial < 3; ++trial) {
        const index_t n = 40 + 15 * trial, m = 12 + 5 * trial;
        // H = banded SPD (upper triangle), A = sparse full-row-rank-ish
        std::vector<std::vector<std::pair<int, real_t>>> rows(static_cast<std::size_t>(n));
        for (index_t r = 0; r < n; ++r) {
            rows[static_cast<std::size_t>(r)].emplace_back(static_cast<int>(r), 4.0 + U(gen));
            for (index_t c = r + 1; c < std::min(n, r + 4); ++c) rows[static_cast<std::size_t>(r)].emplace_back(static_cast<int>(c), 0.5 * U(gen));
        }
        std::vector<int> hs{0}, hc;
        std::vector<real_t> hv;
        for (auto& row : rows) {
            for (auto& [c, v] : row) hc.push_back(c), hv.push_back(v);
            hs.push_back(static_cast<int>(hc.size()));
        }
        std::vector<int> as{0}, ac;
        std::vector<real_t> av;
        for (index_t r = 0; r < m; ++r) {
            for (index_t c = (3 * r) % n; c < n; c += 7 + r % 3) ac.push_back(static_cast<int>(c)), av.push_back(U(gen));
            as.push_back(static_cast<int>(ac.size()));
        }
        std::vector<real_t> g(static_cast<std::size_t>(n)), b(static_cast<std::size_t>(m)), d, lambda;
        for (auto& v : g) v = U(gen);
        for (auto& v : b) v = U(gen);
        KktSolver kkt;
        kkt.Solve(n, hs, hc, hv, g.data(), m, as.data(), ac.data(), av.data(), b.data(), d, lambda);
        // residuals: H d + A^T lambda + g = 0,  A d = b
        std::vector<real_t> r1(g), r2(static_cast<std::size_t>(m), 0.0);
        for (index_t r = 0; r < n; ++r)
            for (int k = hs[static_cast<std::size_t>(r)]; k < hs[static_cast<std::size_t>(r) + 1]; ++k) {
                const std::size_t c = static_cast<std::size_t>(hc[static_cast<std::size_t>(k)]);
                r1[static_cast<std::size_t>(r)] += hv[static_cast<std::size_t>(k)] * d[c];
                if (c != static_cast<std::size_t>(r)) r1[c] += hv[static_cast<std::size_t>(k)] * d[static_cast<std::size_t>(r)];
            }
        for (index_t r = 0; r < m; ++r)
            for (int k = as[static_cast<std::size_t>(r)]; k < as[static_cast<std::size_t>(r) + 1]; ++k) {
                const std::size_t c = static_cast<std::size_t>(ac[static_cast<std::size_t>(k)]);
                r1[c] += av[static_cast<std::size_t>(k)] * lambda[static_cast<std::size_t>(r)];
                r2[static_cast<std::size_t>(r)] += av[static_cast<std::size_t>(k)] * d[c];
            }
        real_t worst = 0;
        for (real_t v : r1) worst = std::max(worst, std::fabs(v));
        for (index_t r = 0; r < m; ++r) worst = std::max(worst, std::fabs(r2[static_cast<std::size_t>(r)] - b[static_cast<std::size_t>(r)]));
        EXPECT_TRUE(worst < 1e-10);
        // dump the problem and the solution for the dense cross-check
        std::printf("QP %td %td\n", n, m);
        for (index_t r = 0; r < n; ++r)
            for (int k = hs[static_cast<std::size_t>(r)]; k < hs[static_cast<std::size_t>(r) + 1]; ++k)
                std::printf("H %td %d %.17g\n", r, hc[static_cast<std::size_t>(k)], hv[static_cast<std::size_t>(k)]);
        for (index_t r = 0; r < m; ++r)
            for (int k = as[static_cast<std::size_t>(r)]; k < as[static_cast<std::size_t>(r) + 1]; ++k)
                std::printf("A %td %d %.17g\n", r, ac[static_cast<std::size_t>(k)], av[static_cast<std::size_t>(k)]);
        for (index_t i = 0; i < n; ++i) std::printf("g %td %.17g\nd %td %.17g\n", i, g[static_cast<std::size_t>(i)], i, d[static_cast<std::size_t>(i)]);
        for (index_t i = 0; i < m; ++i) std::printf("b %td %.17g\n", i, b[static_cast<std::size_t>(i)]);
        std::printf("ENDQP\n");
    }

    // ---- line search --------------------------------------------------------------------------------
    {
        // feasible quadratic bowl: Armijo branch accepts the full Newton step
        VectorXr w{2};
        w[0] = 1.0, w[1] = -2.0;
        const std::vector<real_t> grad{2.0, -4.0}, step{-1.0, 2.0};
        const bool ok = BacktrackingLineSearch{false}.Do(
            grad, step, [](const VectorXr& x) { return x.squaredNorm(); }, [](const VectorXr&) { return 0.0; }, w);
        EXPECT_TRUE(ok && std::fabs(w[0]) < 1e-15 && std::fabs(w[1]) < 1e-15);
        // infeasible start: a step is accepted only if it reduces the violation (it must backtrack once here)
        VectorXr z{1};
        z[0] = 1.0;
        const std::vector<real_t> g1{1.0}, s1{-4.0};
        const bool ok2 = BacktrackingLineSearch{false}.Do(
            g1, s1, [](const VectorXr& x) { return x[0]; }, [](const VectorXr& x) { return std::fabs(x[0] + 0.5); }, z);
        EXPECT_TRUE(ok2 && std::fabs(z[0] - (-1.0)) < 1e-15);  // alpha = 0.5: |1 - 2 + 0.5| = 0.5 < 1.5; alpha = 1 gives 2.5
        // an ascent direction with zero violation is rejected down to alphaMin
        VectorXr y{1};
        y[0] = 0.0;
        const std::vector<real_t> g2{1.0}, s2{1.0};
        const bool ok3 = BacktrackingLineSearch{false}.Do(
            g2, s2, [](const VectorXr& x) { return x[0] + 1.0; }, [](const VectorXr&) { return 0.0; }, y);
        EXPECT_TRUE(!ok3 && y[0] == 0.0);
    }
    std::printf(g_failures ? "FAILED %d\n" : "PASSED\n", g_failures);
    return g_failures ? 1 : 0;
}
