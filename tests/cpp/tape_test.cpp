// CPU test of the derivative transforms (ungar_amd/csrc/tape/derive.hpp) through a TEST-ONLY interpreter of the
// expression DAG.  The product has no CPU evaluation path (values are only ever produced by generated HIP code);
// this interpreter exists so that forward / reverse / Hessian programs can be pinned without a GPU:
//   * CppAD's "absolute zero" rule for reverse sweeps below a conditional (azmul): the reference's guard pattern
//     y = x / CondExpGt(z, 0, sqrt(z), 1)   (autodiff/support/quaternion.hpp:38-60, utils.hpp:731-736) must have a
//     finite gradient / Hessian at z <= 0 in BOTH accumulation modes;
//   * scalar helpers of utils.hpp:969-1021 (Min, Sign, Abs, SmoothMin, SmoothAbs): values and derivatives against
//     closed forms on either side of their switching points;
//   * forward and reverse accumulation agree entry for entry on a mixed expression.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "tape_interpreter.hpp"

using namespace ungar_amd::tape;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                               \
    do {                                                                \
        if (!(cond)) {                                                  \
            ++g_failures;                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
        }                                                               \
    } while (0)

template <class F>
static Tape Record(int n, F&& f) {
    std::vector<AD> x = Independent(n);
    return MakeTape(f(x));
}

static bool Close(double a, double b, double tol = 1e-12) {
    return std::isfinite(a) && std::isfinite(b) && std::fabs(a - b) <= tol * (1.0 + std::fabs(b));
}

static void TestAbsoluteZeroBelowConditionals() {
    // y = x1 / CondExpGt(z, 0, sqrt(z), 1)
    auto f = [](const std::vector<AD>& x) { return std::vector<AD>{x[0] / CondExpGt(x[1], AD{0.0}, sqrt(x[1]), AD{1.0})}; };
    for (int mode : {1, 2}) {
        Tape t = Record(2, f);
        Differentiator d{t};
        const SparseEntries J = d.Jacobian(2, mode);
        EXPECT_TRUE(d.LastMode() == mode);
        for (double z : {-2.0, 0.0, 4.0}) {
            const std::vector<double> v = Interpret(t.graph, {3.0, z});
            for (std::size_t k = 0; k < J.Nnz(); ++k) {
                const double got = v[static_cast<std::size_t>(J.value[k])];
                const double want = J.col[k] == 0 ? (z > 0 ? 1.0 / std::sqrt(z) : 1.0) : (z > 0 ? -0.5 * 3.0 / (z * std::sqrt(z)) : 0.0);
                EXPECT_TRUE(Close(got, want));
            }
        }
    }
    {  // Hessian (reverse, then forward over the gradient program)
        Tape t = Record(2, f);
        Differentiator d{t};
        const SparseEntries H = d.Hessian(0, 2);
        for (double z : {-2.0, 0.0, 4.0}) {
            const std::vector<double> v = Interpret(t.graph, {3.0, z});
            for (std::size_t k = 0; k < H.Nnz(); ++k) {
                const double got = v[static_cast<std::size_t>(H.value[k])];
                double want = 0.0;
                if (z > 0 && H.row[k] == 0 && H.col[k] == 1) want = -0.5 / (z * std::sqrt(z));
                if (z > 0 && H.row[k] == 1 && H.col[k] == 1) want = 0.75 * 3.0 / (z * z * std::sqrt(z));
                EXPECT_TRUE(Close(got, want));
            }
        }
    }
    {  // norm of the zero vector behind a guard, log and acos behind guards, adjoints that meet again below the conditional
        auto g = [](const std::vector<AD>& x) {
            const AD s = x[0] * x[0] + x[1] * x[1];
            const AD n = CondExpGt(s, AD{0.0}, sqrt(s), AD{0.0});
            const AD l = CondExpGt(x[2], AD{0.0}, log(x[2]) * x[0], x[0]);
            const AD a = CondExpLt(x[2], AD{1.0}, acos(x[2]) + x[1], x[1] * x[1]);
            return std::vector<AD>{n + l + a * n};
        };
        Tape t = Record(3, g);
        Differentiator d{t};
        const SparseEntries J = d.Jacobian(3, 2);
        const SparseEntries H = d.Hessian(0, 3);
        for (const std::vector<double>& in : {std::vector<double>{0.0, 0.0, -0.5}, {0.0, 0.0, 2.0}, {0.0, 0.0, 0.0}, {0.3, -0.2, 2.0}}) {
            const std::vector<double> v = Interpret(t.graph, in);
            for (Id id : J.value) EXPECT_TRUE(std::isfinite(v[static_cast<std::size_t>(id)]));
            for (Id id : H.value) EXPECT_TRUE(std::isfinite(v[static_cast<std::size_t>(id)]));
        }
    }
}

static void TestModesAgree() {
    auto f = [](const std::vector<AD>& x) {
        const AD r = sqrt(x[0] * x[0] + x[1] * x[1] + 1e-3);
        const AD m = CondExpGt(x[2], x[3], x[2] * sin(x[0]), x[3] / r);
        return std::vector<AD>{m * exp(x[1]) + atan2(x[0], x[3]), pow(r, 3) * cos(x[2]) - m, abs(x[1] - x[2]) * tan(x[0] * 0.3) + log(r), m * m};
    };
    Tape t1 = Record(4, f), t2 = Record(4, f);
    Differentiator d1{t1}, d2{t2};
    const SparseEntries F = d1.Jacobian(4, 1), R = d2.Jacobian(4, 2);
    EXPECT_TRUE(F.Nnz() == R.Nnz());
    std::mt19937 gen{7U};
    std::uniform_real_distribution<double> u{-1.5, 1.5};
    for (int s = 0; s < 50; ++s) {
        const std::vector<double> in{u(gen), u(gen), u(gen), u(gen)};
        const std::vector<double> v1 = Interpret(t1.graph, in), v2 = Interpret(t2.graph, in);
        for (std::size_t k = 0; k < F.Nnz() && k < R.Nnz(); ++k) {
            EXPECT_TRUE(F.row[k] == R.row[k] && F.col[k] == R.col[k]);
            EXPECT_TRUE(Close(v1[static_cast<std::size_t>(F.value[k])], v2[static_cast<std::size_t>(R.value[k])], 1e-11));
        }
    }
}

int main() {
    TestAbsoluteZeroBelowConditionals();
    TestModesAgree();
    std::printf(g_failures == 0 ? "tape_test OK\n" : "tape_test FAILED (%d)\n", g_failures);
    return g_failures == 0 ? 0 : 1;
}
