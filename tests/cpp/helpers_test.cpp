// CPU test of the scalar / quaternion helpers that model lambdas call while a tape is recorded (SURVEY.md section 8(a)
// row A10): every helper is recorded on ad_scalar_t, differentiated in BOTH accumulation modes, evaluated with the
// test-only interpreter on either side of its switching point, and compared with
//   * the same helper on real_t (value parity of the recorded expression with the host expression), and
//   * its closed-form derivative.
// Reference: include/ungar/utils/utils.hpp:731-749 (ApproximateNorm / ApproximateExponentialMap), :820-852 (Pow, Sqrt),
// :969-1021 (Min, SmoothMin, Sign, Abs, SmoothAbs); include/ungar/autodiff/support/quaternion.hpp:34-192 (AD-safe
// inverse / normalize(d) / slerp).
#include <cmath>
#include <cstdio>
#include <functional>
#include <random>
#include <vector>

#include "tape_interpreter.hpp"
#include "ungar/utils/utils.hpp"

using namespace Ungar;
namespace tape = ungar_amd::tape;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                               \
    do {                                                                \
        if (!(cond)) {                                                  \
            ++g_failures;                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
        }                                                               \
    } while (0)

static bool Close(double a, double b, double tol = 1e-12) {
    return std::isfinite(a) && std::isfinite(b) && std::fabs(a - b) <= tol * (1.0 + std::fabs(b));
}

struct Recorded {
    tape::Tape t;
    tape::SparseEntries J;
    /// value of output i and dense Jacobian row-major at `in`
    void Eval(const std::vector<double>& in, std::vector<double>& y, std::vector<double>& jac) const {
        const std::vector<double> v = tape::Interpret(t.graph, in);
        y.clear();
        for (tape::Id o : t.outputs) y.push_back(v[static_cast<std::size_t>(o)]);
        jac.assign(t.outputs.size() * in.size(), 0.0);
        for (std::size_t k = 0; k < J.Nnz(); ++k)
            jac[static_cast<std::size_t>(J.row[k]) * in.size() + static_cast<std::size_t>(J.col[k])] = v[static_cast<std::size_t>(J.value[k])];
    }
};

static Recorded Record(int n, const std::function<std::vector<ad_scalar_t>(const std::vector<ad_scalar_t>&)>& f, int mode) {
    Recorded r;
    const std::vector<ad_scalar_t> x = tape::Independent(n);
    r.t = tape::MakeTape(f(x));
    tape::Differentiator d{r.t};
    r.J = d.Jacobian(n, mode);
    return r;
}

/// Scalar helper of one or two arguments: value vs the real_t helper, derivative vs the closed form `dref`.
static void CheckScalarHelper(const char* name, int nIn, const std::function<ad_scalar_t(const std::vector<ad_scalar_t>&)>& fad,
                              const std::function<double(const std::vector<double>&)>& freal,
                              const std::function<std::vector<double>(const std::vector<double>&)>& dref, const std::vector<std::vector<double>>& points) {
    for (int mode : {1, 2}) {
        const Recorded r = Record(nIn, [&](const std::vector<ad_scalar_t>& x) { return std::vector<ad_scalar_t>{fad(x)}; }, mode);
        for (const std::vector<double>& in : points) {
            std::vector<double> y, jac;
            r.Eval(in, y, jac);
            const std::vector<double> want = dref(in);
            bool ok = Close(y[0], freal(in));
            for (int j = 0; j < nIn; ++j) ok = ok && Close(jac[static_cast<std::size_t>(j)], want[static_cast<std::size_t>(j)], 1e-11);
            if (!ok) std::printf("helper %s mode %d at (%g%s): value %.17g vs %.17g, d %.17g vs %.17g\n", name, mode, in[0], nIn > 1 ? ", ..." : "", y[0], freal(in), jac[0], want[0]);
            EXPECT_TRUE(ok);
        }
    }
}

static void TestScalarHelpers() {
    const double eps = std::numeric_limits<double>::epsilon();
    CheckScalarHelper("Min", 2, [](const auto& x) { return Utils::Min(x[0], x[1]); }, [](const auto& x) { return Utils::Min(x[0], x[1]); },
                      [](const auto& x) { return x[0] > x[1] ? std::vector<double>{0.0, 1.0} : std::vector<double>{1.0, 0.0}; },
                      {{-1.0, 2.0}, {2.0, -1.0}, {0.5, 0.5}, {-3.0, -3.5}});
    CheckScalarHelper("Sign", 1, [](const auto& x) { return Utils::Sign(x[0]) * x[0] * x[0]; },  // times x^2 so that a Jacobian entry exists
                      [](const auto& x) { return Utils::Sign(x[0]) * x[0] * x[0]; },
                      [](const auto& x) { return std::vector<double>{Utils::Sign(x[0]) * 2.0 * x[0]}; }, {{-1.5}, {0.0}, {2.5}});
    CheckScalarHelper("Abs", 1, [](const auto& x) { return Utils::Abs(x[0]); }, [](const auto& x) { return Utils::Abs(x[0]); },
                      [](const auto& x) { return std::vector<double>{static_cast<double>(x[0] > 0) - static_cast<double>(x[0] < 0)}; },
                      {{-1.5}, {0.0}, {2.5}});
    CheckScalarHelper("SmoothAbs", 1, [](const auto& x) { return Utils::SmoothAbs(x[0]); }, [](const auto& x) { return Utils::SmoothAbs(x[0]); },
                      [eps](const auto& x) { return std::vector<double>{x[0] / std::sqrt(x[0] * x[0] + eps)}; }, {{-1.5}, {0.0}, {1e-9}, {2.5}});
    CheckScalarHelper("SmoothAbs(eps)", 1, [](const auto& x) { return Utils::SmoothAbs(x[0], ad_scalar_t{1e-2}); },
                      [](const auto& x) { return Utils::SmoothAbs(x[0], 1e-2); },
                      [](const auto& x) { return std::vector<double>{x[0] / std::sqrt(x[0] * x[0] + 1e-2)}; }, {{-0.2}, {0.0}, {0.7}});
    auto smoothMinD = [](double alpha) {
        return [alpha](const std::vector<double>& x) {
            const double ea = std::exp(-alpha * x[0]), eb = std::exp(-alpha * x[1]), D = ea + eb, S = (x[0] * ea + x[1] * eb) / D;
            return std::vector<double>{ea / D * (1.0 - alpha * x[0] + alpha * S), eb / D * (1.0 - alpha * x[1] + alpha * S)};
        };
    };
    CheckScalarHelper("SmoothMin", 2, [](const auto& x) { return Utils::SmoothMin(x[0], x[1]); }, [](const auto& x) { return Utils::SmoothMin(x[0], x[1]); },
                      smoothMinD(8.0), {{-1.0, 2.0}, {2.0, -1.0}, {0.5, 0.5}, {0.1, 0.12}});
    CheckScalarHelper("SmoothMin(alpha)", 2, [](const auto& x) { return Utils::SmoothMin(x[0], x[1], ad_scalar_t{2.5}); },
                      [](const auto& x) { return Utils::SmoothMin(x[0], x[1], 2.5); }, smoothMinD(2.5), {{-1.0, 2.0}, {0.3, 0.25}});
    CheckScalarHelper("Pow(x, 3)", 1, [](const auto& x) { return Utils::Pow(x[0], 3); }, [](const auto& x) { return Utils::Pow(x[0], 3); },
                      [](const auto& x) { return std::vector<double>{3.0 * x[0] * x[0]}; }, {{-1.5}, {0.0}, {2.0}});
    CheckScalarHelper("Pow(x, 2.5)", 1, [](const auto& x) { return Utils::Pow(x[0], 2.5); }, [](const auto& x) { return Utils::Pow(x[0], 2.5); },
                      [](const auto& x) { return std::vector<double>{2.5 * std::pow(x[0], 1.5)}; }, {{0.5}, {2.0}});
    CheckScalarHelper("Sqrt", 1, [](const auto& x) { return Utils::Sqrt(x[0]); }, [](const auto& x) { return Utils::Sqrt(x[0]); },
                      [](const auto& x) { return std::vector<double>{0.5 / std::sqrt(x[0])}; }, {{0.25}, {9.0}});
    CheckScalarHelper("ApproximateNorm", 3, [](const auto& x) { return Utils::ApproximateNorm(Vector3ad{x[0], x[1], x[2]}); },
                      [](const auto& x) { return Utils::ApproximateNorm(Vector3r{x[0], x[1], x[2]}); },
                      [eps](const auto& x) {
                          const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + eps);
                          return std::vector<double>{x[0] / n, x[1] / n, x[2] / n};
                      },
                      {{0.0, 0.0, 0.0}, {1.0, -2.0, 0.5}});
}

static std::vector<ad_scalar_t> Coeffs(const Quaternionad& q) {
    return {q.x(), q.y(), q.z(), q.w()};
}

static void TestQuaternionLayer() {
    std::mt19937 gen{11U};
    std::normal_distribution<double> nrm;
    auto randomUnit = [&] {
        Quaternionr q{nrm(gen), nrm(gen), nrm(gen), nrm(gen)};
        return q.normalized();
    };
    // ---- slerp: recorded vs host branches vs the textbook formula, derivative w.r.t. t vs central differences -------------
    for (int mode : {1, 2}) {
        const Recorded r = Record(9, [](const std::vector<ad_scalar_t>& x) {
            const Quaternionad a{x[3], x[0], x[1], x[2]}, b{x[7], x[4], x[5], x[6]};
            return Coeffs(a.slerp(x[8], b));
        }, mode);
        for (int s = 0; s < 40; ++s) {
            const Quaternionr a = randomUnit();
            Quaternionr b = randomUnit();
            if (s % 5 == 3) b = a;                                                   // identical: the linear branch (|d| >= 1 - eps)
            if (s % 5 == 4) b = Quaternionr{-a.w(), -a.x(), -a.y(), -a.z()};         // antipodal: linear branch with d < 0
            const double t = 0.05 + 0.9 * (s % 7) / 6.0;
            const std::vector<double> in{a.x(), a.y(), a.z(), a.w(), b.x(), b.y(), b.z(), b.w(), t};
            std::vector<double> y, jac;
            r.Eval(in, y, jac);
            const Quaternionr host = a.slerp(t, b);
            const double d = a.dot(b), ad = std::fabs(d), th = std::acos(std::min(ad, 1.0));
            const bool linear = ad >= 1.0 - std::numeric_limits<double>::epsilon();
            const double s0 = linear ? 1.0 - t : std::sin((1.0 - t) * th) / std::sin(th);
            double s1 = linear ? t : std::sin(t * th) / std::sin(th);
            if (d < 0) s1 = -s1;
            const double want[4] = {s0 * a.x() + s1 * b.x(), s0 * a.y() + s1 * b.y(), s0 * a.z() + s1 * b.z(), s0 * a.w() + s1 * b.w()};
            const double hostc[4] = {host.x(), host.y(), host.z(), host.w()};
            for (int i = 0; i < 4; ++i) {
                EXPECT_TRUE(Close(y[static_cast<std::size_t>(i)], want[i], 1e-12));
                EXPECT_TRUE(Close(y[static_cast<std::size_t>(i)], hostc[i], 1e-15));  // same expression, recorded or not
            }
            for (double v : jac) EXPECT_TRUE(std::isfinite(v));  // also on the guarded branch (sin(acos(1)) = 0 in the other one)
            if (!linear) {
                const double h = 1e-6;
                const Quaternionr p = a.slerp(t + h, b), m = a.slerp(t - h, b);
                const double fd[4] = {(p.x() - m.x()) / (2 * h), (p.y() - m.y()) / (2 * h), (p.z() - m.z()) / (2 * h), (p.w() - m.w()) / (2 * h)};
                for (int i = 0; i < 4; ++i) EXPECT_TRUE(std::fabs(jac[static_cast<std::size_t>(i) * 9 + 8] - fd[i]) < 1e-7);
            } else {
                for (int i = 0; i < 4; ++i) EXPECT_TRUE(Close(jac[static_cast<std::size_t>(i) * 9 + 8], (d < 0 ? -1.0 : 1.0) * (&b.x())[i] - (&a.x())[i], 1e-12));
            }
        }
    }
    // ---- inverse: q * q^-1 = 1 for any non-null q; the null quaternion maps to itself with finite derivatives ---------------
    for (int mode : {1, 2}) {
        const Recorded r = Record(4, [](const std::vector<ad_scalar_t>& x) {
            const Quaternionad q{x[3], x[0], x[1], x[2]};
            return Coeffs(q.inverse());
        }, mode);
        for (int s = 0; s < 10; ++s) {
            const Quaternionr q{nrm(gen), nrm(gen), nrm(gen), nrm(gen)};
            std::vector<double> y, jac;
            r.Eval({q.x(), q.y(), q.z(), q.w()}, y, jac);
            const Quaternionr inv{y[3], y[0], y[1], y[2]}, prod = q * inv, host = q.inverse();
            EXPECT_TRUE(Close(prod.w(), 1.0) && std::fabs(prod.x()) < 1e-14 && std::fabs(prod.y()) < 1e-14 && std::fabs(prod.z()) < 1e-14);
            EXPECT_TRUE(Close(y[0], host.x(), 1e-15) && Close(y[3], host.w(), 1e-15));
            const double n2 = q.squaredNorm();  // d(inv.w)/d(w) = 1/n2 - 2 w^2 / n2^2
            EXPECT_TRUE(Close(jac[3 * 4 + 3], 1.0 / n2 - 2.0 * q.w() * q.w() / (n2 * n2), 1e-11));
        }
        std::vector<double> y, jac;
        r.Eval({0.0, 0.0, 0.0, 0.0}, y, jac);
        for (double v : y) EXPECT_TRUE(v == 0.0);
        for (double v : jac) EXPECT_TRUE(std::isfinite(v));
        EXPECT_TRUE(Close(jac[0], -1.0) && Close(jac[3 * 4 + 3], 1.0));  // conj / 1 on the guarded branch
    }
    // ---- normalized / normalize: unit result, null vector untouched, finite derivatives there ------------------------------
    for (int mode : {1, 2}) {
        const Recorded r3 = Record(3, [](const std::vector<ad_scalar_t>& x) {
            const Vector3ad v{x[0], x[1], x[2]};
            const Vector3ad n = v.normalized();
            return std::vector<ad_scalar_t>{n[0], n[1], n[2]};
        }, mode);
        const Recorded r4 = Record(4, [](const std::vector<ad_scalar_t>& x) {
            Quaternionad q{x[3], x[0], x[1], x[2]};
            q.normalize();
            Vector4ad c{q.x(), q.y(), q.z(), q.w()};
            c.normalize();  // idempotent
            return std::vector<ad_scalar_t>{c[0], c[1], c[2], c[3]};
        }, mode);
        std::vector<double> y, jac;
        r3.Eval({3.0, 0.0, -4.0}, y, jac);
        EXPECT_TRUE(Close(y[0], 0.6) && Close(y[2], -0.8) && y[1] == 0.0);
        EXPECT_TRUE(Close(jac[0], (1.0 - 0.36) / 5.0) && Close(jac[2], 0.6 * 0.8 / 5.0));  // (I - n n^T) / |v|
        r3.Eval({0.0, 0.0, 0.0}, y, jac);
        for (double v : y) EXPECT_TRUE(v == 0.0);
        for (double v : jac) EXPECT_TRUE(std::isfinite(v));
        r4.Eval({1.0, -2.0, 2.0, 4.0}, y, jac);
        EXPECT_TRUE(Close(y[0], 0.2) && Close(y[1], -0.4) && Close(y[2], 0.4) && Close(y[3], 0.8));
        r4.Eval({0.0, 0.0, 0.0, 0.0}, y, jac);
        for (double v : y) EXPECT_TRUE(v == 0.0);
        for (double v : jac) EXPECT_TRUE(std::isfinite(v));
    }
    // ---- real-scalar operations the reference forbids on recorded scalars still work on real_t --------------------------------
    {
        Quaternionr q;
        q.setFromTwoVectors(Vector3r{1.0, 0.0, 0.0}, Vector3r{0.0, 1.0, 0.0});
        const Vector3r e = q * Vector3r{1.0, 0.0, 0.0};
        EXPECT_TRUE(std::fabs(e[0]) < 1e-15 && Close(e[1], 1.0) && std::fabs(e[2]) < 1e-15);
        q.setFromTwoVectors(Vector3r{0.0, 0.0, 2.0}, Vector3r{0.0, 0.0, -1.0});  // opposite vectors
        const Vector3r f = q * Vector3r{0.0, 0.0, 1.0};
        EXPECT_TRUE(Close(f[2], -1.0) && Close(q.norm(), 1.0));
        const Quaternionr a = randomUnit();
#if defined(UNGAR_AMD_USE_SYSTEM_EIGEN)
        Eigen::Matrix3d R;
#else
        MatrixXr R(3, 3);
#endif
        for (int c = 0; c < 3; ++c) {
            const Vector3r col = a * Vector3r::Unit(c);
            for (int rr = 0; rr < 3; ++rr) R(rr, c) = col[rr];
        }
        Quaternionr back;
        back = R;
        const double sgn = back.dot(a) < 0 ? -1.0 : 1.0;
        EXPECT_TRUE(Close(sgn * back.x(), a.x(), 1e-13) && Close(sgn * back.w(), a.w(), 1e-13));
    }
}

int main() {
    TestScalarHelpers();
    TestQuaternionLayer();
    std::printf(g_failures == 0 ? "helpers_test OK\n" : "helpers_test FAILED (%d)\n", g_failures);
    return g_failures == 0 ? 0 : 1;
}
