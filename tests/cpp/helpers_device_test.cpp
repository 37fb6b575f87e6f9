// GPU test of the scalar / quaternion helpers that model lambdas call while a tape is recorded:
// each helper goes through the PRODUCT pipeline -- Autodiff::MakeFunction -> record -> derive -> HIP emission -> hipcc -> device --
// and its emitted kernels are evaluated on either side of every switching point:
//   value     against the same helper on real_t (host expression),
//   Jacobian  against the closed form (or central differences of the host expression where the closed form is the expression itself),
//   Hessian   against the closed form / central differences of the device Jacobian (scalar-valued cases).
// Run once per accumulation mode of the Jacobian (UNGAR_AMD_JACOBIAN_MODE = 1 forward, 2 reverse; tests/test_tape_host.py drives it).
// Reference: include/ungar/utils/utils.hpp:969-1021 (Min, SmoothMin, Sign, Abs, SmoothAbs), :820-852 (Pow, Sqrt), :731-749;
// include/ungar/autodiff/support/quaternion.hpp:34-192 (inverse, normalize, slerp); optimization/soft_inequality_constraint.hpp:77-205.
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "ungar/autodiff/function.hpp"
#include "ungar/optimization/soft_inequality_constraint.hpp"
#include "ungar/utils/utils.hpp"

using namespace Ungar;

static int g_failures = 0, g_checks = 0;
static std::string g_folder;

static void Expect(bool ok, const std::string& what) {
    ++g_checks;
    if (!ok) {
        ++g_failures;
        std::printf("FAIL %s\n", what.c_str());
    }
}
static bool Close(double a, double b, double tol) {
    return std::isfinite(a) && std::isfinite(b) && std::fabs(a - b) <= tol * (1.0 + std::fabs(b));
}

using Real = std::function<std::vector<double>(const std::vector<double>&)>;

struct Case {
    std::string name;
    index_t n;
    ADFunction record;               // the helper on recorded scalars
    Real host;                       // the same helper on real_t
    Real jacobian;                   // closed-form dense Jacobian (row-major m x n), empty: central differences of `host`
    Real hessian;                    // closed-form dense Hessian (n x n) of output 0, empty: central differences of the device Jacobian
    std::vector<std::vector<double>> points;
    bool smooth = true;              // false: a switching point is among `points`; finite differences are skipped there
    double fdTolerance = 1e-6;
};

static VectorXr ToVector(const std::vector<double>& v) {
    VectorXr x{static_cast<index_t>(v.size())};
    for (std::size_t i = 0; i < v.size(); ++i) x[static_cast<index_t>(i)] = v[i];
    return x;
}

static void Run(const Case& c) {
    VectorXad probe;
    c.record(VectorXad::Random(c.n), probe);
    const index_t m = probe.size();
    const bool scalar = m == 1;
    Autodiff::Function f = Autodiff::MakeFunction({c.record, c.n, index_t{0}, "a10_" + c.name, scalar ? EnabledDerivatives::ALL : EnabledDerivatives::JACOBIAN, g_folder}, true);
    auto deviceJacobian = [&](const std::vector<double>& at) { return Linalg::ToDense(f.Jacobian(ToVector(at))); };
    for (const std::vector<double>& at : c.points) {
        const std::string where = c.name + " at (" + std::to_string(at[0]) + (c.n > 1 ? ", ...)" : ")");
        const VectorXr y = f(ToVector(at));
        const std::vector<double> want = c.host(at);
        for (index_t i = 0; i < m; ++i) Expect(Close(y[i], want[static_cast<std::size_t>(i)], 1e-13), where + ": value " + std::to_string(i));
        const std::vector<real_t> J = deviceJacobian(at);
        if (c.jacobian) {
            const std::vector<double> Jw = c.jacobian(at);
            for (std::size_t k = 0; k < Jw.size(); ++k) Expect(Close(J[k], Jw[k], 1e-11), where + ": Jacobian entry " + std::to_string(k) + " = " + std::to_string(J[k]) + " vs " + std::to_string(Jw[k]));
        } else if (c.smooth) {
            for (index_t j = 0; j < c.n; ++j) {
                std::vector<double> p = at, q = at;
                const double h = 1e-6;
                p[static_cast<std::size_t>(j)] += h;
                q[static_cast<std::size_t>(j)] -= h;
                const std::vector<double> yp = c.host(p), ym = c.host(q);
                for (index_t i = 0; i < m; ++i)
                    Expect(std::fabs(J[static_cast<std::size_t>(i * c.n + j)] - (yp[static_cast<std::size_t>(i)] - ym[static_cast<std::size_t>(i)]) / (2 * h)) <= c.fdTolerance * (1.0 + std::fabs(J[static_cast<std::size_t>(i * c.n + j)])),
                           where + ": Jacobian vs central differences (" + std::to_string(i) + ", " + std::to_string(j) + ")");
            }
        }
        for (const real_t v : J) Expect(std::isfinite(v), where + ": finite Jacobian");
        if (!scalar) continue;
        const std::vector<real_t> H = Linalg::ToDense(f.Hessian(ToVector(at)));  // upper triangle
        for (const real_t v : H) Expect(std::isfinite(v), where + ": finite Hessian");
        if (c.hessian) {
            const std::vector<double> Hw = c.hessian(at);
            for (index_t r = 0; r < c.n; ++r)
                for (index_t s = r; s < c.n; ++s)
                    Expect(Close(H[static_cast<std::size_t>(r * c.n + s)], Hw[static_cast<std::size_t>(r * c.n + s)], 1e-10), where + ": Hessian (" + std::to_string(r) + ", " + std::to_string(s) + ")");
        } else if (c.smooth) {
            for (index_t s = 0; s < c.n; ++s) {
                std::vector<double> p = at, q = at;
                const double h = 1e-6;
                p[static_cast<std::size_t>(s)] += h;
                q[static_cast<std::size_t>(s)] -= h;
                const std::vector<real_t> Jp = deviceJacobian(p), Jm = deviceJacobian(q);
                for (index_t r = 0; r <= s; ++r)
                    Expect(std::fabs(H[static_cast<std::size_t>(r * c.n + s)] - (Jp[static_cast<std::size_t>(r)] - Jm[static_cast<std::size_t>(r)]) / (2 * h)) <= c.fdTolerance * (1.0 + std::fabs(H[static_cast<std::size_t>(r * c.n + s)])),
                           where + ": Hessian vs differences of the device Jacobian (" + std::to_string(r) + ", " + std::to_string(s) + ")");
            }
        }
    }
    std::printf("%-28s m=%td n=%td  %zu points\n", c.name.c_str(), m, c.n, c.points.size());
}

template <class S>
static Quaternion<S> Quat(const S& x, const S& y, const S& z, const S& w) {
    return Quaternion<S>{w, x, y, z};
}

int main(int argc, char** argv) {
    g_folder = argc > 1 ? argv[1] : "/tmp/ungar_amd_a10";
    const double eps = std::numeric_limits<double>::epsilon();
    try {
        std::vector<Case> cases;
        // ---- utils.hpp:969-1021
        cases.push_back({"sign_x_squared", 1, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::Sign(x[0]) * x[0] * x[0]; },
                         [](const auto& x) { return std::vector<double>{Utils::Sign(x[0]) * x[0] * x[0]}; },
                         [](const auto& x) { return std::vector<double>{Utils::Sign(x[0]) * 2.0 * x[0]}; },
                         [](const auto& x) { return std::vector<double>{Utils::Sign(x[0]) * 2.0}; }, {{-1.5}, {-1e-3}, {0.0}, {1e-3}, {2.5}}, false});
        cases.push_back({"min", 2, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::Min(x[0], x[1]) * x[0]; },
                         [](const auto& x) { return std::vector<double>{Utils::Min(x[0], x[1]) * x[0]}; },
                         [](const auto& x) { return x[0] > x[1] ? std::vector<double>{x[1], x[0]} : std::vector<double>{2.0 * x[0], 0.0}; },
                         [](const auto& x) { return x[0] > x[1] ? std::vector<double>{0.0, 1.0, 1.0, 0.0} : std::vector<double>{2.0, 0.0, 0.0, 0.0}; },
                         {{-1.0, 2.0}, {2.0, -1.0}, {0.5, 0.5}, {-3.0, -3.5}}, false});
        cases.push_back({"abs", 1, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::Abs(x[0]) * x[0]; },
                         [](const auto& x) { return std::vector<double>{Utils::Abs(x[0]) * x[0]}; },
                         [](const auto& x) { return std::vector<double>{2.0 * std::fabs(x[0])}; },
                         [](const auto& x) { return std::vector<double>{2.0 * (static_cast<double>(x[0] > 0) - static_cast<double>(x[0] < 0))}; }, {{-1.5}, {0.0}, {2.5}}, false});
        cases.push_back({"smooth_abs", 1, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::SmoothAbs(x[0]); },
                         [](const auto& x) { return std::vector<double>{Utils::SmoothAbs(x[0])}; },
                         [eps](const auto& x) { return std::vector<double>{x[0] / std::sqrt(x[0] * x[0] + eps)}; },
                         [eps](const auto& x) { return std::vector<double>{eps / std::pow(x[0] * x[0] + eps, 1.5)}; }, {{-1.5}, {-1e-9}, {0.0}, {1e-9}, {2.5}}});
        cases.push_back({"smooth_abs_wide", 1, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::SmoothAbs(x[0], ad_scalar_t{1e-2}); },
                         [](const auto& x) { return std::vector<double>{Utils::SmoothAbs(x[0], 1e-2)}; },
                         [](const auto& x) { return std::vector<double>{x[0] / std::sqrt(x[0] * x[0] + 1e-2)}; },
                         [](const auto& x) { return std::vector<double>{1e-2 / std::pow(x[0] * x[0] + 1e-2, 1.5)}; }, {{-0.2}, {0.0}, {0.7}}});
        auto smoothMinGradient = [](double alpha) {
            return [alpha](const std::vector<double>& x) {
                const double ea = std::exp(-alpha * x[0]), eb = std::exp(-alpha * x[1]), D = ea + eb, S = (x[0] * ea + x[1] * eb) / D;
                return std::vector<double>{ea / D * (1.0 - alpha * x[0] + alpha * S), eb / D * (1.0 - alpha * x[1] + alpha * S)};
            };
        };
        cases.push_back({"smooth_min", 2, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::SmoothMin(x[0], x[1]); },
                         [](const auto& x) { return std::vector<double>{Utils::SmoothMin(x[0], x[1])}; }, smoothMinGradient(8.0), nullptr,
                         {{-1.0, 2.0}, {2.0, -1.0}, {0.5, 0.5}, {0.1, 0.12}}});
        cases.push_back({"smooth_min_alpha", 2, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::SmoothMin(x[0], x[1], ad_scalar_t{2.5}); },
                         [](const auto& x) { return std::vector<double>{Utils::SmoothMin(x[0], x[1], 2.5)}; }, smoothMinGradient(2.5), nullptr, {{-1.0, 2.0}, {0.3, 0.25}}});
        cases.push_back({"pow_real_exponent", 2, [](const VectorXad& x, VectorXad& y) { y.resize(1); y << Utils::Pow(x[0], x[1]); },  // exp / log in the emitted code
                         [](const auto& x) { return std::vector<double>{std::pow(x[0], x[1])}; },
                         [](const auto& x) { return std::vector<double>{x[1] * std::pow(x[0], x[1] - 1.0), std::log(x[0]) * std::pow(x[0], x[1])}; },
                         [](const auto& x) {
                             const double p = std::pow(x[0], x[1]), l = std::log(x[0]);
                             return std::vector<double>{x[1] * (x[1] - 1.0) * p / (x[0] * x[0]), p / x[0] * (1.0 + x[1] * l), 0.0, l * l * p};
                         },
                         {{0.5, 2.5}, {2.0, -1.5}, {1.0, 3.0}}});
        // ---- relaxed barriers on either side of their switching points (soft_inequality_constraint.hpp:77-205)
        for (const bool log : {true, false}) {
            const double k = 3.0, e = 0.25;
            const RelaxedLogBarrierFunction lb{0.0, k, e};
            const RelaxedPolyBarrierFunction pb{0.0, k, e};
            cases.push_back({log ? "relaxed_log_barrier" : "relaxed_poly_barrier", 1,
                             [log, k, e](const VectorXad& x, VectorXad& y) {
                                 y.resize(1);
                                 if (log) y << RelaxedLogBarrierFunction{0.0, k, e}.Evaluate(ad_scalar_t{x[0]});
                                 else y << RelaxedPolyBarrierFunction{0.0, k, e}.Evaluate(ad_scalar_t{x[0]});
                             },
                             [log, lb, pb](const auto& x) { return std::vector<double>{log ? lb.Evaluate(x[0]) : pb.Evaluate(x[0])}; },
                             [log, lb, pb](const auto& x) { return std::vector<double>{log ? lb.FirstDerivative(x[0]) : pb.FirstDerivative(x[0])}; },
                             [log, lb, pb](const auto& x) { return std::vector<double>{log ? lb.SecondDerivative(x[0]) : pb.SecondDerivative(x[0])}; },
                             {{-0.4}, {-1e-6}, {0.0}, {1e-6}, {0.1}, {0.25 - 1e-9}, {0.25}, {0.25 + 1e-9}, {0.9}, {4.0}}, false});
        }
        // ---- quaternion layer (autodiff/support/quaternion.hpp:34-192): acos / sin in the emitted code, guarded branches
        cases.push_back({"quaternion_inverse", 4,
                         [](const VectorXad& x, VectorXad& y) {
                             const Quaternionad q = Quat<ad_scalar_t>(x[0], x[1], x[2], x[3]).inverse();
                             y.resize(4);
                             y << q.x(), q.y(), q.z(), q.w();
                         },
                         [](const auto& x) {
                             const Quaternionr q = Quat<real_t>(x[0], x[1], x[2], x[3]).inverse();
                             return std::vector<double>{q.x(), q.y(), q.z(), q.w()};
                         },
                         nullptr, nullptr, {{0.3, -0.4, 0.5, 0.7}, {1.0, 2.0, -2.0, 0.5}, {0.0, 0.0, 0.0, 0.0}}, true, 1e-6});
        cases.back().smooth = false;  // the null quaternion is the guarded point: values and finiteness only there ...
        cases.push_back({"quaternion_inverse_smooth", 4, cases.back().record, cases.back().host, nullptr, nullptr, {{0.3, -0.4, 0.5, 0.7}, {1.0, 2.0, -2.0, 0.5}}});  // ... derivatives here
        cases.push_back({"vector_normalized", 3,
                         [](const VectorXad& x, VectorXad& y) {
                             const Vector3ad n = Vector3ad{x[0], x[1], x[2]}.normalized();
                             y.resize(3);
                             y << n[0], n[1], n[2];
                         },
                         [](const auto& x) {
                             const Vector3r n = Vector3r{x[0], x[1], x[2]}.normalized();
                             return std::vector<double>{n[0], n[1], n[2]};
                         },
                         [](const auto& x) {  // (I - n n^T) / |v|
                             const double l = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
                             std::vector<double> J(9);
                             for (int r = 0; r < 3; ++r)
                                 for (int s = 0; s < 3; ++s) J[static_cast<std::size_t>(3 * r + s)] = ((r == s) - x[static_cast<std::size_t>(r)] * x[static_cast<std::size_t>(s)] / (l * l)) / l;
                             return J;
                         },
                         nullptr, {{3.0, 0.0, -4.0}, {0.1, 0.2, -0.3}}});
        cases.push_back({"vector_normalized_null", 3, cases.back().record, cases.back().host, nullptr, nullptr, {{0.0, 0.0, 0.0}}, false});
        const ADFunction slerp = [](const VectorXad& x, VectorXad& y) {
            const Quaternionad a = Quat<ad_scalar_t>(x[0], x[1], x[2], x[3]), b = Quat<ad_scalar_t>(x[4], x[5], x[6], x[7]);
            const Quaternionad s = a.slerp(x[8], b);
            y.resize(4);
            y << s.x(), s.y(), s.z(), s.w();
        };
        const Real slerpHost = [](const std::vector<double>& x) {
            const Quaternionr s = Quat<real_t>(x[0], x[1], x[2], x[3]).slerp(x[8], Quat<real_t>(x[4], x[5], x[6], x[7]));
            return std::vector<double>{s.x(), s.y(), s.z(), s.w()};
        };
        const Quaternionr a = Quaternionr{0.7, 0.3, -0.4, 0.5}.normalized(), b = Quaternionr{0.2, -0.6, 0.1, 0.75}.normalized();
        auto pair = [](const Quaternionr& p, const Quaternionr& q, double t) { return std::vector<double>{p.x(), p.y(), p.z(), p.w(), q.x(), q.y(), q.z(), q.w(), t}; };
        cases.push_back({"slerp", 9, slerp, slerpHost, nullptr, nullptr, {pair(a, b, 0.3), pair(b, a, 0.85), pair(a, Quaternionr{-b.w(), -b.x(), -b.y(), -b.z()}, 0.4)}, true, 2e-6});
        // identical / antipodal arguments take the linear branch (|d| >= 1 - eps): sin(acos(1)) = 0 sits in the branch NOT taken
        cases.push_back({"slerp_guarded", 9, slerp, slerpHost, nullptr, nullptr, {pair(a, a, 0.3), pair(a, Quaternionr{-a.w(), -a.x(), -a.y(), -a.z()}, 0.6)}, false});
        cases.push_back({"slerp_weighted_scalar", 9,  // a scalar function of the slerp, so that the Hessian kernel runs over acos / sin as well
                         [slerp](const VectorXad& x, VectorXad& y) {
                             VectorXad s;
                             slerp(x, s);
                             y.resize(1);
                             y << 0.3 * s[0] - 1.1 * s[1] * s[2] + 0.7 * s[3] * s[3];
                         },
                         [slerpHost](const std::vector<double>& x) {
                             const std::vector<double> s = slerpHost(x);
                             return std::vector<double>{0.3 * s[0] - 1.1 * s[1] * s[2] + 0.7 * s[3] * s[3]};
                         },
                         nullptr, nullptr, {pair(a, b, 0.3), pair(b, a, 0.85)}, true, 5e-6});
        for (const Case& c : cases) Run(c);
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(g_failures == 0 ? "helpers_device_test OK (%d checks)\n" : "helpers_device_test FAILED (%d of %d checks)\n", g_failures == 0 ? g_checks : g_failures, g_checks);
    return g_failures == 0 ? 0 : 1;
}
