// GPU test of the C++ host facade (run by tests/test_cpp_facade.py on the MI355X box).
// Mirrors the reference's test/autodiff/function.test.cpp (ExponentialMap :33-59, Jacobian :61-109,
// Hessian :111-142: closed-form ground truths at :81-89 and :131) and then records the quadrotor
// shooting-node function through the variable-map API exactly the way a user model lambda does,
// printing value and dense Jacobian for the Python side to compare with the oracle.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "ungar/autodiff/function.hpp"
#include "ungar/autodiff/vector_composer.hpp"
#include "ungar/variable_map.hpp"

using namespace Ungar;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                                 \
    do {                                                                  \
        if (!(cond)) {                                                    \
            ++g_failures;                                                 \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
        }                                                                 \
    } while (0)

static VectorXr RandomVector(index_t n, std::mt19937& gen) {
    std::uniform_real_distribution<real_t> d{-1.0, 1.0};
    VectorXr v{n};
    for (index_t i = 0; i < n; ++i) v[i] = d(gen);
    return v;
}

static void TestExponentialMap(const std::string& folder) {
    const auto impl = [](const VectorXad& xp, VectorXad& y) {
        const Vector3ad v{xp[0], xp[1], xp[2]};
        y = Utils::ApproximateExponentialMap(v).coeffs();
    };
    Autodiff::Function::Blueprint bp{impl, 3, 0, "function_test_exponential_map", EnabledDerivatives::JACOBIAN, folder};
    Autodiff::Function f = Autodiff::MakeFunction(bp, true);
    EXPECT_TRUE(f.DependentVariableSize() == 4 && f.IndependentVariableSize() == 3 && f.ParameterSize() == 0);
    std::mt19937 gen{0U};
    for (int k = 0; k < 64; ++k) {
        const VectorXr x = RandomVector(3, gen);
        const Quaternionr exact = Utils::ExponentialMap(x);
        const VectorXr y = f(x);
        for (index_t i = 0; i < 4; ++i) EXPECT_TRUE(std::fabs(y[i] - exact.coeffs()[i]) < 1e-7);
        EXPECT_TRUE(f.TestJacobian(x));
    }
    EXPECT_TRUE(f.TestJacobian(VectorXr::Zero(3)));  // function.test.cpp:55-58
    std::printf("exponential_map cache_hit=%d\n", f.LoadedFromCache() ? 1 : 0);
    // second Make without recompile must reuse the hashed code object
    Autodiff::Function g = Autodiff::MakeFunction(bp, false);
    EXPECT_TRUE(g.LoadedFromCache());
}

static void TestJacobianClosedForm(const std::string& folder) {
    // y = [p |x|^2, 2 x0^2]   (function.test.cpp:70-79)
    const auto impl = [](const VectorXad& xp, VectorXad& y) {
        y.resize(2);
        const ad_scalar_t p = xp[4];
        y[0] = p * (xp[0] * xp[0] + xp[1] * xp[1] + xp[2] * xp[2] + xp[3] * xp[3]);
        y[1] = 2.0 * Utils::Pow(xp[0], 2);
    };
    Autodiff::Function::Blueprint bp{impl, 4, 1, "function_test_jacobian", EnabledDerivatives::JACOBIAN, folder};
    Autodiff::Function f = Autodiff::MakeFunction(bp, true);
    EXPECT_TRUE(f.ImplementsJacobian() && !f.ImplementsHessian());
    std::mt19937 gen{1U};
    for (int k = 0; k < 64; ++k) {
        const VectorXr xp = RandomVector(5, gen);
        const real_t p = xp[4];
        const VectorXr y = f(xp);
        const real_t n2 = xp[0] * xp[0] + xp[1] * xp[1] + xp[2] * xp[2] + xp[3] * xp[3];
        EXPECT_TRUE(std::fabs(y[0] - p * n2) < 1e-12 && std::fabs(y[1] - 2 * xp[0] * xp[0]) < 1e-12);
        const auto& J = f.Jacobian(xp);
        EXPECT_TRUE(J.rows() == 2 && J.cols() == 4);  // m x n: parameter column trimmed
        for (index_t j = 0; j < 4; ++j) EXPECT_TRUE(std::fabs(J.coeff(0, j) - 2 * p * xp[j]) < 1e-12);
        EXPECT_TRUE(std::fabs(J.coeff(1, 0) - 4 * xp[0]) < 1e-12);
        for (index_t j = 1; j < 4; ++j) EXPECT_TRUE(J.coeff(1, j) == 0.0);
        EXPECT_TRUE(J.nonZeros() == 5);
        EXPECT_TRUE(f.TestJacobian(xp));
    }
    bool threw = false;
    try {
        (void)f.Hessian(VectorXr::Zero(5));
    } catch (const std::logic_error&) {
        threw = true;
    }
    EXPECT_TRUE(threw);
}

static void TestHessianClosedForm(const std::string& folder) {
    // y = p |x|^2, Hessian = 2 p I (upper triangle)   (function.test.cpp:120-131)
    const auto impl = [](const VectorXad& xp, VectorXad& y) {
        y.resize(1);
        y[0] = xp[4] * (xp[0] * xp[0] + xp[1] * xp[1] + xp[2] * xp[2] + xp[3] * xp[3]);
    };
    Autodiff::Function::Blueprint bp{impl, 4, 1, "function_test_hessian", EnabledDerivatives::ALL, folder};
    Autodiff::Function f = Autodiff::MakeFunction(bp, true);
    std::mt19937 gen{2U};
    for (int k = 0; k < 64; ++k) {
        const VectorXr xp = RandomVector(5, gen);
        const auto& H = f.Hessian(xp);
        EXPECT_TRUE(H.rows() == 4 && H.cols() == 4 && H.nonZeros() == 4);
        for (index_t i = 0; i < 4; ++i)
            for (index_t j = 0; j < 4; ++j) EXPECT_TRUE(std::fabs(H.coeff(i, j) - (i == j ? 2 * xp[4] : 0.0)) < 1e-12);
        EXPECT_TRUE(f.TestHessian(xp));
    }
}

// ---- quadrotor shooting node through the variable-map API ----------------------------------------
namespace quadrotor {
constexpr auto ROTORS = 4_c;
UNGAR_VARIABLE(position, 3);
UNGAR_VARIABLE(orientation, Q);
UNGAR_VARIABLE(linear_velocity, 3);
UNGAR_VARIABLE(b_angular_velocity, 3);
UNGAR_VARIABLE(x) <<= (position, orientation, linear_velocity, b_angular_velocity);
UNGAR_VARIABLE(rotor_speed, 1);
UNGAR_VARIABLE(u) <<= ROTORS * rotor_speed;
UNGAR_VARIABLE(step_size, 1);
UNGAR_VARIABLE(mass, 1);
UNGAR_VARIABLE(b_moi_diagonal, 3);
UNGAR_VARIABLE(b_propeller_position, 3);
UNGAR_VARIABLE(standard_gravity, 1);
UNGAR_VARIABLE(thrust_constant, 1);
UNGAR_VARIABLE(drag_constant, 1);
UNGAR_VARIABLE(parameters) <<= (step_size, mass, b_moi_diagonal, ROTORS * b_propeller_position, standard_gravity, thrust_constant, drag_constant);
UNGAR_VARIABLE(xup) <<= (x, u, parameters);
}  // namespace quadrotor

static void QuadrotorNodeThroughFacade(const std::string& folder) {
    using namespace quadrotor;
    const auto dynamics = [&](const VectorXad& xUnderlying, const VectorXad& uUnderlying, const VectorXad& parUnderlying) -> VectorXad {
        const auto x_ = MakeVariableLazyMap(xUnderlying, x);
        const auto u_ = MakeVariableLazyMap(uUnderlying, u);
        const auto par_ = MakeVariableLazyMap(parUnderlying, parameters);
        const auto [dt, g0, b, d] = par_.GetTuple(step_size, standard_gravity, thrust_constant, drag_constant);
        const auto [m, moi] = par_.GetTuple(mass, b_moi_diagonal);
        const auto [p, q, pDot, bOmega] = x_.GetTuple(position, orientation, linear_velocity, b_angular_velocity);
        Vector3ad force = Vector3ad::Zero(), thrustMoment = Vector3ad::Zero(), dragMoment = Vector3ad::Zero();
        for (const auto i : enumerate(ROTORS)) {
            const auto& r = u_.Get(rotor_speed, i);
            const auto arm = par_.Get(b_propeller_position, i);
            const Vector3ad thrust = b * Utils::Pow(r, 2) * Vector3ad::UnitZ();
            force += thrust;
            thrustMoment += arm.cross(thrust);
            dragMoment += d * Utils::Pow(r, 2) * Vector3ad::UnitZ() * Utils::Pow(-1.0, i);
        }
        const Vector3ad pDotDot = (q * force - m * g0 * Vector3ad::UnitZ()) / m;
        const Vector3ad bOmegaDot = moi.cwiseInverse().cwiseProduct(thrustMoment + dragMoment - bOmega.cross(moi.cwiseProduct(bOmega)));
        auto xNext_ = MakeVariableMap<ad_scalar_t>(x);
        auto [pNext, qNext, pDotNext, bOmegaNext] = xNext_.GetTuple(position, orientation, linear_velocity, b_angular_velocity);
        pDotNext = pDot + dt * pDotDot;
        bOmegaNext = bOmega + dt * bOmegaDot;
        pNext = p + dt * pDotNext;
        qNext = q * Utils::ApproximateExponentialMap(dt * bOmegaNext);
        return xNext_.Get();
    };
    const auto impl = [&](const VectorXad& v, VectorXad& y) {
        const auto v_ = MakeVariableLazyMap(v, xup);
        y = dynamics(v_.Get(x), v_.Get(u), v_.Get(parameters));
    };
    Autodiff::Function::Blueprint bp{impl, x.Size() + u.Size(), parameters.Size(), "facade_quadrotor_node", EnabledDerivatives::JACOBIAN, folder};
    Autodiff::Function f = Autodiff::MakeFunction(bp, true);
    EXPECT_TRUE(f.DependentVariableSize() == 13 && f.IndependentVariableSize() == 17 && f.ParameterSize() == 20);
    // deterministic input, reference parameter values (quadrotor.example.cpp:326-343)
    auto v_ = MakeVariableMap<real_t>(xup);
    v_.Get(position) = Vector3r(0.3, -0.4, 1.2);
    v_.Get(orientation) = Quaternionr(0.8, 0.2, -0.4, 0.4).normalized();
    v_.Get(linear_velocity) = Vector3r(0.5, 0.1, -0.2);
    v_.Get(b_angular_velocity) = Vector3r(-0.3, 0.6, 0.2);
    v_.Get(u).setLinSpaced(14.0, 17.0);
    v_.Get(step_size) = 1.0 / 30.0;
    v_.Get(mass) = 1.5;
    v_.Get(b_moi_diagonal).setConstant(3e-2);
    v_.Get(b_propeller_position, 0) = Vector3r(0.2, 0.2, 0.0);
    v_.Get(b_propeller_position, 1) = Vector3r(-0.2, 0.2, 0.0);
    v_.Get(b_propeller_position, 2) = Vector3r(-0.2, -0.2, 0.0);
    v_.Get(b_propeller_position, 3) = Vector3r(0.2, -0.2, 0.0);
    v_.Get(standard_gravity) = 9.80665;
    v_.Get(thrust_constant) = 0.015;
    v_.Get(drag_constant) = 0.1;
    const VectorXr& in = v_.Get();
    const VectorXr y = f(in);
    const auto& J = f.Jacobian(in);
    EXPECT_TRUE(J.nonZeros() == 118);  // SURVEY.md §8(a) A6 structural nnz
    EXPECT_TRUE(f.TestJacobian(in));
    const auto real = Utils::ToRealFunction(dynamics);  // A13: the same lambda run on doubles
    const VectorXr yReal = real(VectorXr{in.head(13)}, VectorXr{in.segment(13, 4)}, VectorXr{in.tail(20)});
    for (index_t i = 0; i < 13; ++i) EXPECT_TRUE(std::fabs(yReal[i] - y[i]) < 1e-12);
    {   // PCIe-inclusive latency of the single-instance host path (H2D + batch-1 launch + D2H + sync)
        const auto t0 = std::chrono::steady_clock::now();
        const int reps = 2000;
        real_t acc = 0;
        for (int i = 0; i < reps; ++i) acc += f.Jacobian(in).valuePtr()[0];
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("HOST_CALL_LATENCY_US %.2f (Function::Jacobian, quadrotor node, 37 inputs -> 118 values; checksum %.3g)\n", us, acc);
    }
    std::printf("QUADROTOR_IN");
    for (index_t i = 0; i < in.size(); ++i) std::printf(" %.17g", in[i]);
    std::printf("\nQUADROTOR_F");
    for (index_t i = 0; i < y.size(); ++i) std::printf(" %.17g", y[i]);
    std::printf("\nQUADROTOR_J");
    for (const real_t e : Linalg::ToDense(J)) std::printf(" %.17g", e);
    std::printf("\n");
}

/// a function loaded from a cache hit (derive / emit / compile skipped, sparsity read from the
/// entry's meta file) evaluates exactly like the freshly compiled one; an EDITED function of the same name never picks up
/// the stale entry (the reference's existence-only cache would, function.hpp:420-451); a linear objective has an enabled but
/// structurally empty Hessian that evaluates to an empty matrix instead of failing.
static void TestCacheAndEmptyDerivatives(const std::string& folder) {
    const auto make = [&](real_t gain, bool recompile) {
        const auto impl = [gain](const VectorXad& xp, VectorXad& y) {
            y.resize(1);
            using std::sin;
            y[0] = gain * xp[3] * (xp[0] * xp[0] + xp[1] * xp[1] * xp[2]) + sin(xp[0] * xp[2]);
        };
        return Autodiff::MakeFunction(Autodiff::Function::Blueprint{impl, 3, 1, "function_test_cache_probe", EnabledDerivatives::ALL, folder}, recompile);
    };
    Autodiff::Function fresh = make(1.5, true);
    EXPECT_TRUE(!fresh.LoadedFromCache());
    Autodiff::Function cached = make(1.5, false);
    EXPECT_TRUE(cached.LoadedFromCache());
    Autodiff::Function edited = make(2.5, false);  // same name, different tape
    EXPECT_TRUE(!edited.LoadedFromCache());
    std::mt19937 gen{5U};
    for (int k = 0; k < 16; ++k) {
        const VectorXr xp = RandomVector(4, gen);
        EXPECT_TRUE(fresh(xp)[0] == cached(xp)[0]);
        const std::vector<real_t> j0 = Linalg::ToDense(fresh.Jacobian(xp)), j1 = Linalg::ToDense(cached.Jacobian(xp));
        const std::vector<real_t> h0 = Linalg::ToDense(fresh.Hessian(xp)), h1 = Linalg::ToDense(cached.Hessian(xp));
        EXPECT_TRUE(j0 == j1 && h0 == h1 && j0.size() == 3 && h0.size() == 9);
        EXPECT_TRUE(std::fabs(edited(xp)[0] - (fresh(xp)[0] + 1.0 * xp[3] * (xp[0] * xp[0] + xp[1] * xp[1] * xp[2]))) < 1e-12);
        EXPECT_TRUE(std::fabs(h0[0] - (2.0 * 1.5 * xp[3] - xp[2] * xp[2] * std::sin(xp[0] * xp[2]))) < 1e-12);
        EXPECT_TRUE(cached.TestJacobian(xp) && cached.TestHessian(xp));
    }
    // linear objective: Hessian enabled, structurally empty
    const auto linear = [](const VectorXad& xp, VectorXad& y) {
        y.resize(1);
        y[0] = 2.0 * xp[0] - xp[1] + xp[2] * 0.5;
    };
    Autodiff::Function lin = Autodiff::MakeFunction(Autodiff::Function::Blueprint{linear, 3, 0, "function_test_linear_objective", EnabledDerivatives::ALL, folder}, true);
    const VectorXr x = RandomVector(3, gen);
    EXPECT_TRUE(lin.ImplementsHessian() && lin.Hessian(x).nonZeros() == 0 && lin.Hessian(x).rows() == 3);
    const std::vector<real_t> jl = Linalg::ToDense(lin.Jacobian(x));
    EXPECT_TRUE(jl.size() == 3 && jl[0] == 2.0 && jl[1] == -1.0 && jl[2] == 0.5);
    // a function of the parameters only: Jacobian enabled, structurally empty
    const auto paramOnly = [](const VectorXad& xp, VectorXad& y) {
        y.resize(2);
        y[0] = xp[2] * xp[2];
        y[1] = xp[2] + 1.0;
    };
    Autodiff::Function par = Autodiff::MakeFunction(Autodiff::Function::Blueprint{paramOnly, 2, 1, "function_test_parameters_only", EnabledDerivatives::JACOBIAN, folder}, true);
    VectorXr xp3{3};
    xp3 << 0.1, 0.2, 3.0;
    EXPECT_TRUE(par.Jacobian(xp3).nonZeros() == 0 && par(xp3)[0] == 9.0 && par(xp3)[1] == 4.0);
    // the guard pattern of the reference's AD-safe quaternion layer, differentiated in reverse mode on the device
    const auto guarded = [](const VectorXad& xp, VectorXad& y) {
        y.resize(1);
        using std::sqrt;
        y[0] = xp[0] / ::ungar_amd::tape::CondExpGt(xp[1], ad_scalar_t{0.0}, sqrt(xp[1]), ad_scalar_t{1.0});
    };
    Autodiff::Function grd = Autodiff::MakeFunction(Autodiff::Function::Blueprint{guarded, 2, 0, "function_test_guarded_sqrt", EnabledDerivatives::ALL, folder}, true);
    for (const real_t z : {-2.0, 0.0, 4.0}) {
        VectorXr in{2};
        in << 3.0, z;
        const std::vector<real_t> jg = Linalg::ToDense(grd.Jacobian(in)), hg = Linalg::ToDense(grd.Hessian(in));
        EXPECT_TRUE(std::fabs(jg[0] - (z > 0 ? 0.5 : 1.0)) < 1e-15 && std::fabs(jg[1] - (z > 0 ? -0.5 * 3.0 / 8.0 : 0.0)) < 1e-15);
        for (const real_t e : hg) EXPECT_TRUE(std::isfinite(e));
    }
    std::printf("cache probe ok\n");
}

int main(int argc, char** argv) {
    const std::string folder = argc > 1 ? argv[1] : "/tmp/ungar_amd_cpp_test";
    try {
        if (argc > 2 && std::string(argv[2]) == "latency") {  // bench.py: only the single-instance host call (BASELINE config 1's execution model) is timed
            QuadrotorNodeThroughFacade(folder);
            return g_failures ? 1 : 0;
        }
        TestCacheAndEmptyDerivatives(folder);
        TestExponentialMap(folder);
        TestJacobianClosedForm(folder);
        TestHessianClosedForm(folder);
        QuadrotorNodeThroughFacade(folder);
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(g_failures ? "FAILED %d checks\n" : "ALL PASSED\n", g_failures);
    return g_failures ? 1 : 0;
}
