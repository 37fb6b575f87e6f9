// TEST INFRASTRUCTURE / CPU BASELINE -- genuine-reference timing: builds the REAL
// `Ungar::Autodiff::Function` (CppAD + CppADCodeGen, gcc-JIT'ed C, dlopen) for the quadrotor shooting node written with the
// reference's own headers, and times `Jacobian()` single-threaded, one instance per call -- the reference's execution model.
//
// It can only be compiled where the reference's third-party stack is installed (<cppad/cg.hpp>, Eigen, Boost.Hana); none of
// it is in this image, so oracle/ref_timing/build_reference_timing.sh probes for the header first and bench.py reports
// "genuine_reference": "unavailable: cppad/cg.hpp not found" instead of a number when the probe fails.  Nothing here is part of
// the product; nothing of the reference is copied: the program only INCLUDES the reference's headers from where they are
// installed and restates the quadrotor node lambda of example/mpc/quadrotor.example.cpp:126-190 against them.
//
// usage: time_reference_function <seconds>   -> prints one JSON line {"evals_per_s": ..., "nnz": ..., "calls": ...}
#define UNGAR_CONFIG_ENABLE_AUTODIFF
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "ungar/autodiff/function.hpp"
#include "ungar/variable_map.hpp"

int main(int argc, char** argv) {
    using namespace Ungar;
    const double seconds = argc > 1 ? std::atof(argv[1]) : 5.0;
    constexpr auto NUM_ROTORS = 4_c;
    UNGAR_VARIABLE(position, 3);
    UNGAR_VARIABLE(orientation, Q);
    UNGAR_VARIABLE(linear_velocity, 3);
    UNGAR_VARIABLE(b_angular_velocity, 3);
    UNGAR_VARIABLE(x) <<= (position, orientation, linear_velocity, b_angular_velocity);
    UNGAR_VARIABLE(rotor_speed, 1);
    UNGAR_VARIABLE(u) <<= NUM_ROTORS * rotor_speed;
    // parameters in the order of the product's node model: dt, m, moi(3), 4 x propeller position(3), g0, b, d
    const auto node = [&](const VectorXad& xp, VectorXad& y) {
        const VectorXad xs = xp.head(13), us = xp.segment(13, 4), ps = xp.tail(20);
        const auto x_ = MakeVariableLazyMap(xs, x);
        const ad_scalar_t dt = ps[0], m = ps[1], g0 = ps[17], b = ps[18], d = ps[19];
        const Vector3ad moi = ps.segment<3>(2);
        const auto [p, q, pDot, bOmega] = x_.GetTuple(position, orientation, linear_velocity, b_angular_velocity);
        Vector3ad sumF = Vector3ad::Zero(), sumM = Vector3ad::Zero(), sumD = Vector3ad::Zero();
        for (int i = 0; i < 4; ++i) {
            const Vector3ad thrust = b * Utils::Pow(us[i], 2) * Vector3ad::UnitZ();
            const Vector3ad pP = ps.segment<3>(5 + 3 * i);
            sumF += thrust;
            sumM += pP.cross(thrust);
            sumD += d * Utils::Pow(us[i], 2) * Vector3ad::UnitZ() * ((i % 2) ? -1.0 : 1.0);
        }
        const Vector3ad pDotDot = (q * sumF - m * g0 * Vector3ad::UnitZ()) / m;
        const Vector3ad bOmegaDot = moi.cwiseInverse().cwiseProduct(sumM + sumD - bOmega.cross(moi.cwiseProduct(bOmega)));
        auto xNext_ = MakeVariableMap<ad_scalar_t>(x);
        auto [pNext, qNext, pDotNext, bOmegaNext] = xNext_.GetTuple(position, orientation, linear_velocity, b_angular_velocity);
        pDotNext = pDot + dt * pDotDot;
        bOmegaNext = bOmega + dt * bOmegaDot;
        pNext = p + dt * pDotNext;
        qNext = q * Utils::ApproximateExponentialMap(dt * bOmegaNext);
        y = xNext_.Get();
    };
    Autodiff::Function::Blueprint bp{node, 17_idx, 20_idx, "ungar_amd_reference_timing_quadrotor_node"sv, EnabledDerivatives::JACOBIAN};
    const Autodiff::Function f = Autodiff::MakeFunction(bp, true);
    std::mt19937 gen{0U};
    std::uniform_real_distribution<real_t> uni{-1.0, 1.0};
    std::vector<VectorXr> inputs(2048, VectorXr::Zero(37));
    for (auto& v : inputs) {
        for (int i = 0; i < 17; ++i) v[i] = uni(gen);
        v.segment<4>(3).normalize();
        v.tail(20) << 1.0 / 30.0, 1.5, 3e-2, 3e-2, 3e-2, 0.2, 0.2, 0.0, -0.2, 0.2, 0.0, -0.2, -0.2, 0.0, 0.2, -0.2, 0.0, 9.80665, 0.015, 0.1;
    }
    long calls = 0;
    double checksum = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    double elapsed = 0.0;
    do {
        for (const auto& v : inputs) checksum += f.Jacobian(v).coeff(0, 0);
        calls += static_cast<long>(inputs.size());
        elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (elapsed < seconds);
    std::printf("{\"evals_per_s\": %.6g, \"nnz\": %ld, \"calls\": %ld, \"checksum\": %.6g}\n", calls / elapsed, static_cast<long>(f.Jacobian(inputs[0]).nonZeros()), calls, checksum);
    return 0;
}
