// GPU test (C++20 host code over the C ABI): the reference's quadrotor OCP AS WRITTEN -- horizon 30, 523 decision variables, the
// input-rate term that couples u_k with u_{k-1} (example/mpc/quadrotor.example.cpp:196-291) -- solved for a BATCH of perturbed
// instances by Ungar::BatchedSoftSQPOptimizer (stage functions, Riccati QP solve on the device) and, for a sample of the instances,
// by the facade's whole-horizon Ungar::SoftSQPOptimizer (sparse KKT solve of the very QP the reference hands to OSQP,
// soft_sqp.hpp:143-158).  Search direction, accepted step size and iterate must agree after one and after two iterations.
//   usage: batched_quadrotor_test <codegen folder> [batch] [compared instances]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "ungar/autodiff/function.hpp"
#include "ungar/autodiff/vector_composer.hpp"
#include "ungar/optimization/batched_soft_sqp.hpp"
#include "ungar/optimization/soft_sqp.hpp"
#include "ungar/variable_map.hpp"

using namespace Ungar;

constexpr auto N = 30_c;
constexpr auto ROTORS = 4_c;
// ---- the whole-horizon variables of the reference's example (:52-117)
UNGAR_VARIABLE(position, 3);
UNGAR_VARIABLE(orientation, Q);
UNGAR_VARIABLE(linear_velocity, 3);
UNGAR_VARIABLE(b_angular_velocity, 3);
UNGAR_VARIABLE(x) <<= (position, orientation, linear_velocity, b_angular_velocity);
UNGAR_VARIABLE(X) <<= (N + 1_c) * x;
UNGAR_VARIABLE(rotor_speed, 1);
UNGAR_VARIABLE(u) <<= ROTORS * rotor_speed;
UNGAR_VARIABLE(U) <<= N * u;
UNGAR_VARIABLE(step_size, 1);
UNGAR_VARIABLE(mass, 1);
UNGAR_VARIABLE(b_moi_diagonal, 3);
UNGAR_VARIABLE(b_propeller_position, 3);
UNGAR_VARIABLE(standard_gravity, 1);
UNGAR_VARIABLE(thrust_constant, 1);
UNGAR_VARIABLE(drag_constant, 1);
UNGAR_VARIABLE(max_rotor_speed, 1);
UNGAR_VARIABLE(reference_position, 3);
UNGAR_VARIABLE(reference_orientation, Q);
UNGAR_VARIABLE(reference_linear_velocity, 3);
UNGAR_VARIABLE(b_reference_angular_velocity, 3);
UNGAR_VARIABLE(measured_position, 3);
UNGAR_VARIABLE(measured_orientation, Q);
UNGAR_VARIABLE(measured_linear_velocity, 3);
UNGAR_VARIABLE(b_measured_angular_velocity, 3);
UNGAR_VARIABLE(measured_state) <<= (measured_position, measured_orientation, measured_linear_velocity, b_measured_angular_velocity);
UNGAR_VARIABLE(decision_variables) <<= (X, U);
UNGAR_VARIABLE(parameters) <<= (step_size, mass, b_moi_diagonal, ROTORS * b_propeller_position, standard_gravity, thrust_constant, drag_constant,
                                max_rotor_speed, (N + 1_c) * reference_position, (N + 1_c) * reference_orientation,
                                (N + 1_c) * reference_linear_velocity, (N + 1_c) * b_reference_angular_velocity, measured_state);
UNGAR_VARIABLE(variables) <<= (decision_variables, parameters);
// ---- one node of the stage form: row = [previous input | x | u | knot parameters | instance parameters]
UNGAR_VARIABLE(previous_input) <<= ROTORS * rotor_speed;
UNGAR_VARIABLE(reference_state) <<= (reference_position, reference_orientation, reference_linear_velocity, b_reference_angular_velocity);
UNGAR_VARIABLE(input_rate_weight, 1);  // 1e-6 for 0 < k < N, else 0  (:222-227)
UNGAR_VARIABLE(input_weight, 1);       // 1e-6 for k < N, 0 at the terminal knot  (:228-232)
UNGAR_VARIABLE(knot_parameters) <<= (reference_state, input_rate_weight, input_weight);
UNGAR_VARIABLE(instance_parameters) <<= (step_size, mass, b_moi_diagonal, ROTORS * b_propeller_position, standard_gravity, thrust_constant, drag_constant, max_rotor_speed);
UNGAR_VARIABLE(dynamics_node) <<= (x, u, knot_parameters, instance_parameters);
UNGAR_VARIABLE(cost_node) <<= (previous_input, x, u, knot_parameters, instance_parameters);

/// quadrotor.example.cpp:126-190 on any map that holds the inertial and physical parameters.
template <class Params>
static VectorXad Dynamics(const VectorXad& xk, const VectorXad& uk, const Params& p_) {
    const auto x_ = MakeVariableLazyMap(xk, x);
    const auto u_ = MakeVariableLazyMap(uk, u);
    const auto [dt, g0, b, d] = p_.GetTuple(step_size, standard_gravity, thrust_constant, drag_constant);
    const auto [m, moi] = p_.GetTuple(mass, b_moi_diagonal);
    const auto [p, q, pDot, bOmega] = x_.GetTuple(position, orientation, linear_velocity, b_angular_velocity);
    Vector3ad force = Vector3ad::Zero(), thrustMoment = Vector3ad::Zero(), dragMoment = Vector3ad::Zero();
    for (const auto i : enumerate(ROTORS)) {
        const auto& r = u_.Get(rotor_speed, i);
        const Vector3ad thrust = b * Utils::Pow(r, 2) * Vector3ad::UnitZ();
        force += thrust;
        thrustMoment += p_.Get(b_propeller_position, i).cross(thrust);
        dragMoment += d * Utils::Pow(r, 2) * Vector3ad::UnitZ() * Utils::Pow(-1.0, i);
    }
    const Vector3ad pDotDot = (q * force - m * g0 * Vector3ad::UnitZ()) / m;
    const Vector3ad bOmegaDot = moi.cwiseInverse().cwiseProduct(thrustMoment + dragMoment - bOmega.cross(moi.cwiseProduct(bOmega)));
    auto next_ = MakeVariableMap<ad_scalar_t>(x);
    auto [pN, qN, pDotN, bOmegaN] = next_.GetTuple(position, orientation, linear_velocity, b_angular_velocity);
    pDotN = pDot + dt * pDotDot;
    bOmegaN = bOmega + dt * bOmegaDot;
    pN = p + dt * pDotN;
    qN = q * Utils::ApproximateExponentialMap(dt * bOmegaN);
    return next_.Get();
}

template <class Map, class Ref>
static ad_scalar_t Tracking(const Map& m_, const Ref& r_) {
    const auto q = m_.Get(orientation);
    const auto qRef = r_.Get(reference_orientation);
    return (m_.Get(position) - r_.Get(reference_position)).squaredNorm() +
           Utils::Min((q.coeffs() - qRef.coeffs()).squaredNorm(), (q.coeffs() + qRef.coeffs()).squaredNorm()) +
           (m_.Get(linear_velocity) - r_.Get(reference_linear_velocity)).squaredNorm() +
           (m_.Get(b_angular_velocity) - r_.Get(b_reference_angular_velocity)).squaredNorm();
}

int main(int argc, char** argv) {
    const std::string folder = argc > 1 ? argv[1] : "/tmp/ungar_amd_batched_quadrotor";
    const index_t batch = argc > 2 ? std::atol(argv[2]) : 1024, compared = argc > 3 ? std::atol(argv[3]) : 8;
    try {
        // ---- whole-horizon problem (what SoftSQPOptimizer consumes; quadrotor.example.cpp:196-320)
        const auto objective = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            ad_scalar_t value{0.0};
            for (const auto k : enumerate(N + 1_step)) {
                const auto q = v_.Get(orientation, k);
                const auto qRef = v_.Get(reference_orientation, k);
                value += (v_.Get(position, k) - v_.Get(reference_position, k)).squaredNorm() +
                         Utils::Min((q.coeffs() - qRef.coeffs()).squaredNorm(), (q.coeffs() + qRef.coeffs()).squaredNorm()) +
                         (v_.Get(linear_velocity, k) - v_.Get(reference_linear_velocity, k)).squaredNorm() +
                         (v_.Get(b_angular_velocity, k) - v_.Get(b_reference_angular_velocity, k)).squaredNorm();
                if (k && k != N) value += 1e-6 * (v_.Get(u, k) - v_.Get(u, k - 1_step)).squaredNorm();
                if (k != N) value += 1e-6 * v_.Get(u, k).squaredNorm();
            }
            y.resize(1_idx);
            y << value;
        };
        const auto equality = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            Autodiff::VectorComposer composer;
            composer << v_.Get(x, 0_step) - v_.Get(measured_state);
            for (const auto k : enumerate(N)) {
                const VectorXad par = v_.Get(parameters);
                composer << v_.Get(x, k + 1_step) - Dynamics(v_.Get(x, k), v_.Get(u, k), MakeVariableLazyMap(par, parameters));
            }
            y = composer.Compose();
        };
        const auto inequality = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            Autodiff::VectorComposer composer;
            const auto& rMax = v_.Get(max_rotor_speed);
            for (const auto k : enumerate(N))
                for (const auto i : enumerate(ROTORS)) {
                    const auto& r = v_.Get(rotor_speed, k, i);
                    composer << r - rMax;
                    composer << -r;
                }
            y = composer.Compose();
        };
        auto nlp = MakeNLPProblem(
            Autodiff::MakeFunction({objective, decision_variables.Size(), parameters.Size(), "bq_whole_obj", EnabledDerivatives::ALL, folder}, false),
            Autodiff::MakeFunction({equality, decision_variables.Size(), parameters.Size(), "bq_whole_eqs", EnabledDerivatives::JACOBIAN, folder}, false),
            Autodiff::MakeFunction({inequality, decision_variables.Size(), parameters.Size(), "bq_whole_ineqs", EnabledDerivatives::JACOBIAN, folder}, false));

        // ---- the same problem in stage form
        const index_t nPar = knot_parameters.Size() + instance_parameters.Size();
        const auto stageDynamics = [&](const VectorXad& v, VectorXad& y) {
            const auto n_ = MakeVariableLazyMap(v, dynamics_node);
            const VectorXad xk = n_.Get(x), uk = n_.Get(u), par = n_.Get(instance_parameters);
            y = Dynamics(xk, uk, MakeVariableLazyMap(par, instance_parameters));
        };
        const auto stageCost = [&](const VectorXad& v, VectorXad& y) {
            const auto n_ = MakeVariableLazyMap(v, cost_node);
            const VectorXad xk = n_.Get(x), ref = n_.Get(reference_state);
            const auto x_ = MakeVariableLazyMap(xk, x);
            const auto r_ = MakeVariableLazyMap(ref, reference_state);
            y.resize(1_idx);
            y << Tracking(x_, r_) + n_.Get(input_rate_weight) * (n_.Get(u) - n_.Get(previous_input)).squaredNorm() + n_.Get(input_weight) * n_.Get(u).squaredNorm();
        };
        const auto stageInequality = [&](const VectorXad& v, VectorXad& y) {
            const auto n_ = MakeVariableLazyMap(v, cost_node);
            const VectorXad uk = n_.Get(u);
            const auto u_ = MakeVariableLazyMap(uk, u);
            Autodiff::VectorComposer composer;
            for (const auto i : enumerate(ROTORS)) {
                const auto& r = u_.Get(rotor_speed, i);
                composer << r - n_.Get(max_rotor_speed);
                composer << -r;
            }
            y = composer.Compose();
        };
        ShootingProblem problem;
        problem.horizon = N;
        problem.stateSize = x.Size();
        problem.inputSize = u.Size();
        problem.carrySize = u.Size();
        problem.carryInputs = true;
        problem.knotParameterSize = knot_parameters.Size();
        problem.instanceParameterSize = instance_parameters.Size();
        problem.dynamics.emplace(Autodiff::MakeFunction({stageDynamics, x.Size() + u.Size(), nPar, "bq_stage_dyn", EnabledDerivatives::JACOBIAN, folder}, false));
        problem.cost.emplace(Autodiff::MakeFunction({stageCost, u.Size() + x.Size() + u.Size(), nPar, "bq_stage_cost", EnabledDerivatives::ALL, folder}, false));
        problem.inequality.emplace(Autodiff::MakeFunction({stageInequality, u.Size() + x.Size() + u.Size(), nPar, "bq_stage_ineq", EnabledDerivatives::JACOBIAN, folder}, false));
        const index_t nv = problem.RowSize();
        if (nv != cost_node.Size()) {
            std::printf("FAIL row size %td vs %td\n", nv, cost_node.Size());
            return 1;
        }
        if (const char* groups = std::getenv("UNGAR_AMD_STACKED_CANDIDATES")) BatchedSoftSQPOptimizer::maxStackedCandidates = std::atol(groups);  // (test of the group logic)
        BatchedSoftSQPOptimizer batched{std::move(problem), batch, false, 1.0, 2};
        if (const char* stage = std::getenv("UNGAR_TEST_FIRST_STAGE")) batched.SetFirstLineSearchStage(std::atol(stage));  // staged line search: same iterates

        // ---- perturbed instances: one whole-horizon variable vector each (parameter values of quadrotor.example.cpp:326-358)
        std::mt19937_64 rng{20260929};
        std::normal_distribution<real_t> normal{0.0, 1.0};
        std::vector<VectorXr> instances;
        const real_t hover = std::sqrt(1.5 * 9.80665 / 0.015 / 4.0);
        for (index_t b = 0; b < batch; ++b) {
            auto v_ = MakeVariableMap<real_t>(variables);
            v_.Get(step_size) = 1.0 / static_cast<real_t>(N);
            v_.Get(mass) = 1.5;
            v_.Get(b_moi_diagonal).setConstant(3e-2);
            v_.Get(b_propeller_position, 0) = Vector3r(0.2, 0.2, 0.0);
            v_.Get(b_propeller_position, 1) = Vector3r(-0.2, 0.2, 0.0);
            v_.Get(b_propeller_position, 2) = Vector3r(-0.2, -0.2, 0.0);
            v_.Get(b_propeller_position, 3) = Vector3r(0.2, -0.2, 0.0);
            v_.Get(standard_gravity) = 9.80665;
            v_.Get(thrust_constant) = 0.015;
            v_.Get(drag_constant) = 0.1;
            v_.Get(max_rotor_speed) = b % 3 == 0 ? 16.0 : 1e2;  // every third instance flies with its rotor-speed bound active (hover is 15.66)
            const real_t yaw = 0.3 * normal(rng), climb = 0.2 * normal(rng);
            for (const auto k : enumerate(N + 1_step)) {
                const real_t t = static_cast<real_t>(k) / static_cast<real_t>(N);
                v_.Get(position, k) = Vector3r(0.3 * t + 0.02 * normal(rng), -0.2 * t * t + 0.02 * normal(rng), 4.0 + climb * t + 0.02 * normal(rng));
                v_.Get(orientation, k) = Quaternionr(1.0, 0.1 * t + 0.02 * normal(rng), -0.05 * t + 0.02 * normal(rng), yaw * t).normalized();
                v_.Get(linear_velocity, k) = Vector3r(0.3 + 0.05 * normal(rng), -0.4 * t, climb + 0.05 * normal(rng));
                v_.Get(b_angular_velocity, k) = Vector3r(0.1 * normal(rng), -0.2 * t, yaw);
                v_.Get(reference_position, k) = Vector3r(0.5 * t, 0.1 * normal(rng), 4.0 + climb);
                v_.Get(reference_orientation, k) = Quaternionr(b % 2 ? -1.0 : 1.0, 0.0, 0.0, 0.5 * yaw * t).normalized();  // odd instances: other hemisphere (Min)
                v_.Get(reference_linear_velocity, k) = Vector3r(0.5, 0.0, 0.0);
                v_.Get(b_reference_angular_velocity, k) = Vector3r(0.0, 0.0, 0.5 * yaw);
            }
            for (const auto k : enumerate(N))
                for (const auto i : enumerate(ROTORS)) v_.Get(rotor_speed, k, i) = hover * (1.0 + 0.03 * normal(rng));
            v_.Get(measured_position) = v_.Get(position, 0_step) + 0.01 * Vector3r(normal(rng), normal(rng), normal(rng));
            v_.Get(measured_orientation) = v_.Get(orientation, 0_step);
            v_.Get(measured_linear_velocity) = v_.Get(linear_velocity, 0_step) + 0.01 * Vector3r(normal(rng), normal(rng), normal(rng));
            v_.Get(b_measured_angular_velocity) = v_.Get(b_angular_velocity, 0_step);
            instances.push_back(v_.Get());
        }

        // ---- node rows of every instance
        std::vector<real_t> rows(static_cast<std::size_t>(batched.RowsSize())), xm(static_cast<std::size_t>(batch * x.Size()));
        for (index_t b = 0; b < batch; ++b) {
            const auto v_ = MakeVariableLazyMap(instances[static_cast<std::size_t>(b)], variables);
            for (index_t k = 0; k <= N; ++k) {
                VectorXr row{nv};
                auto n_ = MakeVariableLazyMap(row, cost_node);
                n_.Get(previous_input).setZero();  // rows 1..N: filled by the optimizer; row 0: no previous input (weight 0)
                n_.Get(x) = v_.Get(x, k);
                n_.Get(u) = v_.Get(u, k < N ? k : N - 1);  // row N: a dummy input (weights 0)
                n_.Get(reference_position) = v_.Get(reference_position, k);
                n_.Get(reference_orientation) = v_.Get(reference_orientation, k);
                n_.Get(reference_linear_velocity) = v_.Get(reference_linear_velocity, k);
                n_.Get(b_reference_angular_velocity) = v_.Get(b_reference_angular_velocity, k);
                n_.Get(input_rate_weight) = (k > 0 && k < N) ? 1e-6 : 0.0;
                n_.Get(input_weight) = k < N ? 1e-6 : 0.0;
                n_.Get(step_size) = v_.Get(step_size);
                n_.Get(mass) = v_.Get(mass);
                n_.Get(b_moi_diagonal) = v_.Get(b_moi_diagonal);
                for (const auto i : enumerate(ROTORS)) n_.Get(b_propeller_position, i) = v_.Get(b_propeller_position, i);
                n_.Get(standard_gravity) = v_.Get(standard_gravity);
                n_.Get(thrust_constant) = v_.Get(thrust_constant);
                n_.Get(drag_constant) = v_.Get(drag_constant);
                n_.Get(max_rotor_speed) = v_.Get(max_rotor_speed);
                for (index_t j = 0; j < nv; ++j) rows[static_cast<std::size_t>((b * (N + 1) + k) * nv + j)] = row[j];
            }
            const VectorXr m = v_.Get(measured_state);
            for (index_t j = 0; j < x.Size(); ++j) xm[static_cast<std::size_t>(b * x.Size() + j)] = m[j];
        }
        batched.SetRows(rows.data(), xm.data());

        // ---- iterate both; compare on the sampled instances
        const index_t nx = x.Size(), nu = u.Size(), nz = nx + nu, dec = decision_variables.Size();
        real_t worstStep = 0.0, worstIterate = 0.0, worstAlpha = 0.0;
        std::vector<VectorXr> facade(static_cast<std::size_t>(compared));
        std::vector<index_t> sample;
        for (index_t s = 0; s < compared; ++s) sample.push_back(s < 3 ? s : (s * 131 + 7) % batch);  // (the first three cover bound-active / other-hemisphere instances)
        for (int iteration = 1; iteration <= 2; ++iteration) {
            batched.Iterate();
            const std::vector<real_t> dZ = batched.StateSteps(), dU = batched.InputSteps(), accepted = batched.AcceptedStepSizes();
            const std::vector<int32_t> status = batched.QpStatus();
            batched.GetRows(rows.data());
            for (index_t s = 0; s < compared; ++s) {
                const index_t b = sample[static_cast<std::size_t>(s)];
                if (status[static_cast<std::size_t>(b)] != 0) {
                    std::printf("FAIL QP status %d for instance %td\n", status[static_cast<std::size_t>(b)], b);
                    return 1;
                }
                VectorXr& z = facade[static_cast<std::size_t>(s)];
                if (iteration == 1) z = instances[static_cast<std::size_t>(b)];
                const VectorXr before = z;
                SoftSQPOptimizer optimizer{false, 1.0, index_t{1}};
                const VectorXr after = optimizer.Optimize(nlp, z);
                const std::vector<real_t>& d = optimizer.LastStep();
                real_t num = 0.0, den = 0.0, scaleD = 0.0, scaleX = 0.0;
                for (index_t i = 0; i < dec; ++i) {
                    num += (after[i] - before[i]) * d[static_cast<std::size_t>(i)];
                    den += d[static_cast<std::size_t>(i)] * d[static_cast<std::size_t>(i)];
                    scaleD = std::max(scaleD, std::abs(d[static_cast<std::size_t>(i)]));
                    scaleX = std::max(scaleX, std::abs(after[i]));
                }
                const real_t alphaFacade = den > 0.0 ? num / den : 0.0;
                worstAlpha = std::max(worstAlpha, std::abs(alphaFacade - accepted[static_cast<std::size_t>(b)]));
                for (index_t k = 0; k <= N; ++k)
                    for (index_t i = 0; i < nx; ++i) {
                        worstStep = std::max(worstStep, std::abs(dZ[static_cast<std::size_t>((b * (N + 1) + k) * nz + nu + i)] - d[static_cast<std::size_t>(k * nx + i)]) / scaleD);
                        worstIterate = std::max(worstIterate, std::abs(rows[static_cast<std::size_t>((b * (N + 1) + k) * nv + nu + i)] - after[k * nx + i]) / scaleX);
                    }
                for (index_t k = 0; k < N; ++k)
                    for (index_t i = 0; i < nu; ++i) {
                        worstStep = std::max(worstStep, std::abs(dU[static_cast<std::size_t>((b * N + k) * nu + i)] - d[static_cast<std::size_t>((N + 1) * nx + k * nu + i)]) / scaleD);
                        worstIterate = std::max(worstIterate, std::abs(rows[static_cast<std::size_t>((b * (N + 1) + k) * nv + nz + i)] - after[(N + 1) * nx + k * nu + i]) / scaleX);
                    }
                for (index_t i = 0; i < dec; ++i) z[i] = after[i];
                std::printf("iteration %d instance %4td: step size facade %.6g batched %.6g\n", iteration, b, alphaFacade, accepted[static_cast<std::size_t>(b)]);
            }
            index_t moved = 0;
            for (const real_t a : accepted) moved += a > 0.0;
            std::printf("iteration %d: %td of %td instances accepted a step; worst |d - d_facade| / |d|max = %.3e, worst |x - x_facade| / |x|max = %.3e, worst step-size difference %.3e\n",
                        iteration, moved, batch, worstStep, worstIterate, worstAlpha);
        }
        {   // wall clock of further iterations (all instances, device only): synchronise, iterate, synchronise
            const int timed = 5;
            (void)batched.AcceptedStepSizes();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < timed; ++i) batched.Iterate();
            (void)batched.AcceptedStepSizes();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
            std::printf("timing: %.3f ms per SQP iteration of %td instances (%.3g instances/s)\n", ms, batch, static_cast<double>(batch) / ms * 1e3);
        }
        const bool ok = worstStep <= 1e-9 && worstIterate <= 1e-9 && worstAlpha <= 1e-9;
        std::printf("%s batched quadrotor SQP (batch %td, %td compared)\n", ok ? "PASS" : "FAIL", batch, compared);
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
}
