// GPU test (C++20 host code over the C ABI): the reference's RC-car OCP AS WRITTEN -- Pacejka tyre model, horizon 30, 246 decision
// variables, input-rate term coupling u_k with u_{k-1}, terminal position tracking, |d|, |delta| <= 15 through Utils::Abs and a minimum
// forward velocity behind the POLY barrier (example/mpc/rc_car.example.cpp:131-285) -- solved for a BATCH of perturbed instances by
// Ungar::BatchedSoftSQPOptimizer (stage functions with the previous input carried in the stage state, 8 + 2 Riccati block) and, for a
// sample of the instances, by the facade's whole-horizon Ungar::SoftSQPOptimizer.  Search direction, accepted step size and iterate must
// agree after one and after two iterations.     usage: batched_rc_car_test <codegen folder> [batch] [compared instances]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
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
// ---- the whole-horizon variables of the reference's example (:50-122)
UNGAR_VARIABLE(position, 2);
UNGAR_VARIABLE(yaw, 1);
UNGAR_VARIABLE(b_linear_velocity, 2);
UNGAR_VARIABLE(yaw_rate, 1);
UNGAR_VARIABLE(x) <<= (position, yaw, b_linear_velocity, yaw_rate);
UNGAR_VARIABLE(X) <<= (N + 1_c) * x;
UNGAR_VARIABLE(pwm_duty_cycle, 1);
UNGAR_VARIABLE(steering_angle, 1);
UNGAR_VARIABLE(u) <<= (pwm_duty_cycle, steering_angle);
UNGAR_VARIABLE(U) <<= N * u;
UNGAR_VARIABLE(step_size, 1);
UNGAR_VARIABLE(mass, 1);
UNGAR_VARIABLE(b_moi, 1);
UNGAR_VARIABLE(front_wheel_distance, 1);
UNGAR_VARIABLE(rear_wheel_distance, 1);
UNGAR_VARIABLE(ptm_front_b, 1);
UNGAR_VARIABLE(ptm_front_c, 1);
UNGAR_VARIABLE(ptm_front_d, 1);
UNGAR_VARIABLE(ptm_rear_b, 1);
UNGAR_VARIABLE(ptm_rear_c, 1);
UNGAR_VARIABLE(ptm_rear_d, 1);
UNGAR_VARIABLE(ptm_cm1, 1);
UNGAR_VARIABLE(ptm_cm2, 1);
UNGAR_VARIABLE(ptm_cr0, 1);
UNGAR_VARIABLE(ptm_cr2, 1);
UNGAR_VARIABLE(reference_position, 2);
UNGAR_VARIABLE(reference_trajectory) <<= (N + 1_c) * reference_position;
UNGAR_VARIABLE(measured_position, 2);
UNGAR_VARIABLE(measured_yaw, 1);
UNGAR_VARIABLE(b_measured_linear_velocity, 2);
UNGAR_VARIABLE(measured_yaw_rate, 1);
UNGAR_VARIABLE(measured_state) <<= (measured_position, measured_yaw, b_measured_linear_velocity, measured_yaw_rate);
UNGAR_VARIABLE(car) <<= (step_size, mass, b_moi, front_wheel_distance, rear_wheel_distance, ptm_front_b, ptm_front_c, ptm_front_d, ptm_rear_b, ptm_rear_c, ptm_rear_d, ptm_cm1,
                         ptm_cm2, ptm_cr0, ptm_cr2);  // the 15 model parameters, in the order of the example's `parameters`
UNGAR_VARIABLE(decision_variables) <<= (X, U);
UNGAR_VARIABLE(parameters) <<= (car, reference_trajectory, measured_state);
UNGAR_VARIABLE(variables) <<= (decision_variables, parameters);
// ---- one node of the stage form: row = [previous input | x | u | knot parameters | instance parameters]
UNGAR_VARIABLE(previous_duty_cycle, 1);
UNGAR_VARIABLE(previous_steering_angle, 1);
UNGAR_VARIABLE(previous_input) <<= (previous_duty_cycle, previous_steering_angle);
UNGAR_VARIABLE(input_rate_weight, 1);  // 1e-6 for 0 < k < N  (:216-220)
UNGAR_VARIABLE(input_weight, 1);       // 1e-6 for k < N      (:214)
UNGAR_VARIABLE(knot_parameters) <<= (reference_position, input_rate_weight, input_weight);
UNGAR_VARIABLE(dynamics_node) <<= (x, u, knot_parameters, car);
UNGAR_VARIABLE(stage_node) <<= (previous_input, x, u, knot_parameters, car);

/// rc_car.example.cpp:131-185 on any map that holds the model parameters.
template <class Params>
static VectorXad Dynamics(const VectorXad& xk, const VectorXad& uk, const Params& p_) {
    using std::atan;
    using std::cos;
    using std::sin;
    const auto x_ = MakeVariableLazyMap(xk, x);
    const auto u_ = MakeVariableLazyMap(uk, u);
    const auto& dt = p_.Get(step_size);
    const auto [Bf, Cf, Df] = p_.GetTuple(ptm_front_b, ptm_front_c, ptm_front_d);
    const auto [Br, Cr, Dr] = p_.GetTuple(ptm_rear_b, ptm_rear_c, ptm_rear_d);
    const auto [Cm1, Cm2] = p_.GetTuple(ptm_cm1, ptm_cm2);
    const auto [Cr0, Cr2] = p_.GetTuple(ptm_cr0, ptm_cr2);
    const auto [m, bMOI, lf, lr] = p_.GetTuple(mass, b_moi, front_wheel_distance, rear_wheel_distance);
    const auto [p, phi, v, omega] = x_.GetTuple(position, yaw, b_linear_velocity, yaw_rate);
    const auto [d, delta] = u_.GetTuple(pwm_duty_cycle, steering_angle);
    const real_t eps = std::numeric_limits<real_t>::epsilon();
    const ad_scalar_t alphaf = -atan((omega * lf + v.y()) / (v.x() + eps)) + delta;
    const ad_scalar_t alphar = atan((omega * lr - v.y()) / (v.x() + eps));
    const ad_scalar_t Ffy = Df * sin(Cf * atan(Bf * alphaf));
    const ad_scalar_t Fry = Dr * sin(Cr * atan(Br * alphar));
    const ad_scalar_t Frx = (Cm1 - Cm2 * v.x()) * d - Cr0 - Cr2 * Utils::Pow(v.x(), 2);
    const Vector2ad vDot{(Frx - Ffy * sin(delta) + m * v.y() * omega) / m, (Fry + Ffy * cos(delta) - m * v.x() * omega) / m};
    const ad_scalar_t omegaDot = (Ffy * lf * cos(delta) - Fry * lr) / bMOI;
    auto next_ = MakeVariableMap<ad_scalar_t>(x);
    auto [pN, phiN, vN, omegaN] = next_.GetTuple(position, yaw, b_linear_velocity, yaw_rate);
    vN = v + dt * vDot;
    omegaN = omega + dt * omegaDot;
    pN = p + dt * Vector2ad{vN.x() * cos(phi) - vN.y() * sin(phi), vN.x() * sin(phi) + vN.y() * cos(phi)};
    phiN = phi + dt * omegaN;
    return next_.Get();
}

int main(int argc, char** argv) {
    const std::string folder = argc > 1 ? argv[1] : "/tmp/ungar_amd_batched_rc_car";
    const index_t batch = argc > 2 ? std::atol(argv[2]) : 1024, compared = argc > 3 ? std::atol(argv[3]) : 8;
    try {
        // ---- whole-horizon problem (rc_car.example.cpp:191-310)
        const auto objective = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            ad_scalar_t value{0.0};
            for (const auto k : enumerate(N)) {
                value += (v_.Get(position, k) - v_.Get(reference_position, k)).squaredNorm();
                value += 1e-6 * v_.Get(u, k).squaredNorm();
                if (k) value += 1e-6 * (v_.Get(u, k) - v_.Get(u, k - 1_step)).squaredNorm();
            }
            value += (v_.Get(position, N) - v_.Get(reference_position, N)).squaredNorm();
            y.resize(1_idx);
            y << value;
        };
        const auto equality = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            Autodiff::VectorComposer composer;
            composer << v_.Get(x, 0_step) - v_.Get(measured_state);
            for (const auto k : enumerate(N)) {
                const VectorXad model = v_.Get(car);
                composer << v_.Get(x, k + 1_step) - Dynamics(v_.Get(x, k), v_.Get(u, k), MakeVariableLazyMap(model, car));
            }
            y = composer.Compose();
        };
        const auto inequality = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            Autodiff::VectorComposer composer;
            for (const auto k : enumerate(N)) {
                composer << Utils::Abs(v_.Get(pwm_duty_cycle, k)) - 15.0;
                composer << Utils::Abs(v_.Get(steering_angle, k)) - 15.0;
                composer << 0.3 - v_.Get(b_linear_velocity, k).x();
            }
            y = composer.Compose();
        };
        auto nlp = MakeNLPProblem(
            Autodiff::MakeFunction({objective, decision_variables.Size(), parameters.Size(), "brc_whole_obj", EnabledDerivatives::ALL, folder}, false),
            Autodiff::MakeFunction({equality, decision_variables.Size(), parameters.Size(), "brc_whole_eqs", EnabledDerivatives::JACOBIAN, folder}, false),
            Autodiff::MakeFunction({inequality, decision_variables.Size(), parameters.Size(), "brc_whole_ineqs", EnabledDerivatives::JACOBIAN, folder}, false));
        if (decision_variables.Size() != 246 || nlp.equalityConstraints.DependentVariableSize() != 186 || nlp.inequalityConstraints.DependentVariableSize() != 90) {
            std::printf("FAIL sizes\n");
            return 1;
        }

        // ---- the same problem in stage form
        const index_t nPar = knot_parameters.Size() + car.Size(), nxu = x.Size() + u.Size(), nd = u.Size() + nxu;
        const auto stageDynamics = [&](const VectorXad& v, VectorXad& y) {
            const auto n_ = MakeVariableLazyMap(v, dynamics_node);
            const VectorXad xk = n_.Get(x), uk = n_.Get(u), model = n_.Get(car);
            y = Dynamics(xk, uk, MakeVariableLazyMap(model, car));
        };
        const auto stageCost = [&](const VectorXad& v, VectorXad& y) {
            const auto n_ = MakeVariableLazyMap(v, stage_node);
            y.resize(1_idx);
            y << (n_.Get(position) - n_.Get(reference_position)).squaredNorm() + n_.Get(input_weight) * n_.Get(u).squaredNorm() +
                     n_.Get(input_rate_weight) * (n_.Get(u) - n_.Get(previous_input)).squaredNorm();
        };
        const auto stageInequality = [&](const VectorXad& v, VectorXad& y) {
            const auto n_ = MakeVariableLazyMap(v, stage_node);
            Autodiff::VectorComposer composer;
            composer << Utils::Abs(n_.Get(pwm_duty_cycle)) - 15.0;
            composer << Utils::Abs(n_.Get(steering_angle)) - 15.0;
            composer << 0.3 - n_.Get(b_linear_velocity).x();
            y = composer.Compose();
        };
        ShootingProblem problem;
        problem.horizon = N;
        problem.stateSize = x.Size();
        problem.inputSize = u.Size();
        problem.carrySize = u.Size();
        problem.carryInputs = true;
        problem.knotParameterSize = knot_parameters.Size();
        problem.instanceParameterSize = car.Size();
        problem.dynamics.emplace(Autodiff::MakeFunction({stageDynamics, nxu, nPar, "brc_stage_dyn", EnabledDerivatives::JACOBIAN, folder}, false));
        problem.cost.emplace(Autodiff::MakeFunction({stageCost, nd, nPar, "brc_stage_cost", EnabledDerivatives::ALL, folder}, false));
        problem.inequality.emplace(Autodiff::MakeFunction({stageInequality, nd, nPar, "brc_stage_ineq", EnabledDerivatives::JACOBIAN, folder}, false));
        const index_t nv = problem.RowSize();
        if (nv != stage_node.Size()) {
            std::printf("FAIL row size\n");
            return 1;
        }
        const real_t dt = 1.0 / static_cast<real_t>(N);
        if (const char* groups = std::getenv("UNGAR_AMD_STACKED_CANDIDATES")) BatchedSoftSQPOptimizer::maxStackedCandidates = std::atol(groups);  // (test of the group logic)
        BatchedSoftSQPOptimizer batched{std::move(problem), batch, false, dt, 2, 100.0, 1e-2};  // the example's optimizer settings (:363)

        // ---- perturbed instances (parameter values of rc_car.example.cpp:320-352)
        std::mt19937_64 rng{20261001};
        std::normal_distribution<real_t> normal{0.0, 1.0};
        std::vector<VectorXr> instances;
        for (index_t b = 0; b < batch; ++b) {
            auto v_ = MakeVariableMap<real_t>(variables);
            v_.Get(mass) = 0.041;
            v_.Get(b_moi) = 27.8e-6;
            v_.Get(front_wheel_distance) = 0.029;
            v_.Get(rear_wheel_distance) = 0.033;
            v_.Get(step_size) = dt;
            v_.Get(ptm_front_b) = 2.579;
            v_.Get(ptm_front_c) = 1.2;
            v_.Get(ptm_front_d) = 0.192;
            v_.Get(ptm_rear_b) = 3.3852;
            v_.Get(ptm_rear_c) = 1.2691;
            v_.Get(ptm_rear_d) = 0.1737;
            v_.Get(ptm_cm1) = 0.287;
            v_.Get(ptm_cm2) = 0.0545;
            v_.Get(ptm_cr0) = 0.0518;
            v_.Get(ptm_cr2) = 0.00035;
            const real_t speed = b % 4 == 3 ? 0.32 : 1.0 + 0.2 * normal(rng);  // every fourth car crawls at the minimum-velocity bound (barrier active)
            const real_t phase = 0.5 * normal(rng);
            v_.Get(measured_position) = 0.02 * Vector2r(normal(rng), normal(rng));
            v_.Get(measured_yaw) = 0.05 * normal(rng);
            v_.Get(b_measured_linear_velocity) = Vector2r(speed, 0.02 * normal(rng));
            v_.Get(measured_yaw_rate) = 0.05 * normal(rng);
            for (const auto k : enumerate(N + 1_step)) {
                const real_t t = static_cast<real_t>(k) * dt;
                v_.Get(position, k) = v_.Get(measured_position) + Vector2r(speed * t + 0.01 * normal(rng), 0.01 * normal(rng));
                v_.Get(yaw, k) = v_.Get(measured_yaw) + 0.02 * normal(rng);
                v_.Get(b_linear_velocity, k) = Vector2r(speed + 0.02 * normal(rng), 0.02 * normal(rng));
                v_.Get(yaw_rate, k) = 0.05 * normal(rng);
                v_.Get(reference_position, k) = Vector2r(1.0 * t, 0.2 * std::sin(2.0 * 3.141592653589793 / 8.0 * t + phase));
            }
            for (const auto k : enumerate(N)) {
                v_.Get(pwm_duty_cycle, k) = 0.3 * normal(rng);  // both signs: both branches of Utils::Abs
                v_.Get(steering_angle, k) = 0.1 * normal(rng);
            }
            instances.push_back(v_.Get());
        }

        if (compared < 0) {  // `batched_rc_car_test <folder> <batch> -1 <file>`: dump the whole-horizon functions of the OCP as written at instance 3 (a car at the minimum-velocity bound) for tests/test_whole_horizon.py
            const std::string file = argc > 4 ? argv[4] : "";
            const VectorXr& in = instances[static_cast<std::size_t>(batch > 3 ? 3 : 0)];
            std::ofstream out(file);
            out.precision(17);
            auto vector = [&](const char* tag, const VectorXr& v) {
                out << tag << " " << v.size() << "\n";
                for (index_t i = 0; i < v.size(); ++i) out << v[i] << "\n";
            };
            auto sparse = [&](const char* tag, const Autodiff::SparseMatrix& A) {
                out << tag << " " << A.rows() << " " << A.cols() << " " << A.nonZeros() << "\n";
                for (index_t r = 0; r < A.rows(); ++r)
                    for (int k = A.outerIndexPtr()[r]; k < A.outerIndexPtr()[r + 1]; ++k) out << r << " " << A.innerIndexPtr()[k] << " " << A.valuePtr()[k] << "\n";
            };
            vector("INPUT", in);
            vector("OBJ", nlp.objective(in));
            sparse("OBJ_JAC", nlp.objective.Jacobian(in));
            sparse("OBJ_HES", nlp.objective.Hessian(in));
            vector("EQ", nlp.equalityConstraints(in));
            sparse("EQ_JAC", nlp.equalityConstraints.Jacobian(in));
            vector("INEQ", nlp.inequalityConstraints(in));
            sparse("INEQ_JAC", nlp.inequalityConstraints.Jacobian(in));
            std::printf("DUMPED %s\n", file.c_str());
            return 0;
        }
        // ---- node rows
        const index_t nx = x.Size(), nu = u.Size(), nz = nx + nu, dec = decision_variables.Size();
        std::vector<real_t> rows(static_cast<std::size_t>(batched.RowsSize())), xm(static_cast<std::size_t>(batch * nx));
        for (index_t b = 0; b < batch; ++b) {
            const auto v_ = MakeVariableLazyMap(instances[static_cast<std::size_t>(b)], variables);
            for (index_t k = 0; k <= N; ++k) {
                VectorXr row{nv};
                row.setZero();
                auto n_ = MakeVariableLazyMap(row, stage_node);
                n_.Get(x) = v_.Get(x, k);
                n_.Get(u) = v_.Get(u, k < N ? k : N - 1);  // row N: a dummy input (weights 0)
                n_.Get(reference_position) = v_.Get(reference_position, k);
                n_.Get(input_rate_weight) = (k > 0 && k < N) ? 1e-6 : 0.0;
                n_.Get(input_weight) = k < N ? 1e-6 : 0.0;
                n_.Get(car) = v_.Get(car);
                for (index_t j = 0; j < nv; ++j) rows[static_cast<std::size_t>((b * (N + 1) + k) * nv + j)] = row[j];
            }
            const VectorXr m = v_.Get(measured_state);
            for (index_t j = 0; j < nx; ++j) xm[static_cast<std::size_t>(b * nx + j)] = m[j];
        }
        batched.SetRows(rows.data(), xm.data());

        // ---- iterate both; compare on the sampled instances
        real_t worstStep = 0.0, worstIterate = 0.0, worstAlpha = 0.0;
        std::vector<VectorXr> facade(static_cast<std::size_t>(compared));
        std::vector<index_t> sample;
        for (index_t s = 0; s < compared; ++s) sample.push_back(s < 4 ? s : (s * 131 + 7) % batch);
        for (int iteration = 1; iteration <= 2; ++iteration) {
            batched.Iterate();
            const std::vector<real_t> dZ = batched.StateSteps(), dU = batched.InputSteps(), accepted = batched.AcceptedStepSizes();
            const std::vector<int32_t> status = batched.QpStatus();
            batched.GetRows(rows.data());
            index_t failed = 0;
            for (const int32_t st : status) failed += st != 0;
            if (failed) {
                std::printf("FAIL %td instances report an unsolved QP\n", failed);
                return 1;
            }
            for (index_t s = 0; s < compared; ++s) {
                const index_t b = sample[static_cast<std::size_t>(s)];
                VectorXr& z = facade[static_cast<std::size_t>(s)];
                if (iteration == 1) z = instances[static_cast<std::size_t>(b)];
                const VectorXr before = z;
                SoftSQPOptimizer optimizer{false, dt, index_t{1}, 100.0, 1e-2};
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
                std::printf("iteration %d instance %4td: step size facade %.6g batched %.6g  (|d|max %.3g)\n", iteration, b, alphaFacade, accepted[static_cast<std::size_t>(b)], scaleD);
            }
            index_t moved = 0;
            for (const real_t a : accepted) moved += a > 0.0;
            std::printf("iteration %d: %td of %td instances accepted a step; worst |d - d_facade| / |d|max = %.3e, worst |x - x_facade| / |x|max = %.3e, worst step-size difference %.3e\n",
                        iteration, moved, batch, worstStep, worstIterate, worstAlpha);
        }
        {
            const int timed = 5;
            (void)batched.AcceptedStepSizes();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < timed; ++i) batched.Iterate();
            (void)batched.AcceptedStepSizes();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
            std::printf("timing: %.3f ms per SQP iteration of %td instances (%.3g instances/s)\n", ms, batch, static_cast<double>(batch) / ms * 1e3);
        }
        const bool ok = worstStep <= 1e-9 && worstIterate <= 1e-9 && worstAlpha <= 1e-9;
        std::printf("%s batched rc_car SQP (batch %td, %td compared)\n", ok ? "PASS" : "FAIL", batch, compared);
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
}
