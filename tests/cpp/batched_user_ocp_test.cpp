// GPU test (C++20 host code over the C ABI): a USER's optimal-control problem of stage sizes the library was NOT compiled for -- compile with
// -DUSER_NX=10 -DUSER_NU=3 -DUSER_NE=0 (10 + 3) or -DUSER_NX=20 -DUSER_NU=9 -DUSER_NE=4 (20 + 9 with 4 stage equality rows) -- through
// Ungar::BatchedSoftSQPOptimizer and, for a sample of the instances, through the facade's whole-horizon Ungar::SoftSQPOptimizer (the reference's interface:
// optimization/concepts.hpp:153-262 takes ANY NLPProblem, soft_sqp.hpp:42-281).  The batched driver must run the register-resident Riccati recursion and
// the one-wavefront assembly kernel INSTANTIATED FOR THESE SIZES by the kernel factory (routes 2 / 2, or 3 for the assembly without equality rows), and
// search direction, accepted step size and iterate must agree with the facade's sparse KKT solve after one and after two iterations.
//   usage: batched_user_ocp_test <codegen folder> [batch] [compared instances]
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

using namespace Ungar;

#ifndef USER_NX
#define USER_NX 10
#define USER_NU 3
#define USER_NE 0
#endif
constexpr index_t NX = USER_NX, NU = USER_NU, NE = USER_NE, N = 12;
constexpr index_t NH = 2 * NU;                 // input bounds -1.5 <= u_j <= 1.5
#ifndef USER_PARAMETERS
#define USER_PARAMETERS 1  // 0: the same problem with constant data -- a ShootingProblem WITHOUT knot and instance parameters (its rows are [x | u] alone)
#endif
constexpr bool kParameters = USER_PARAMETERS != 0;
constexpr index_t KNOT = kParameters ? NX + 1 + NE : 0;  // knot parameters: reference state, input weight, right-hand sides of the equality rows
constexpr index_t INST = kParameters ? 2 : 0;            // instance parameters: step size, gain of the nonlinear term
constexpr index_t DEC = (N + 1) * NX + N * NU; // whole horizon: z = (X, U)
constexpr index_t PAR = (N + 1) * KNOT + INST + NX;  // (knot parameters of every knot, instance parameters, measured state)

static real_t Acoef(index_t i, index_t j) { return std::abs(i - j) <= 2 ? 0.4 * std::sin(1.3 * static_cast<real_t>(i) + 0.7 * static_cast<real_t>(j)) : 0.0; }
static real_t Bcoef(index_t i, index_t j) { return (i + j) % 3 == 0 ? std::cos(0.9 * static_cast<real_t>(i) + 1.7 * static_cast<real_t>(j)) : 0.0; }

/// x+ = x + dt (A x + B u + gain sin(x_i) x_(i+1)): any indexable of AD scalars.
template <class X, class U>
static VectorXad Dynamics(const X& x, const U& u, const ad_scalar_t& dt, const ad_scalar_t& gain) {
    using std::sin;
    VectorXad next{NX};
    for (index_t i = 0; i < NX; ++i) {
        ad_scalar_t rate = gain * sin(x[i]) * x[(i + 1) % NX];
        for (index_t j = 0; j < NX; ++j)
            if (Acoef(i, j) != 0.0) rate += Acoef(i, j) * x[j];
        for (index_t j = 0; j < NU; ++j)
            if (Bcoef(i, j) != 0.0) rate += Bcoef(i, j) * u[j];
        next[i] = x[i] + dt * rate;
    }
    return next;
}
template <class X, class U, class P>
static ad_scalar_t StageCost(const X& x, const U& u, const P& knot) {
    ad_scalar_t value{0.0};
    for (index_t i = 0; i < NX; ++i) value += (1.0 + 0.1 * static_cast<real_t>(i)) * (x[i] - knot[i]) * (x[i] - knot[i]);
    for (index_t j = 0; j < NU; ++j) value += knot[NX] * u[j] * u[j];
    return value;
}

struct KnotView {  // the knot parameters at `offset` of v, or the constants that stand for them
    const VectorXad& v;
    index_t offset;
    ad_scalar_t operator[](index_t e) const {
        if constexpr (kParameters) return v[offset + e];
        else return ad_scalar_t{e < NX ? 0.5 * std::sin(static_cast<real_t>(e)) : (e == NX ? 0.05 : 0.1 * static_cast<real_t>(e - NX))};
    }
};
static ad_scalar_t StepSize(const VectorXad& v, index_t offset) {
    if constexpr (kParameters) return v[offset];
    else return ad_scalar_t{0.08};
}
static ad_scalar_t Gain(const VectorXad& v, index_t offset) {
    if constexpr (kParameters) return v[offset + 1];
    else return ad_scalar_t{0.3};
}

struct Slice {  // view of a VectorXad
    const VectorXad& v;
    index_t offset;
    const ad_scalar_t& operator[](index_t i) const { return v[offset + i]; }
};

int main(int argc, char** argv) {
    const std::string folder = argc > 1 ? argv[1] : "/tmp/ungar_amd_batched_user";
    const index_t batch = argc > 2 ? std::atol(argv[2]) : 256, compared = argc > 3 ? std::atol(argv[3]) : 4;
    const std::string tag = "user_" + std::to_string(NX) + "_" + std::to_string(NU) + "_" + std::to_string(NE) + (kParameters ? "" : "_const");
    try {
        // ---- whole-horizon problem: z = (X, U), parameters = (knot parameters x (N + 1), dt, gain, measured state)
        auto knotOf = [](index_t k) { return DEC + k * KNOT; };
        const index_t instOffset = DEC + (N + 1) * KNOT, measuredOffset = instOffset + INST;
        const auto objective = [&](const VectorXad& v, VectorXad& y) {
            ad_scalar_t value{0.0};
            for (index_t k = 0; k < N; ++k) value += StageCost(Slice{v, k * NX}, Slice{v, (N + 1) * NX + k * NU}, KnotView{v, knotOf(k)});
            {  // knot N: the state part (with parameters: input weight 0 at knot N; without: the inputs of row N are zero)
                const KnotView knot{v, knotOf(N)};
                for (index_t i = 0; i < NX; ++i) value += (1.0 + 0.1 * static_cast<real_t>(i)) * (v[N * NX + i] - knot[i]) * (v[N * NX + i] - knot[i]);
            }
            y.resize(1);
            y << value;
        };
        const auto equality = [&](const VectorXad& v, VectorXad& y) {
            Autodiff::VectorComposer composer;
            for (index_t i = 0; i < NX; ++i) composer << v[i] - v[measuredOffset + i];
            for (index_t k = 0; k < N; ++k) {
                const VectorXad next = Dynamics(Slice{v, k * NX}, Slice{v, (N + 1) * NX + k * NU}, StepSize(v, instOffset), Gain(v, instOffset));
                for (index_t i = 0; i < NX; ++i) composer << v[(k + 1) * NX + i] - next[i];
            }
            for (index_t k = 0; k < N; ++k)
                for (index_t j = 0; j < NE; ++j) composer << v[(N + 1) * NX + k * NU + j] + 0.3 * v[k * NX + j] - KnotView{v, knotOf(k)}[NX + 1 + j];
            y = composer.Compose();
        };
        const auto inequality = [&](const VectorXad& v, VectorXad& y) {
            Autodiff::VectorComposer composer;
            for (index_t k = 0; k < N; ++k)
                for (index_t j = 0; j < NU; ++j) {
                    composer << v[(N + 1) * NX + k * NU + j] - 1.5;
                    composer << -v[(N + 1) * NX + k * NU + j] - 1.5;
                }
            y = composer.Compose();
        };
        auto nlp = MakeNLPProblem(Autodiff::MakeFunction({objective, DEC, PAR, tag + "_whole_obj", EnabledDerivatives::ALL, folder}, false),
                                  Autodiff::MakeFunction({equality, DEC, PAR, tag + "_whole_eqs", EnabledDerivatives::JACOBIAN, folder}, false),
                                  Autodiff::MakeFunction({inequality, DEC, PAR, tag + "_whole_ineqs", EnabledDerivatives::JACOBIAN, folder}, false));

        // ---- the same problem in stage form: row = [x | u | knot parameters | instance parameters]
        const index_t nxu = NX + NU, nPar = KNOT + INST;
        const auto stageDynamics = [&](const VectorXad& v, VectorXad& y) { y = Dynamics(Slice{v, 0}, Slice{v, NX}, StepSize(v, nxu + KNOT), Gain(v, nxu + KNOT)); };
        const auto stageCost = [&](const VectorXad& v, VectorXad& y) {
            y.resize(1);
            y << StageCost(Slice{v, 0}, Slice{v, NX}, KnotView{v, nxu});
        };
        const auto stageInequality = [&](const VectorXad& v, VectorXad& y) {
            Autodiff::VectorComposer composer;
            for (index_t j = 0; j < NU; ++j) {
                composer << v[NX + j] - 1.5;
                composer << -v[NX + j] - 1.5;
            }
            y = composer.Compose();
        };
        const auto stageEquality = [&](const VectorXad& v, VectorXad& y) {
            Autodiff::VectorComposer composer;
            for (index_t j = 0; j < NE; ++j) composer << v[NX + j] + 0.3 * v[j] - KnotView{v, nxu}[NX + 1 + j];
            y = composer.Compose();
        };
        const auto MakeProblem = [&]() {
            ShootingProblem problem;
            problem.horizon = N;
            problem.stateSize = NX;
            problem.inputSize = NU;
            problem.knotParameterSize = KNOT;
            problem.instanceParameterSize = INST;
            problem.dynamics.emplace(Autodiff::MakeFunction({stageDynamics, nxu, nPar, tag + "_stage_dyn", EnabledDerivatives::JACOBIAN, folder}, false));
            problem.cost.emplace(Autodiff::MakeFunction({stageCost, nxu, nPar, tag + "_stage_cost", EnabledDerivatives::ALL, folder}, false));
            problem.inequality.emplace(Autodiff::MakeFunction({stageInequality, nxu, nPar, tag + "_stage_ineq", EnabledDerivatives::JACOBIAN, folder}, false));
            if constexpr (NE > 0) problem.equality.emplace(Autodiff::MakeFunction({stageEquality, nxu, nPar, tag + "_stage_eq", EnabledDerivatives::JACOBIAN, folder}, false));
            return problem;
        };
        ShootingProblem problem = MakeProblem();
        const index_t nv = problem.RowSize();
        const real_t multiplier = 0.1;
        BatchedSoftSQPOptimizer batched{std::move(problem), batch, false, multiplier, 2, 100.0, 1e-2};
        std::printf("routes: riccati %d, assembly %d (2: instantiated for these sizes by the kernel factory; 3: run-time-size one-wavefront assembly)\n", batched.RiccatiRoute(),
                    batched.AssembleRoute());
        if (!std::getenv("USER_OCP_ANY_ROUTE") && (batched.RiccatiRoute() != 2 || batched.AssembleRoute() != (NE > 0 ? 2 : 3))) {  // (USER_OCP_ANY_ROUTE: soak runs over sizes that take other kernels)
            std::printf("FAIL the register-resident kernels were not taken: %s\n", ungar_last_error());
            return 1;
        }

        // ---- perturbed instances
        std::mt19937_64 rng{20261003};
        std::normal_distribution<real_t> normal{0.0, 1.0};
        std::vector<VectorXr> instances;
        for (index_t b = 0; b < batch; ++b) {
            VectorXr v{DEC + PAR};
            v.setZero();
            for (index_t i = 0; i < NX; ++i) v[measuredOffset + i] = 0.3 * normal(rng);
            if constexpr (kParameters) {
                v[instOffset] = 0.08 * (1.0 + 0.1 * normal(rng));
                v[instOffset + 1] = 0.3 * (1.0 + 0.2 * normal(rng));
            }
            for (index_t k = 0; k <= N; ++k) {
                for (index_t i = 0; i < NX; ++i) {
                    if constexpr (kParameters) v[knotOf(k) + i] = 0.5 * std::sin(0.3 * static_cast<real_t>(k) + static_cast<real_t>(i)) + 0.05 * normal(rng);
                    v[k * NX + i] = v[measuredOffset + i] + 0.05 * normal(rng);
                }
                if constexpr (kParameters) {
                    v[knotOf(k) + NX] = k < N ? 0.05 : 0.0;
                    for (index_t j = 0; j < NE; ++j) v[knotOf(k) + NX + 1 + j] = 0.2 * normal(rng);
                }
            }
            for (index_t k = 0; k < N; ++k)
                for (index_t j = 0; j < NU; ++j) v[(N + 1) * NX + k * NU + j] = (b % 3 == 0 ? 1.45 : 0.3) * normal(rng) * (b % 3 == 0 ? 0.3 : 1.0) + (b % 3 == 0 ? 1.2 : 0.0);  // every third instance near the bound
            instances.push_back(v);
        }
        std::vector<real_t> rows(static_cast<std::size_t>(batched.RowsSize())), xm(static_cast<std::size_t>(batch * NX));
        for (index_t b = 0; b < batch; ++b) {
            const VectorXr& v = instances[static_cast<std::size_t>(b)];
            for (index_t k = 0; k <= N; ++k) {
                real_t* row = rows.data() + (b * (N + 1) + k) * nv;
                for (index_t i = 0; i < NX; ++i) row[i] = v[k * NX + i];
                for (index_t j = 0; j < NU; ++j) row[NX + j] = k < N ? v[(N + 1) * NX + k * NU + j] : 0.0;
                for (index_t i = 0; i < KNOT; ++i) row[nxu + i] = v[knotOf(k) + i];
                if constexpr (kParameters) {
                    row[nxu + KNOT] = v[instOffset];
                    row[nxu + KNOT + 1] = v[instOffset + 1];
                }
            }
            for (index_t i = 0; i < NX; ++i) xm[static_cast<std::size_t>(b * NX + i)] = v[measuredOffset + i];
        }
        batched.SetRows(rows.data(), xm.data());

        real_t worstStep = 0.0, worstIterate = 0.0, worstAlpha = 0.0;
        std::vector<VectorXr> facade(static_cast<std::size_t>(compared));
        std::vector<index_t> sample;
        for (index_t s = 0; s < compared; ++s) sample.push_back(s < 3 ? s : (s * 131 + 7) % batch);
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
                SoftSQPOptimizer optimizer{false, multiplier, index_t{1}, 100.0, 1e-2};
                const VectorXr after = optimizer.Optimize(nlp, z);
                const std::vector<real_t>& d = optimizer.LastStep();
                real_t num = 0.0, den = 0.0, scaleD = 0.0, scaleX = 0.0;
                for (index_t i = 0; i < DEC; ++i) {
                    num += (after[i] - before[i]) * d[static_cast<std::size_t>(i)];
                    den += d[static_cast<std::size_t>(i)] * d[static_cast<std::size_t>(i)];
                    scaleD = std::max(scaleD, std::abs(d[static_cast<std::size_t>(i)]));
                    scaleX = std::max(scaleX, std::abs(after[i]));
                }
                const real_t alphaFacade = den > 0.0 ? num / den : 0.0;
                worstAlpha = std::max(worstAlpha, std::abs(alphaFacade - accepted[static_cast<std::size_t>(b)]));
                for (index_t k = 0; k <= N; ++k)
                    for (index_t i = 0; i < NX; ++i) {
                        worstStep = std::max(worstStep, std::abs(dZ[static_cast<std::size_t>((b * (N + 1) + k) * NX + i)] - d[static_cast<std::size_t>(k * NX + i)]) / scaleD);
                        worstIterate = std::max(worstIterate, std::abs(rows[static_cast<std::size_t>((b * (N + 1) + k) * nv + i)] - after[k * NX + i]) / scaleX);
                    }
                for (index_t k = 0; k < N; ++k)
                    for (index_t j = 0; j < NU; ++j) {
                        worstStep = std::max(worstStep, std::abs(dU[static_cast<std::size_t>((b * N + k) * NU + j)] - d[static_cast<std::size_t>((N + 1) * NX + k * NU + j)]) / scaleD);
                        worstIterate = std::max(worstIterate, std::abs(rows[static_cast<std::size_t>((b * (N + 1) + k) * nv + NX + j)] - after[(N + 1) * NX + k * NU + j]) / scaleX);
                    }
                if (std::getenv("USER_OCP_DEBUG") && s == 0) {
                    const auto qp = batched.AssembledQp();
                    const index_t nd = NX + NU;
                    for (index_t k = 0; k <= N; k += N / 2) {
                        std::printf("  k %td W_uu diag:", k);
                        for (index_t j = 0; j < NU; ++j) std::printf(" %.6g", qp.W[static_cast<std::size_t>(((b * (N + 1) + k) * nd + NX + j) * nd + NX + j)]);
                        std::printf("  w_u:");
                        for (index_t j = 0; j < NU; ++j) std::printf(" %.6g", qp.w[static_cast<std::size_t>((b * (N + 1) + k) * nd + NX + j)]);
                        std::printf("  W_xx diag:");
                        for (index_t j = 0; j < NX; ++j) std::printf(" %.4g", qp.W[static_cast<std::size_t>(((b * (N + 1) + k) * nd + j) * nd + j)]);
                        std::printf("  u:");
                        for (index_t j = 0; j < NU; ++j) std::printf(" %.6g", before[(N + 1) * NX + (k < N ? k : N - 1) * NU + j]);
                        std::printf("\n");
                    }
                    for (index_t k = 0; k <= N; k += N / 2) {
                        std::printf("  k %td dx batched:", k);
                        for (index_t i = 0; i < NX; ++i) std::printf(" %.4g", dZ[static_cast<std::size_t>((b * (N + 1) + k) * NX + i)]);
                        std::printf("\n  k %td dx facade :", k);
                        for (index_t i = 0; i < NX; ++i) std::printf(" %.4g", d[static_cast<std::size_t>(k * NX + i)]);
                        std::printf("\n");
                    }
                    for (index_t k = 0; k < N; k += N / 2) {
                        std::printf("  k %td du batched:", k);
                        for (index_t j = 0; j < NU; ++j) std::printf(" %.4g", dU[static_cast<std::size_t>((b * N + k) * NU + j)]);
                        std::printf("\n  k %td du facade :", k);
                        for (index_t j = 0; j < NU; ++j) std::printf(" %.4g", d[static_cast<std::size_t>((N + 1) * NX + k * NU + j)]);
                        std::printf("\n");
                    }
                }
                for (index_t i = 0; i < DEC; ++i) z[i] = after[i];
                std::printf("iteration %d instance %4td: step size facade %.6g batched %.6g  (|d|max %.3g)\n", iteration, b, alphaFacade, accepted[static_cast<std::size_t>(b)], scaleD);
            }
            index_t moved = 0;
            for (const real_t a : accepted) moved += a > 0.0;
            std::printf("iteration %d: %td of %td instances accepted a step; worst |d - d_facade| / |d|max = %.3e, worst |x - x_facade| / |x|max = %.3e, worst step-size difference %.3e\n",
                        iteration, moved, batch, worstStep, worstIterate, worstAlpha);
        }
        // Parameters edited ON THE DEVICE between iterations (an MPC loop moving its references): the next Iterate() must see them.  Same rows through SetRows (host
        // upload) and through DeviceRows() (device-side write after the optimiser took its parameter image of the OLD rows): identical search directions, bit for bit.
        bool staleOk = true;
        if constexpr (kParameters) {
            batched.GetRows(rows.data());
            std::vector<real_t> moved = rows;
            for (index_t b = 0; b < batch; ++b)
                for (index_t k = 0; k <= N; ++k)
                    for (index_t i = 0; i < NX; ++i) moved[static_cast<std::size_t>((b * (N + 1) + k) * nv + nxu + i)] += 0.25 * std::cos(0.1 * static_cast<real_t>(b + k + i));  // reference states
            batched.SetRows(moved.data(), xm.data());
            batched.Iterate();
            const std::vector<real_t> viaHost = batched.StateSteps();
            batched.SetRows(rows.data(), xm.data());  // parameter image of the old references
            if (ungar_device_upload(batched.DeviceRows(), moved.data(), static_cast<int64_t>(moved.size() * sizeof(real_t))) != 0) throw std::runtime_error("upload through DeviceRows failed");
            batched.Iterate();
            const std::vector<real_t> viaDevice = batched.StateSteps();
            batched.SetRows(rows.data(), xm.data());
            batched.Iterate();
            const std::vector<real_t> oldReferences = batched.StateSteps();
            staleOk = viaHost == viaDevice && viaHost != oldReferences;
            std::printf("%s parameters written through DeviceRows() reach the next iteration (and change it)\n", staleOk ? "ok:" : "FAIL");
        }
        // The values at the trial points come from ONE function stitched from the stage functions' tapes (BatchedSoftSQPOptimizer::BuildStageValues); evaluated one by
        // one instead (fuseStageValues = false) the same rows must take the same steps.
        bool fusedOk = batched.StageValuesFused();
        {
            batched.GetRows(rows.data());
            batched.SetRows(rows.data(), xm.data());
            batched.Iterate();
            const std::vector<real_t> alphaFused = batched.AcceptedStepSizes();
            std::vector<real_t> rowsFused(rows.size());
            batched.GetRows(rowsFused.data());
            BatchedSoftSQPOptimizer::fuseStageValues = false;
            BatchedSoftSQPOptimizer plain{MakeProblem(), batch, false, multiplier, 2, 100.0, 1e-2};
            BatchedSoftSQPOptimizer::fuseStageValues = true;
            plain.SetRows(rows.data(), xm.data());
            plain.Iterate();
            std::vector<real_t> rowsPlain(rows.size());
            plain.GetRows(rowsPlain.data());
            real_t worst = 0.0;
            for (std::size_t i = 0; i < rowsPlain.size(); ++i) worst = std::max(worst, std::abs(rowsPlain[i] - rowsFused[i]));
            fusedOk = fusedOk && !plain.StageValuesFused() && alphaFused == plain.AcceptedStepSizes() && worst <= 1e-12;
            std::printf("%s stage values in one launch: same step sizes, iterates within %.1e of the stage functions evaluated one by one\n", fusedOk ? "ok:" : "FAIL", worst);
        }
        const bool ok = worstStep <= 1e-9 && worstIterate <= 1e-9 && worstAlpha <= 1e-9 && staleOk && fusedOk;
        std::printf("%s batched user OCP %td + %td, %td equality rows%s (batch %td, %td compared)\n", ok ? "PASS" : "FAIL", NX, NU, NE, kParameters ? "" : ", no parameters", batch, compared);
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
}
