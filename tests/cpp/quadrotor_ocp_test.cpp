// GPU test: the whole-horizon quadrotor OCP of the reference (example/mpc/quadrotor.example.cpp
// PART I-III: objective :196-244, equality constraints :246-266, inequality constraints :268-291)
// written against ungar_amd's facade and evaluated through Ungar::Autodiff::Function on the MI355X.
// It dumps inputs, values and sparse derivatives; tests/test_whole_horizon.py then checks
//   * the block-bidiagonal structure of the equality Jacobian against the per-node kernel
//     (SURVEY.md Appendix A:  d/dx_{k+1} = I,  d/dx_k = -A_k,  d/du_k = -B_k),
//   * objective value / gradient / upper-triangular Hessian against an independent torch model.
#include <cstdio>
#include <fstream>
#include <string>

#include "ungar/autodiff/function.hpp"
#include "ungar/autodiff/vector_composer.hpp"
#include "ungar/variable_map.hpp"

using namespace Ungar;

constexpr auto N = 30_c;
constexpr auto ROTORS = 4_c;
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

static VectorXad Dynamics(const VectorXad& xk, const VectorXad& uk, const VectorXad& par) {
    const auto x_ = MakeVariableLazyMap(xk, x);
    const auto u_ = MakeVariableLazyMap(uk, u);
    const auto p_ = MakeVariableLazyMap(par, parameters);
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

static void DumpSparse(std::ofstream& out, const char* tag, const Autodiff::SparseMatrix& A) {
    out << tag << " " << A.rows() << " " << A.cols() << " " << A.nonZeros() << "\n";
    out.precision(17);
    for (index_t r = 0; r < A.rows(); ++r)
        for (int k = A.outerIndexPtr()[r]; k < A.outerIndexPtr()[r + 1]; ++k) out << r << " " << A.innerIndexPtr()[k] << " " << A.valuePtr()[k] << "\n";
}
static void DumpVector(std::ofstream& out, const char* tag, const VectorXr& v) {
    out << tag << " " << v.size() << "\n";
    out.precision(17);
    for (index_t i = 0; i < v.size(); ++i) out << v[i] << "\n";
}

int main(int argc, char** argv) {
    const std::string folder = argc > 1 ? argv[1] : "/tmp/ungar_amd_ocp_test";
    const std::string dumpPath = argc > 2 ? argv[2] : folder + "/quadrotor_ocp.txt";
    try {
        const auto objective = [&](const VectorXad& v, VectorXad& y) {
            const auto v_ = MakeVariableLazyMap(v, variables);
            ad_scalar_t value{0.0};
            for (const auto k : enumerate(N + 1_step)) {
                const auto p = v_.Get(position, k);
                const auto q = v_.Get(orientation, k);
                const auto pDot = v_.Get(linear_velocity, k);
                const auto bOmega = v_.Get(b_angular_velocity, k);
                value += (p - v_.Get(reference_position, k)).squaredNorm() +
                         Utils::Min((q.coeffs() - v_.Get(reference_orientation, k).coeffs()).squaredNorm(),
                                    (q.coeffs() + v_.Get(reference_orientation, k).coeffs()).squaredNorm()) +
                         (pDot - v_.Get(reference_linear_velocity, k)).squaredNorm() +
                         (bOmega - v_.Get(b_reference_angular_velocity, k)).squaredNorm();
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
            for (const auto k : enumerate(N)) composer << v_.Get(x, k + 1_step) - Dynamics(v_.Get(x, k), v_.Get(u, k), v_.Get(parameters));
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
        Autodiff::Function::Blueprint objBp{objective, decision_variables.Size(), parameters.Size(), "quadrotor_mpc_obj", EnabledDerivatives::ALL, folder};
        Autodiff::Function::Blueprint eqBp{equality, decision_variables.Size(), parameters.Size(), "quadrotor_mpc_eqs", EnabledDerivatives::JACOBIAN, folder};
        Autodiff::Function::Blueprint ineqBp{inequality, decision_variables.Size(), parameters.Size(), "quadrotor_mpc_ineqs", EnabledDerivatives::JACOBIAN, folder};
        std::printf("sizes obj %td eq %td ineq %td (dec %td par %td)\n", objBp.dependentVariableSize, eqBp.dependentVariableSize,
                    ineqBp.dependentVariableSize, decision_variables.Size(), parameters.Size());
        if (objBp.dependentVariableSize != 1 || eqBp.dependentVariableSize != 403 || ineqBp.dependentVariableSize != 240) return 3;
        Autodiff::Function obj = Autodiff::MakeFunction(objBp, true), eq = Autodiff::MakeFunction(eqBp, true), ineq = Autodiff::MakeFunction(ineqBp, true);

        // reference parameter values (quadrotor.example.cpp:326-358) and a deterministic non-trivial trajectory
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
        v_.Get(max_rotor_speed) = 1e2;
        v_.Get(measured_position) = Vector3r(0.0, 0.0, 4.0);
        v_.Get(measured_orientation).setIdentity();
        const real_t hover = std::sqrt(1.5 * 9.80665 / 0.015 / 4.0);
        for (const auto k : enumerate(N + 1_step)) {
            const real_t t = static_cast<real_t>(k) / static_cast<real_t>(N);
            v_.Get(position, k) = Vector3r(0.3 * t, -0.2 * t * t, 4.0 + 0.1 * std::sin(3.0 * t));
            v_.Get(orientation, k) = Quaternionr(1.0, 0.1 * t, -0.05 * t, 0.2 * t).normalized();
            v_.Get(linear_velocity, k) = Vector3r(0.3, -0.4 * t, 0.3 * std::cos(3.0 * t));
            v_.Get(b_angular_velocity, k) = Vector3r(0.1, -0.2 * t, 0.3 * t);
            v_.Get(reference_position, k) = Vector3r(0.5 * t, 0.0, 4.0);
            v_.Get(reference_orientation, k) = Quaternionr(-1.0, 0.0, 0.0, 0.05 * t).normalized();  // other hemisphere: exercises Min
            v_.Get(reference_linear_velocity, k) = Vector3r(0.5, 0.0, 0.0);
            v_.Get(b_reference_angular_velocity, k).setZero();
        }
        for (const auto k : enumerate(N))
            for (const auto i : enumerate(ROTORS)) v_.Get(rotor_speed, k, i) = hover * (1.0 + 0.02 * static_cast<real_t>(i) - 0.01 * static_cast<real_t>(k % 3));
        const VectorXr& in = v_.Get();

        std::ofstream out(dumpPath);
        DumpVector(out, "INPUT", in);
        DumpVector(out, "OBJ", obj(in));
        DumpSparse(out, "OBJ_JAC", obj.Jacobian(in));
        DumpSparse(out, "OBJ_HES", obj.Hessian(in));
        DumpVector(out, "EQ", eq(in));
        DumpSparse(out, "EQ_JAC", eq.Jacobian(in));
        DumpVector(out, "INEQ", ineq(in));
        DumpSparse(out, "INEQ_JAC", ineq.Jacobian(in));
        std::printf("eq jac nnz %td, obj hes nnz %td, ineq jac nnz %td\n", eq.Jacobian(in).nonZeros(), obj.Hessian(in).nonZeros(), ineq.Jacobian(in).nonZeros());
        std::printf("DUMPED %s\n", dumpPath.c_str());
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    return 0;
}
