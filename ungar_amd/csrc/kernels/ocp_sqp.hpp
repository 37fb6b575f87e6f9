// ungar_amd :: argument blocks of the batched SQP kernels (ocp_riccati.hip), shared with the C ABI (runtime/c_api_sqp.cpp).
#pragma once

#include "ocp_riccati.hpp"

namespace ungar_amd::kernels {

/// Relaxed barrier applied to z = -h (soft_inequality_constraint.hpp:77-205): type 0 = POLY, 1 = LOG.
struct BarrierParams {
    int type;
    double stiffness, epsilon;
};

struct MeritArgs {
    int nx, nu, N, nh;
    long long batch;
    RiccatiView X;     // states x_k, k = 0..N
    RiccatiView xm;    // measured state per instance
    RiccatiView f;     // node values f(x_k, u_k), k < N
    RiccatiView cost;  // stage cost value per node (1 element), k < N; base may be null
    RiccatiView costN; // terminal cost per instance (1 element); base may be null
    RiccatiView h;     // nh inequality values per node (h <= 0 feasible); base may be null
    BarrierParams barrier;
    double violationMultiplier;  // theta = multiplier * |g|_2  (soft_sqp.hpp:79-84)
    // optional slope of the objective along the step: sum_k grad_k . [dx_k; du_k] (+ gradN . dx_N)
    RiccatiView grad, gradN, dX, dU;
    double* theta;  // out, per instance
    double* phi;    // out, per instance: objective + barrier
    double* slope;  // out, per instance (written only when grad.base and dX.base are given)
    long long xmPeriod = 0;  // > 0: `batch` counts STACKED trial points (candidate c of instance i at c * xmPeriod + i) and xm is read at instance % xmPeriod
};

constexpr int kMaxLineSearchCandidates = 16;

struct TrialArgs {
    int nx, nu, N;
    long long batch;
    RiccatiView X, U, dX, dU, Xt, Ut;
    double alpha;
    // candidates > 0: ALL trial points of the backtracking search at once -- workgroup c * batch + i writes
    // Xt(c * batch + i) = X(i) + alphas[c] dX(i) (Xt / Ut hold candidates * batch stacked instances)
    int candidates = 0;
    double alphas[kMaxLineSearchCandidates] = {};
};

/// The whole backtracking search of one instance from the merit terms of ALL its candidates (stacked: candidate c of
/// instance i at c * batch + i): the first candidate -- largest step -- that passes the acceptance test is copied into (X, U).
struct SelectArgs {
    int nx, nu, N, candidates;
    long long batch;
    double thetaMin, thetaMax, eta, gammaPhi, gammaTheta;
    double alphas[kMaxLineSearchCandidates];
    const double *theta0, *phi0, *slope, *thetaT, *phiT;
    double* accepted;  // per instance: the accepted step size, 0 = none of the candidates is acceptable
    RiccatiView X, U, Xt, Ut;
    const int* status = nullptr;  // Riccati report per instance (may be null): non-zero = QP not solved, the instance takes no step (soft_sqp.hpp:223-230 asserts there)
};

/// Acceptance test of the reference's backtracking line search (backtracking_line_search.hpp:116-151) for one candidate
/// step size, applied to every instance that has not accepted a larger one yet; accepted instances take the trial point.
struct AcceptArgs {
    int nx, nu, N;
    long long batch;
    double alpha, thetaMin, thetaMax, eta, gammaPhi, gammaTheta;
    const double *theta0, *phi0, *slope, *thetaT, *phiT;
    double* accepted;  // per instance: 0 = still searching, else the accepted step size
    RiccatiView X, U, Xt, Ut;
    const int* status = nullptr;  // as in SelectArgs
};

/// Stage data of the QP from the node kernels' outputs (soft_sqp.hpp:143-155, 247-264 restricted to one knot):
///   b_k = f_k - x_{k+1},  dx_0 = x_m - x_0,
///   W_k = hess cost_k + J_h^T diag(b''(-h)) J_h   (upper triangle of the dense (nx+nu)^2 block; the strict lower triangle is left untouched),
///   w_k = grad cost_k - J_h^T b'(-h).
struct StageQpArgs {
    int nx, nu, N, nh, hesNnz;
    long long batch;
    RiccatiView X, xm, f;         // states (k = 0..N), measured state, node values (k < N)
    RiccatiView costGrad;         // dense 1 x (nx+nu) gradient per node
    RiccatiView costHes;          // hesNnz upper-triangular Hessian values per node, pattern below
    RiccatiView h, hJac;          // nh inequality values and the dense nh x (nx+nu) Jacobian per node (base null: none)
    BarrierParams barrier;
    RiccatiView b, hess, grad, dx0;  // outputs
    unsigned char hesRow[160], hesCol[160];
};

}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_ocp_stage_qp(const ungar_amd::kernels::StageQpArgs* a, void* stream);
extern "C" int ungar_amd_launch_riccati(const ungar_amd::kernels::RiccatiArgs* a, void* stream);
extern "C" int ungar_amd_riccati_route(int nx, int nu, int ne, int prepare);  // ocp_riccati_wave.hip: 0 LDS-resident, 1 register-resident (compiled in), 2 register-resident (kernel factory)
extern "C" int ungar_amd_launch_ocp_merit(const ungar_amd::kernels::MeritArgs* a, void* stream);
extern "C" int ungar_amd_launch_ocp_trial(const ungar_amd::kernels::TrialArgs* a, void* stream);
extern "C" int ungar_amd_launch_ocp_accept(const ungar_amd::kernels::AcceptArgs* a, void* stream);
extern "C" int ungar_amd_launch_ocp_select(const ungar_amd::kernels::SelectArgs* a, void* stream);
