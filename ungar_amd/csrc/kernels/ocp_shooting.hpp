// ungar_amd :: argument blocks of the batched SQP kernels for shooting problems WITH CARRIED QUANTITIES AND STAGE EQUALITY ROWS
// (ocp_shooting.hip), shared with the C ABI (runtime/c_api_sqp.cpp).
//
// The reference's three MPC examples are not stage-separable as written: the objective of the quadrotor / RC car couples u_k with
// u_{k-1} (example/mpc/quadrotor.example.cpp:222-227, rc_car.example.cpp:216-220) and the quadruped's foot-contact equality rows
// couple the foot positions of knots k and k-1 (quadruped.example.cpp:279-304).  Both become stage-local once the quantity of the
// previous knot is CARRIED in the stage state:  z_k = [c_k; x_k],  c_{k+1} = kappa(x_k, u_k)  (kappa = u_k, or the foot positions).
// The carried part is not a decision variable of the reference's problem -- it has no defect, no regularisation, and its increment
// is the linearisation of kappa -- so the QP below has the same solution as the QP the reference hands to OSQP (soft_sqp.hpp:143-158).
//
// Every node (instance b, knot k <= N) owns one ROW of `nv` doubles  [c (nc) | x (nx) | u (nu) | w (nw) | p (np)]  in a node-major
// array; a stage function is an Ungar::Autodiff::Function over a contiguous slice of that row (dynamics / carry over [x|u ; w|p],
// cost / equality / inequality rows over [c|x|u ; w|p]), evaluated for all nodes by ungar_function_*_nodes.  Row N holds x_N (its
// input slots are not decision variables; the stage cost there is the terminal cost, its input derivatives are ignored).
#pragma once

#include "ocp_sqp.hpp"

namespace ungar_amd::kernels {

/// Coordinate pattern of a sparse stage derivative (DEVICE arrays, canonical CSR order as ungar_function_*_sparsity reports it).
struct StagePattern {
    const int* rows;
    const int* cols;
    int nnz;
};

struct ShootingDims {
    int nx, nu, nc, nw, np, N;
    int carryInputs;  // 1: c_{k+1} = u_k (nc == nu), no carry function
    long long batch;
    UNGAR_HD int nz() const { return nc + nx; }
    UNGAR_HD int nd() const { return nc + nx + nu; }  // differentiated part of a row
    UNGAR_HD int nv() const { return nc + nx + nu + nw + np; }
};

/// QP data of every node from the stage functions' sparse outputs (node-major, N + 1 knots per instance, values in pattern order):
///   [A|B]_k  nz x nd : rows 0..nc-1 the carry Jacobian (or the identity on u), rows nc.. the dynamics Jacobian; the columns of c are zero
///   b_k      nz      : [0; f_k - x_{k+1}]
///   W_k      nd x nd : upper triangle of  hess cost_k + J_h^T diag(b''(-h)) J_h + reg I_(x,u)   (k = N: cost only; the z block is the terminal Hessian)
///   w_k      nd      : grad cost_k - J_h^T b'(-h)                                                 (soft_sqp.hpp:143-155, 247-264)
///   E_k      ne x nd : dense equality-row Jacobian;   dz0 = [0; x_m - x_0]
struct ShootingAssembleArgs {
    ShootingDims d;
    const double* rows;  // node rows
    const double* xm;    // batch x nx
    const double *f, *fJ, *cJ, *l, *lg, *lH, *h, *hJ, *eJ;  // stage outputs (null where the function is absent)
    StagePattern pf, pc, pg, pH, ph, pe;
    int nh, ne;
    BarrierParams barrier;
    double regularization;
    double *AB, *b, *W, *w, *E, *dz0;  // outputs: AB / b / E hold N knots per instance, W / w hold N + 1
    // Elimination of the stage equality rows BEFORE the recursion (eliminate != 0), node by node in parallel: Gauss-Jordan on
    // [D | C | e] with the largest input coefficient of a row as its pivot expresses one input per active row through the state and
    // the remaining inputs, u_j = -(g . [z; u] + g0); substituting it into the stage cost and the linearised dynamics leaves an
    // UNCONSTRAINED stage problem in which u_j is a decoupled dummy (unit Hessian, zero gradient, zero column of [A|B]).  The Riccati
    // recursion then runs without equality rows -- its sequential chain no longer carries the (nu + ne)-order KKT block of every knot --
    // and ShootingRecoverKernel restores the eliminated inputs from the reduced rows afterwards.  Rows that are identically zero take
    // no pivot; a row without an input part but with a state part or a residual cannot be met by the inputs of its knot: reported.
    int eliminate;
    const double* e;   // equality values (the equality function's output), N + 1 knots per instance
    double* er;        // out: residuals of the reduced rows, N knots per instance x ne
    int* pivots;       // out: pivot input of every row (-1: none, -2: cannot be met), N knots per instance x ne
};

/// du_j <- -(E'_i . [dz; du] + e'_i) for every reduced row i with pivot input j (after the Riccati solve of the eliminated problem);
/// a row reported as unsatisfiable marks its instance in `status` (-(k + 1)).
struct ShootingRecoverArgs {
    ShootingDims d;
    int ne;
    const double *E, *er;
    const int* pivots;
    const double* dZ;
    double* dU;
    int* status;
};

/// Merit terms per (stacked) instance (soft_sqp.hpp:68-87):  theta = c |[x_0 - x_m; x_{k+1} - f_k; e_k]|_2,  objective = sum_{k <= N} l_k,
/// phi = objective + sum_{k < N} barrier(-h_k),  slope = sum_k grad l_k . [dz_k; du_k].  `period` > 0: `batch` counts stacked trial
/// points, candidate c of instance i at c * period + i; x_m and (dZ, dU) belong to instance index % period.
struct ShootingMeritArgs {
    ShootingDims d;
    const double* rows;
    const double* xm;
    const double *f, *l, *h, *e;
    int nh, ne;
    BarrierParams barrier;
    double violationMultiplier;
    const double* lg;  // sparse cost gradient (pattern pg), null: no slope
    StagePattern pg;
    const double *dZ, *dU;
    double *theta, *phi, *objective, *slope;
    long long period;
    long long rowsStride = 0;  // 0: node-major rows; > 0: UNIT-FASTEST rows, element e of node i = instance * (N + 1) + knot at rows[e * rowsStride + i]
    const int* instances = nullptr;  // period > 0: stacked point s belongs to instance instances[s % period] (a LISTED subset of the instances; null: s % period)
    long long valueStride = 0;       // 0: f, l, h, e dense per node (nx / 1 / nh / ne); > 0: all four point into one array of valueStride doubles per node
};

/// Stacked trial rows: row (c * batch + b, k) = row (b, k) with [c|x|u] += alphas[c] * [dZ; dU]  (k = N: z only), parameters copied;
/// with carryInputs the carried slots of row k + 1 are the trial inputs of row k (the carry function otherwise refreshes them).
struct ShootingTrialArgs {
    ShootingDims d;
    const double* rows;
    const double *dZ, *dU;
    double* trial;
    int candidates;
    double alphas[kMaxLineSearchCandidates];
    // > 0: the trial rows are written UNIT-FASTEST (element e of stacked node i at trial[e * trialStride + i]): the stage functions then read
    // them with coalesced loads and touch only the elements they use (a node-major row is fetched whole, line by line, by every function)
    long long trialStride = 0;
    // listed > 0: only the instances instances[0 .. listed) are stacked -- trial point (c, i) is instance instances[i], at c * listed + i
    const int* instances = nullptr;
    long long listed = 0;
    // Unit-fastest image only: the row elements [first, first + elements) are written, at trial[(e - first) * trialStride + i] (elements = 0: the whole row).
    // [0, nd): the variables -- all a candidate step changes; [nd, nv) with one candidate of length 0: the knot and instance parameters, once per SetRows.
    int first = 0, elements = 0;
};

/// The backtracking search of backtracking_line_search.hpp:116-151 per instance over the stacked candidates, then the bookkeeping of
/// SoftSQPOptimizer::Optimize (soft_sqp.hpp:88-99): an instance without an acceptable step, or whose objective decreased by less than
/// 1e-6, stops iterating (active = 0); inactive instances are left untouched.
struct ShootingSelectArgs {
    ShootingDims d;
    int candidates;
    double thetaMin, thetaMax, eta, gammaPhi, gammaTheta;
    double alphas[kMaxLineSearchCandidates];
    const double *theta0, *phi0, *objective0, *slope, *thetaT, *phiT, *objectiveT;
    double* accepted;
    int* active;
    // Staged search: the candidates may be offered in several calls (the largest steps first, the rest only if some instance still needs
    // them).  first: this call opens the search of the iteration; last: it closes it (an instance without an acceptable step stops iterating);
    // a call that is not the last one counts the instances it leaves unresolved in *unresolved instead.
    int first, last;
    int* unresolved;
    long long trialStride;  // layout of `trial` (ShootingTrialArgs)
    const int* status;  // per instance, from the Riccati solve: non-zero = the QP was not solved (the reference asserts there, soft_sqp.hpp:223-230): no step, instance stops
    double* rows;
    const double* trial;
    // listed > 0: the stacked candidates belong to the instances instances[0 .. listed) (ShootingTrialArgs); nextInstances (may be null): the instances this
    // call leaves unresolved are appended there, at the positions the counter *unresolved hands out (any order: the instances are independent)
    const int* instances = nullptr;
    long long listed = 0;
    int* nextInstances = nullptr;
};

}  // namespace ungar_amd::kernels

extern "C" int ungar_amd_launch_shooting_assemble(const ungar_amd::kernels::ShootingAssembleArgs* a, void* stream);
extern "C" int ungar_amd_shooting_assemble_route(int nz, int nu, int ne, int nh, int prepare);  // ocp_shooting.hip
extern "C" int ungar_amd_launch_shooting_recover(const ungar_amd::kernels::ShootingRecoverArgs* a, void* stream);
extern "C" int ungar_amd_launch_shooting_refresh_carried_inputs(const ungar_amd::kernels::ShootingDims* d, double* rows, void* stream);
extern "C" int ungar_amd_launch_shooting_merit(const ungar_amd::kernels::ShootingMeritArgs* a, void* stream);
extern "C" int ungar_amd_launch_shooting_trial(const ungar_amd::kernels::ShootingTrialArgs* a, void* stream);
extern "C" int ungar_amd_launch_shooting_select(const ungar_amd::kernels::ShootingSelectArgs* a, void* stream);
