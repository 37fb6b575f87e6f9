// ungar_amd :: C ABI of the batched SQP iteration (include/ungar_amd.h, "batched SQP iteration"): argument checks and launches.
#include "measurement.hpp"
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>

#include "../../../include/ungar_amd.h"
#include "../kernels/ocp_shooting.hpp"

namespace ungar_amd::runtime {
int Fail(int code, const std::string& msg);  // c_api.cpp
}

namespace {

using namespace ungar_amd::kernels;
using ungar_amd::runtime::Fail;

RiccatiView View(const ungar_operand& o) {
    return {o.base, o.instance_stride, o.knot_stride, o.element_stride};
}
RiccatiView ViewOrNull(const ungar_operand* o) {
    return o ? View(*o) : RiccatiView{nullptr, 0, 0, 0};
}
BarrierParams Barrier(const ungar_barrier& b) {
    return {b.type, b.stiffness, b.epsilon};
}
bool BadDims(int64_t nx, int64_t nu, int64_t N, int64_t batch) {
    // grids are 32-bit: batch * (N + 1) workgroups (x 16 stacked candidates) must stay below 2^31, and the (row, column) bytes of the
    // stage-QP pattern need nx + nu <= 256
    return nx < 1 || nu < 1 || N < 1 || batch < 0 || nx > 255 || nu > 255 || nx + nu > 256 || N > (1 << 20) || batch * (N + 1) > (INT32_MAX >> 4);
}
bool ToDims(const ungar_shooting_dims& d, ShootingDims* out) {
    if (d.nx < 1 || d.nu < 1 || d.nc < 0 || d.nw < 0 || d.np < 0 || d.horizon < 1 || d.batch < 0 || d.nc + d.nx + d.nu > 256 || d.nw + d.np > (1 << 20) ||
        d.horizon > (1 << 20) || d.batch * (d.horizon + 1) > (INT32_MAX >> 4) || (d.carry_inputs && d.nc != d.nu))
        return false;
    *out = ShootingDims{static_cast<int>(d.nx), static_cast<int>(d.nu), static_cast<int>(d.nc), static_cast<int>(d.nw), static_cast<int>(d.np), static_cast<int>(d.horizon),
                        d.carry_inputs ? 1 : 0, d.batch};
    return true;
}
StagePattern Pattern(const ungar_stage_pattern& p) {
    return {p.rows, p.cols, static_cast<int>(p.nnz)};
}
bool BadPattern(const ungar_stage_pattern& p, const double* values) {
    return p.nnz < 0 || p.nnz > (1 << 20) || (p.nnz > 0 && (!p.rows || !p.cols || !values));
}
int Launched(int err, const char* what) {
    if (err != 0) return Fail(UNGAR_E_HIP, std::string(what) + ": " + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}

}  // namespace

extern "C" {

int ungar_ocp_stage_qp(const ungar_ocp_stage_qp_args* a, void* stream) {
    if (!a || BadDims(a->nx, a->nu, a->horizon, a->batch)) return Fail(UNGAR_E_INVALID, "ungar_ocp_stage_qp: bad dimensions");
    if (a->batch == 0) return UNGAR_OK;
    if (!a->X.base || !a->xm.base || !a->f.base || !a->cost_grad.base || !a->b.base || !a->hess.base || !a->grad.base || !a->dx0.base)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_stage_qp: null operand base");
    if (a->hes_nnz < 0 || a->hes_nnz > 160 || (a->hes_nnz > 0 && (!a->hes_rows || !a->hes_cols || !a->cost_hes.base)))
        return Fail(UNGAR_E_INVALID, "ungar_ocp_stage_qp: bad cost Hessian pattern (at most 160 entries)");
    if (a->nh < 0 || a->nh > 64 || (a->nh > 0 && (!a->h.base || !a->h_jac.base))) return Fail(UNGAR_E_INVALID, "ungar_ocp_stage_qp: bad inequality block (at most 64 rows)");
    StageQpArgs k{};
    k.nx = static_cast<int>(a->nx);
    k.nu = static_cast<int>(a->nu);
    k.N = static_cast<int>(a->horizon);
    k.nh = static_cast<int>(a->nh);
    k.hesNnz = static_cast<int>(a->hes_nnz);
    k.batch = a->batch;
    k.X = View(a->X);
    k.xm = View(a->xm);
    k.f = View(a->f);
    k.costGrad = View(a->cost_grad);
    k.costHes = View(a->cost_hes);
    k.h = a->nh > 0 ? View(a->h) : RiccatiView{nullptr, 0, 0, 0};
    k.hJac = a->nh > 0 ? View(a->h_jac) : RiccatiView{nullptr, 0, 0, 0};
    k.barrier = Barrier(a->barrier);
    k.b = View(a->b);
    k.hess = View(a->hess);
    k.grad = View(a->grad);
    k.dx0 = View(a->dx0);
    const int64_t n = a->nx + a->nu;
    for (int64_t e = 0; e < a->hes_nnz; ++e) {
        if (a->hes_rows[e] < 0 || a->hes_cols[e] < a->hes_rows[e] || a->hes_cols[e] >= n) return Fail(UNGAR_E_INVALID, "ungar_ocp_stage_qp: Hessian pattern must be upper triangular");
        k.hesRow[e] = static_cast<unsigned char>(a->hes_rows[e]);
        k.hesCol[e] = static_cast<unsigned char>(a->hes_cols[e]);
    }
    return Launched(ungar_amd_launch_ocp_stage_qp(&k, stream), "ungar_ocp_stage_qp");
}

int64_t ungar_ocp_riccati_workspace(int64_t nx, int64_t nu, int64_t horizon, int64_t batch) {
    if (BadDims(nx, nu, horizon, batch)) return -1;
    return batch * horizon * nu * (nx + 1);
}

int ungar_ocp_riccati_route(int64_t nx, int64_t nu, int64_t ne, int32_t prepare) {
    if (nx < 1 || nu < 1 || ne < 0 || nx + nu > 256) return 0;
    return ungar_amd_riccati_route(static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(ne), prepare);
}

int ungar_ocp_riccati_solve(const ungar_ocp_qp* q, void* stream) {
    if (!q || BadDims(q->nx, q->nu, q->horizon, q->batch)) return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: bad dimensions");
    if (q->batch == 0) return UNGAR_OK;
    if (!q->jac.base || !q->b.base || !q->hess.base || !q->grad.base || !q->dx0.base || !q->dX.base || !q->dU.base)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: null operand base");
    if (!q->workspace || q->workspace_doubles < ungar_ocp_riccati_workspace(q->nx, q->nu, q->horizon, q->batch))
        return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: workspace too small (see ungar_ocp_riccati_workspace)");
    if (q->ne < 0 || q->ne > 64 || (q->ne > 0 && (!q->eq.base || !q->eq_values.base)) || q->hess_terminal_ld < 0 || (q->hess_terminal_ld > 0 && q->hess_terminal_ld < q->nx))
        return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: bad equality block (at most 64 rows) or terminal leading dimension");
    if (static_cast<std::size_t>(RiccatiScratchDoubles(static_cast<int>(q->nx), static_cast<int>(q->nu), static_cast<int>(q->ne))) * sizeof(double) > 160 * 1024)
        return Fail(UNGAR_E_UNSUPPORTED, "ungar_ocp_riccati_solve: nx + nu too large for the LDS-resident recursion (160 KiB per workgroup)");
    RiccatiArgs k{static_cast<int>(q->nx), static_cast<int>(q->nu), static_cast<int>(q->horizon), q->batch, View(q->jac), View(q->b), View(q->hess), View(q->grad),
                  View(q->hess_terminal), View(q->grad_terminal), View(q->dx0), View(q->dX), View(q->dU), q->workspace, q->regularization, q->status};
    k.ne = static_cast<int>(q->ne);
    if (q->ne > 0) {
        k.eq = View(q->eq);
        k.eqv = View(q->eq_values);
    }
    k.hessNld = static_cast<int>(q->hess_terminal_ld);
    return Launched(ungar_amd_launch_riccati(&k, stream), "ungar_ocp_riccati_solve");
}

int ungar_ocp_merit(const ungar_ocp_merit_args* a, void* stream) { return ungar_ocp_merit_stacked(a, 0, stream); }

int ungar_ocp_merit_stacked(const ungar_ocp_merit_args* a, int64_t period, void* stream) {
    if (!a || BadDims(a->nx, a->nu, a->horizon, a->batch) || a->nh < 0 || period < 0 || (period > 0 && a->batch % period != 0))
        return Fail(UNGAR_E_INVALID, "ungar_ocp_merit: bad dimensions");
    if (a->batch == 0) return UNGAR_OK;
    if (!a->X.base || !a->xm.base || !a->f.base || !a->theta || !a->phi) return Fail(UNGAR_E_INVALID, "ungar_ocp_merit: null operand base");
    MeritArgs k{static_cast<int>(a->nx), static_cast<int>(a->nu), static_cast<int>(a->horizon), static_cast<int>(a->nh), a->batch, View(a->X), View(a->xm), View(a->f),
                      View(a->cost), View(a->cost_terminal), a->nh > 0 ? View(a->h) : RiccatiView{nullptr, 0, 0, 0}, Barrier(a->barrier), a->violation_multiplier,
                      View(a->cost_grad), View(a->cost_grad_terminal), View(a->dX), View(a->dU), a->theta, a->phi, a->slope};
    k.xmPeriod = period;
    return Launched(ungar_amd_launch_ocp_merit(&k, stream), "ungar_ocp_merit");
}

int ungar_ocp_trial_points(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_operand* X, const ungar_operand* U, const ungar_operand* dX,
                           const ungar_operand* dU, const double* alphas, int64_t candidates, const ungar_operand* Xt, const ungar_operand* Ut, void* stream) {
    if (BadDims(nx, nu, horizon, batch) || !X || !U || !dX || !dU || !Xt || !Ut || !alphas || candidates < 1 || candidates > kMaxLineSearchCandidates)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_trial_points: bad argument (1 <= candidates <= 16)");
    if (batch == 0) return UNGAR_OK;
    if (!X->base || !U->base || !dX->base || !dU->base || !Xt->base || !Ut->base) return Fail(UNGAR_E_INVALID, "ungar_ocp_trial_points: null operand base");
    TrialArgs k{static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(horizon), batch, View(*X), View(*U), View(*dX), View(*dU), View(*Xt), View(*Ut), 0.0};
    k.candidates = static_cast<int>(candidates);
    for (int64_t c = 0; c < candidates; ++c) k.alphas[c] = alphas[c];
    return Launched(ungar_amd_launch_ocp_trial(&k, stream), "ungar_ocp_trial_points");
}

int ungar_ocp_line_search_select(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_line_search_parameters* p, const double* alphas, int64_t candidates,
                                 const double* theta0, const double* phi0, const double* slope, const double* theta_trial, const double* phi_trial, double* accepted,
                                 const ungar_operand* X, const ungar_operand* U, const ungar_operand* Xt, const ungar_operand* Ut, const int32_t* status, void* stream) {
    if (BadDims(nx, nu, horizon, batch) || !p || !alphas || candidates < 1 || candidates > kMaxLineSearchCandidates || !theta0 || !phi0 || !slope || !theta_trial || !phi_trial ||
        !accepted || !X || !U || !Xt || !Ut)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_select: bad argument (1 <= candidates <= 16)");
    if (batch == 0) return UNGAR_OK;
    if (!X->base || !U->base || !Xt->base || !Ut->base) return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_select: null operand base");
    SelectArgs k{static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(horizon), static_cast<int>(candidates), batch, p->theta_min, p->theta_max, p->eta, p->gamma_phi,
                 p->gamma_theta, {}, theta0, phi0, slope, theta_trial, phi_trial, accepted, View(*X), View(*U), View(*Xt), View(*Ut)};
    for (int64_t c = 0; c < candidates; ++c) k.alphas[c] = alphas[c];
    k.status = status;
    return Launched(ungar_amd_launch_ocp_select(&k, stream), "ungar_ocp_line_search_select");
}

int ungar_ocp_trial_point(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_operand* X, const ungar_operand* U, const ungar_operand* dX,
                          const ungar_operand* dU, double alpha, const ungar_operand* Xt, const ungar_operand* Ut, void* stream) {
    if (BadDims(nx, nu, horizon, batch) || !X || !U || !dX || !dU || !Xt || !Ut) return Fail(UNGAR_E_INVALID, "ungar_ocp_trial_point: bad argument");
    if (batch == 0) return UNGAR_OK;
    if (!X->base || !U->base || !dX->base || !dU->base || !Xt->base || !Ut->base) return Fail(UNGAR_E_INVALID, "ungar_ocp_trial_point: null operand base");
    const TrialArgs k{static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(horizon), batch, View(*X), View(*U), View(*dX), View(*dU), View(*Xt), View(*Ut), alpha};
    return Launched(ungar_amd_launch_ocp_trial(&k, stream), "ungar_ocp_trial_point");
}

int ungar_ocp_line_search_accept(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_line_search_parameters* p, double alpha, const double* theta0,
                                 const double* phi0, const double* slope, const double* theta_trial, const double* phi_trial, double* accepted, const ungar_operand* X,
                                 const ungar_operand* U, const ungar_operand* Xt, const ungar_operand* Ut, const int32_t* status, void* stream) {
    if (BadDims(nx, nu, horizon, batch) || !p || !theta0 || !phi0 || !slope || !theta_trial || !phi_trial || !accepted || !X || !U || !Xt || !Ut)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_accept: bad argument");
    if (batch == 0) return UNGAR_OK;
    if (!X->base || !U->base || !Xt->base || !Ut->base) return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_accept: null operand base");
    AcceptArgs k{static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(horizon), batch, alpha, p->theta_min, p->theta_max, p->eta, p->gamma_phi, p->gamma_theta,
                 theta0, phi0, slope, theta_trial, phi_trial, accepted, View(*X), View(*U), View(*Xt), View(*Ut)};
    k.status = status;
    return Launched(ungar_amd_launch_ocp_accept(&k, stream), "ungar_ocp_line_search_accept");
}

int ungar_shooting_assemble(const ungar_shooting_assemble_args* a, void* stream) {
    ShootingAssembleArgs k{};
    if (!a || !ToDims(a->dims, &k.d)) return Fail(UNGAR_E_INVALID, "ungar_shooting_assemble: bad dimensions");
    if (k.d.batch == 0) return UNGAR_OK;
    if (!a->rows || !a->xm || !a->f || !a->AB || !a->b || !a->W || !a->w || !a->dz0) return Fail(UNGAR_E_INVALID, "ungar_shooting_assemble: null argument");
    if (a->nh < 0 || a->nh > 256 || a->ne < 0 || a->ne > 64 || (a->nh > 0 && !a->h) || (a->ne > 0 && !a->E)) return Fail(UNGAR_E_INVALID, "ungar_shooting_assemble: bad row counts");
    if (BadPattern(a->f_pattern, a->f_jac) || BadPattern(a->cost_grad_pattern, a->cost_grad) || BadPattern(a->cost_hes_pattern, a->cost_hes) ||
        (a->nh > 0 && BadPattern(a->h_pattern, a->h_jac)) || (a->ne > 0 && BadPattern(a->eq_pattern, a->eq_jac)) || (!k.d.carryInputs && k.d.nc > 0 && BadPattern(a->carry_pattern, a->carry_jac)))
        return Fail(UNGAR_E_INVALID, "ungar_shooting_assemble: bad sparsity pattern");
    k.rows = a->rows;
    k.xm = a->xm;
    k.f = a->f;
    k.fJ = a->f_jac;
    k.cJ = a->carry_jac;
    k.lg = a->cost_grad;
    k.lH = a->cost_hes;
    k.h = a->nh > 0 ? a->h : nullptr;
    k.hJ = a->nh > 0 ? a->h_jac : nullptr;
    k.eJ = a->ne > 0 ? a->eq_jac : nullptr;
    k.pf = Pattern(a->f_pattern);
    k.pc = !k.d.carryInputs && k.d.nc > 0 ? Pattern(a->carry_pattern) : StagePattern{nullptr, nullptr, 0};
    k.pg = Pattern(a->cost_grad_pattern);
    k.pH = Pattern(a->cost_hes_pattern);
    k.ph = a->nh > 0 ? Pattern(a->h_pattern) : StagePattern{nullptr, nullptr, 0};
    k.pe = a->ne > 0 ? Pattern(a->eq_pattern) : StagePattern{nullptr, nullptr, 0};
    k.nh = static_cast<int>(a->nh);
    k.ne = static_cast<int>(a->ne);
    k.barrier = Barrier(a->barrier);
    k.regularization = a->regularization;
    k.AB = a->AB;
    k.b = a->b;
    k.W = a->W;
    k.w = a->w;
    k.E = a->E;
    k.dz0 = a->dz0;
    k.eliminate = a->eliminate_equalities && a->ne > 0 ? (UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ASSEMBLE_SKIP_SUBSTITUTION") ? 2 : 1) : 0;
    if (k.eliminate && UNGAR_MEASUREMENT_SWITCH("UNGAR_AMD_ASSEMBLE_GENERIC")) k.eliminate |= 4;  // measurement switch: no wavefront-specialised sections (same bits)
    k.e = a->eq;
    k.er = a->eq_reduced;
    k.pivots = a->eq_pivots;
    if (k.eliminate && (!a->eq || !a->eq_reduced || !a->eq_pivots)) return Fail(UNGAR_E_INVALID, "ungar_shooting_assemble: eliminating the equality rows needs eq, eq_reduced and eq_pivots");
    return Launched(ungar_amd_launch_shooting_assemble(&k, stream), "ungar_shooting_assemble");
}

int ungar_shooting_assemble_route(int64_t nz, int64_t nu, int64_t ne, int64_t nh, int32_t prepare) {
    if (nz < 1 || nu < 1 || ne < 0 || nh < 0 || nz + nu > 256 || ne > 64) return 0;
    return ungar_amd_shooting_assemble_route(static_cast<int>(nz), static_cast<int>(nu), static_cast<int>(ne), static_cast<int>(nh), prepare);
}

int ungar_shooting_recover_inputs(const ungar_shooting_dims* dims, int64_t ne, const double* E, const double* eq_reduced, const int32_t* eq_pivots, const double* dZ, double* dU,
                                  int32_t* status, void* stream) {
    ShootingRecoverArgs k{};
    if (!dims || !ToDims(*dims, &k.d) || ne < 0 || ne > 64) return Fail(UNGAR_E_INVALID, "ungar_shooting_recover_inputs: bad dimensions");
    if (k.d.batch == 0 || ne == 0) return UNGAR_OK;
    if (!E || !eq_reduced || !eq_pivots || !dZ || !dU) return Fail(UNGAR_E_INVALID, "ungar_shooting_recover_inputs: null argument");
    k.ne = static_cast<int>(ne);
    k.E = E;
    k.er = eq_reduced;
    k.pivots = eq_pivots;
    k.dZ = dZ;
    k.dU = dU;
    k.status = status;
    return Launched(ungar_amd_launch_shooting_recover(&k, stream), "ungar_shooting_recover_inputs");
}

int ungar_shooting_refresh_carried_inputs(const ungar_shooting_dims* dims, double* rows, void* stream) {
    ShootingDims d{};
    if (!dims || !ToDims(*dims, &d) || !d.carryInputs) return Fail(UNGAR_E_INVALID, "ungar_shooting_refresh_carried_inputs: needs the dimensions of a carry_inputs problem");
    if (d.batch == 0) return UNGAR_OK;
    if (!rows) return Fail(UNGAR_E_INVALID, "ungar_shooting_refresh_carried_inputs: null rows");
    return Launched(ungar_amd_launch_shooting_refresh_carried_inputs(&d, rows, stream), "ungar_shooting_refresh_carried_inputs");
}

int ungar_shooting_merit(const ungar_shooting_merit_args* a, void* stream) {
    ShootingMeritArgs k{};
    if (!a || !ToDims(a->dims, &k.d) || a->period < 0 || (a->period > 0 && a->dims.batch % a->period != 0) || a->nh < 0 || a->nh > 256 || a->ne < 0 || a->ne > 64)
        return Fail(UNGAR_E_INVALID, "ungar_shooting_merit: bad dimensions (nh <= 256, ne <= 64 as in ungar_shooting_assemble)");
    if (k.d.batch == 0) return UNGAR_OK;
    if (!a->rows || !a->xm || !a->f || !a->cost || !a->theta || !a->phi) return Fail(UNGAR_E_INVALID, "ungar_shooting_merit: null argument");
    if ((a->nh > 0 && !a->h) || (a->ne > 0 && !a->eq)) return Fail(UNGAR_E_INVALID, "ungar_shooting_merit: nh > 0 needs h and ne > 0 needs eq (their terms would be dropped silently)");
    if (a->cost_grad && (BadPattern(a->cost_grad_pattern, a->cost_grad) || !a->dZ || !a->dU || !a->slope)) return Fail(UNGAR_E_INVALID, "ungar_shooting_merit: the slope needs the gradient pattern, dZ, dU and slope");
    k.rows = a->rows;
    k.xm = a->xm;
    k.f = a->f;
    k.l = a->cost;
    k.h = a->nh > 0 ? a->h : nullptr;
    k.e = a->ne > 0 ? a->eq : nullptr;
    k.nh = static_cast<int>(a->nh);
    k.ne = static_cast<int>(a->ne);
    k.barrier = Barrier(a->barrier);
    k.violationMultiplier = a->violation_multiplier;
    k.lg = a->cost_grad;
    k.pg = a->cost_grad ? Pattern(a->cost_grad_pattern) : StagePattern{nullptr, nullptr, 0};
    k.dZ = a->dZ;
    k.dU = a->dU;
    k.theta = a->theta;
    k.phi = a->phi;
    k.objective = a->objective;
    k.slope = a->slope;
    k.period = a->period;
    k.instances = a->period > 0 ? a->instances : nullptr;
    k.rowsStride = a->rows_stride;
    k.valueStride = a->value_stride;
    if (a->value_stride < 0 || (a->value_stride > 0 && a->value_stride < a->dims.nx + 1 + a->nh + a->ne))
        return Fail(UNGAR_E_INVALID, "ungar_shooting_merit: value_stride smaller than the values of a node (nx + 1 + nh + ne)");
    if (a->rows_stride < 0 || (a->rows_stride > 0 && a->rows_stride < a->dims.batch * (a->dims.horizon + 1))) return Fail(UNGAR_E_INVALID, "ungar_shooting_merit: rows_stride smaller than the number of nodes");
    return Launched(ungar_amd_launch_shooting_merit(&k, stream), "ungar_shooting_merit");
}

int ungar_shooting_trial_rows(const ungar_shooting_dims* dims, const double* rows, const double* dZ, const double* dU, const double* alphas, int64_t candidates,
                              double* trial, int64_t trial_stride, void* stream) {
    return ungar_shooting_trial_rows_listed(dims, rows, dZ, dU, alphas, candidates, nullptr, 0, trial, trial_stride, stream);
}

int ungar_shooting_trial_rows_listed(const ungar_shooting_dims* dims, const double* rows, const double* dZ, const double* dU, const double* alphas, int64_t candidates,
                                     const int32_t* instances, int64_t listed, double* trial, int64_t trial_stride, void* stream) {
    return ungar_shooting_trial_elements(dims, rows, dZ, dU, alphas, candidates, instances, listed, 0, 0, trial, trial_stride, stream);
}

int ungar_shooting_trial_elements(const ungar_shooting_dims* dims, const double* rows, const double* dZ, const double* dU, const double* alphas, int64_t candidates,
                                  const int32_t* instances, int64_t listed, int64_t first_element, int64_t elements, double* trial, int64_t trial_stride, void* stream) {
    ShootingTrialArgs k{};
    if (!dims || !ToDims(*dims, &k.d) || !alphas || candidates < 1 || candidates > kMaxLineSearchCandidates)
        return Fail(UNGAR_E_INVALID, "ungar_shooting_trial_rows: bad argument (1 <= candidates <= 16)");
    if (listed < 0 || listed > k.d.batch || (listed > 0) != (instances != nullptr)) return Fail(UNGAR_E_INVALID, "ungar_shooting_trial_rows_listed: 0 <= listed <= batch, with the list exactly when listed > 0");
    if (k.d.batch == 0) return UNGAR_OK;
    k.instances = instances;
    k.listed = listed;
    if (!rows || !dZ || !dU || !trial) return Fail(UNGAR_E_INVALID, "ungar_shooting_trial_rows: null argument");
    k.rows = rows;
    k.dZ = dZ;
    k.dU = dU;
    k.trial = trial;
    k.candidates = static_cast<int>(candidates);
    for (int64_t c = 0; c < candidates; ++c) k.alphas[c] = alphas[c];
    if (trial_stride < 0 || (trial_stride > 0 && trial_stride < candidates * (listed > 0 ? listed : k.d.batch) * (k.d.N + 1))) return Fail(UNGAR_E_INVALID, "ungar_shooting_trial_rows: trial_stride smaller than the number of stacked nodes");
    k.trialStride = trial_stride;
    if (first_element < 0 || elements < 0 || first_element + elements > k.d.nv() || ((first_element > 0 || elements > 0) && trial_stride == 0))
        return Fail(UNGAR_E_INVALID, "ungar_shooting_trial_elements: the element window must lie inside the row and needs the unit-fastest image (trial_stride > 0)");
    k.first = static_cast<int>(first_element);
    k.elements = static_cast<int>(elements);
    return Launched(ungar_amd_launch_shooting_trial(&k, stream), "ungar_shooting_trial_rows");
}

int ungar_shooting_select(const ungar_shooting_dims* dims, const ungar_line_search_parameters* p, const double* alphas, int64_t candidates, const double* theta0,
                          const double* phi0, const double* objective0, const double* slope, const double* theta_trial, const double* phi_trial,
                          const double* objective_trial, double* accepted, int32_t* active, const int32_t* status, double* rows, const double* trial, int64_t trial_stride,
                          int32_t stage, int32_t* unresolved, void* stream) {
    return ungar_shooting_select_listed(dims, p, alphas, candidates, theta0, phi0, objective0, slope, theta_trial, phi_trial, objective_trial, accepted, active, status, rows, trial, trial_stride,
                                        stage, unresolved, nullptr, 0, nullptr, stream);
}

int ungar_shooting_select_listed(const ungar_shooting_dims* dims, const ungar_line_search_parameters* p, const double* alphas, int64_t candidates, const double* theta0,
                                 const double* phi0, const double* objective0, const double* slope, const double* theta_trial, const double* phi_trial,
                                 const double* objective_trial, double* accepted, int32_t* active, const int32_t* status, double* rows, const double* trial, int64_t trial_stride,
                                 int32_t stage, int32_t* unresolved, const int32_t* instances, int64_t listed, int32_t* next_instances, void* stream) {
    ShootingSelectArgs k{};
    if (!dims || !ToDims(*dims, &k.d) || !p || !alphas || candidates < 1 || candidates > kMaxLineSearchCandidates)
        return Fail(UNGAR_E_INVALID, "ungar_shooting_select: bad argument (1 <= candidates <= 16)");
    if (listed < 0 || listed > k.d.batch || (listed > 0) != (instances != nullptr)) return Fail(UNGAR_E_INVALID, "ungar_shooting_select_listed: 0 <= listed <= batch, with the list exactly when listed > 0");
    if (next_instances && (!unresolved || !(stage & UNGAR_SEARCH_NOT_LAST))) return Fail(UNGAR_E_INVALID, "ungar_shooting_select_listed: next_instances is filled through the counter `unresolved` of a call that is not the last");
    if (next_instances && next_instances == instances) return Fail(UNGAR_E_INVALID, "ungar_shooting_select_listed: next_instances must not be the list being read");
    if (k.d.batch == 0) return UNGAR_OK;
    k.instances = instances;
    k.listed = listed;
    k.nextInstances = next_instances;
    if (!theta0 || !phi0 || !objective0 || !slope || !theta_trial || !phi_trial || !objective_trial || !accepted || !rows || !trial)
        return Fail(UNGAR_E_INVALID, "ungar_shooting_select: null argument");
    if (trial_stride < 0 || (trial_stride > 0 && trial_stride < candidates * (listed > 0 ? listed : k.d.batch) * (k.d.N + 1)))
        return Fail(UNGAR_E_INVALID, "ungar_shooting_select: trial_stride smaller than the number of stacked nodes");
    k.candidates = static_cast<int>(candidates);
    k.thetaMin = p->theta_min;
    k.thetaMax = p->theta_max;
    k.eta = p->eta;
    k.gammaPhi = p->gamma_phi;
    k.gammaTheta = p->gamma_theta;
    for (int64_t c = 0; c < candidates; ++c) k.alphas[c] = alphas[c];
    k.theta0 = theta0;
    k.phi0 = phi0;
    k.objective0 = objective0;
    k.slope = slope;
    k.thetaT = theta_trial;
    k.phiT = phi_trial;
    k.objectiveT = objective_trial;
    k.accepted = accepted;
    k.active = active;
    k.status = status;
    k.first = (stage & UNGAR_SEARCH_NOT_FIRST) ? 0 : 1;
    k.last = (stage & UNGAR_SEARCH_NOT_LAST) ? 0 : 1;
    k.unresolved = unresolved;
    k.trialStride = trial_stride;
    k.rows = rows;
    k.trial = trial;
    return Launched(ungar_amd_launch_shooting_select(&k, stream), "ungar_shooting_select");
}

int ungar_device_malloc(void** out, int64_t bytes) {
    if (!out || bytes < 0) return Fail(UNGAR_E_INVALID, "ungar_device_malloc: bad argument");
    *out = nullptr;
    if (bytes == 0) return UNGAR_OK;
    const hipError_t e = hipMalloc(out, static_cast<std::size_t>(bytes));
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_malloc: ") + hipGetErrorString(e));
    return UNGAR_OK;
}
int ungar_device_free(void* ptr) {
    if (!ptr) return UNGAR_OK;
    const hipError_t e = hipFree(ptr);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_free: ") + hipGetErrorString(e));
    return UNGAR_OK;
}
int ungar_device_upload(void* dst, const void* src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return Fail(UNGAR_E_INVALID, "ungar_device_upload: bad argument");
    if (bytes == 0) return UNGAR_OK;
    const hipError_t e = hipMemcpy(dst, src, static_cast<std::size_t>(bytes), hipMemcpyHostToDevice);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_upload: ") + hipGetErrorString(e));
    return UNGAR_OK;
}
int ungar_device_download(void* dst, const void* src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return Fail(UNGAR_E_INVALID, "ungar_device_download: bad argument");
    if (bytes == 0) return UNGAR_OK;
    const hipError_t e = hipMemcpy(dst, src, static_cast<std::size_t>(bytes), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_download: ") + hipGetErrorString(e));
    return UNGAR_OK;
}
int ungar_device_zero(void* dst, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && !dst)) return Fail(UNGAR_E_INVALID, "ungar_device_zero: bad argument");
    if (bytes == 0) return UNGAR_OK;
    const hipError_t e = hipMemsetAsync(dst, 0, static_cast<std::size_t>(bytes), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_zero: ") + hipGetErrorString(e));
    return UNGAR_OK;
}
int ungar_device_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return Fail(UNGAR_E_INVALID, "ungar_device_copy: bad argument");
    if (bytes == 0) return UNGAR_OK;
    const hipError_t e = hipMemcpyAsync(dst, src, static_cast<std::size_t>(bytes), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_copy: ") + hipGetErrorString(e));
    return UNGAR_OK;
}
int ungar_device_read_polled(void* dst, const void* src, int64_t bytes, void* stream) {
    if (bytes < 0 || bytes > 64 || (bytes > 0 && (!dst || !src))) return Fail(UNGAR_E_INVALID, "ungar_device_read_polled: bad argument (at most 64 bytes)");
    if (bytes == 0) return UNGAR_OK;
    struct Staging {  // per host thread, for the life of the process
        unsigned char* data = nullptr;
        unsigned long long *flag = nullptr, *flagDevice = nullptr;
        unsigned long long tickets = 0;
        bool streamWrites = true;
    };
    static thread_local Staging st;
    hipError_t e = hipSuccess;
    if (!st.data) {
        void* block = nullptr;
        e = hipHostMalloc(&block, 128, hipHostMallocMapped);
        if (e == hipSuccess) {
            st.data = static_cast<unsigned char*>(block);
            st.flag = reinterpret_cast<unsigned long long*>(st.data + 64);
            *st.flag = 0;
            e = hipHostGetDevicePointer(reinterpret_cast<void**>(&st.flagDevice), st.flag, 0);
        }
        if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_read_polled: ") + hipGetErrorString(e));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    e = hipMemcpyAsync(st.data, src, static_cast<std::size_t>(bytes), hipMemcpyDeviceToHost, s);
    bool polled = false;
    if (e == hipSuccess && st.streamWrites) {
        const unsigned long long ticket = ++st.tickets;
        if (hipStreamWriteValue64(s, st.flagDevice, ticket, 0) == hipSuccess) {
            volatile unsigned long long* flag = st.flag;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
            unsigned spins = 0;
            while (*flag != ticket)
                if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() > deadline) break;  // a stuck queue: the wait below reports it
            polled = *flag == ticket;
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            (void)hipGetLastError();
            st.streamWrites = false;
        }
    }
    if (e == hipSuccess && !polled) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_read_polled: ") + hipGetErrorString(e));
    std::memcpy(dst, st.data, static_cast<std::size_t>(bytes));
    return UNGAR_OK;
}
int ungar_device_synchronize(void) {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("ungar_device_synchronize: ") + hipGetErrorString(e));
    return UNGAR_OK;
}

}  // extern "C"
