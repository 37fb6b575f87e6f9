// ungar_amd :: C ABI of the batched SQP iteration (include/ungar_amd.h, "batched SQP iteration"): argument checks and launches.
#include <hip/hip_runtime.h>

#include <string>

#include "../../../include/ungar_amd.h"
#include "../kernels/ocp_sqp.hpp"

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
    return nx < 1 || nu < 1 || N < 1 || batch < 0 || nx > 255 || nu > 255 || N > (1 << 20);
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

int ungar_ocp_riccati_solve(const ungar_ocp_qp* q, void* stream) {
    if (!q || BadDims(q->nx, q->nu, q->horizon, q->batch)) return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: bad dimensions");
    if (q->batch == 0) return UNGAR_OK;
    if (!q->jac.base || !q->b.base || !q->hess.base || !q->grad.base || !q->dx0.base || !q->dX.base || !q->dU.base)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: null operand base");
    if (!q->workspace || q->workspace_doubles < ungar_ocp_riccati_workspace(q->nx, q->nu, q->horizon, q->batch))
        return Fail(UNGAR_E_INVALID, "ungar_ocp_riccati_solve: workspace too small (see ungar_ocp_riccati_workspace)");
    if (static_cast<std::size_t>(RiccatiScratchDoubles(static_cast<int>(q->nx), static_cast<int>(q->nu))) * sizeof(double) > 160 * 1024)
        return Fail(UNGAR_E_UNSUPPORTED, "ungar_ocp_riccati_solve: nx + nu too large for the LDS-resident recursion (160 KiB per workgroup)");
    const RiccatiArgs k{static_cast<int>(q->nx), static_cast<int>(q->nu), static_cast<int>(q->horizon), q->batch, View(q->jac), View(q->b), View(q->hess), View(q->grad),
                        View(q->hess_terminal), View(q->grad_terminal), View(q->dx0), View(q->dX), View(q->dU), q->workspace, q->regularization, q->status};
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
                                 const ungar_operand* X, const ungar_operand* U, const ungar_operand* Xt, const ungar_operand* Ut, void* stream) {
    if (BadDims(nx, nu, horizon, batch) || !p || !alphas || candidates < 1 || candidates > kMaxLineSearchCandidates || !theta0 || !phi0 || !slope || !theta_trial || !phi_trial ||
        !accepted || !X || !U || !Xt || !Ut)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_select: bad argument (1 <= candidates <= 16)");
    if (batch == 0) return UNGAR_OK;
    if (!X->base || !U->base || !Xt->base || !Ut->base) return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_select: null operand base");
    SelectArgs k{static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(horizon), static_cast<int>(candidates), batch, p->theta_min, p->theta_max, p->eta, p->gamma_phi,
                 p->gamma_theta, {}, theta0, phi0, slope, theta_trial, phi_trial, accepted, View(*X), View(*U), View(*Xt), View(*Ut)};
    for (int64_t c = 0; c < candidates; ++c) k.alphas[c] = alphas[c];
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
                                 const ungar_operand* U, const ungar_operand* Xt, const ungar_operand* Ut, void* stream) {
    if (BadDims(nx, nu, horizon, batch) || !p || !theta0 || !phi0 || !slope || !theta_trial || !phi_trial || !accepted || !X || !U || !Xt || !Ut)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_accept: bad argument");
    if (batch == 0) return UNGAR_OK;
    if (!X->base || !U->base || !Xt->base || !Ut->base) return Fail(UNGAR_E_INVALID, "ungar_ocp_line_search_accept: null operand base");
    const AcceptArgs k{static_cast<int>(nx), static_cast<int>(nu), static_cast<int>(horizon), batch, alpha, p->theta_min, p->theta_max, p->eta, p->gamma_phi, p->gamma_theta,
                       theta0, phi0, slope, theta_trial, phi_trial, accepted, View(*X), View(*U), View(*Xt), View(*Ut)};
    return Launched(ungar_amd_launch_ocp_accept(&k, stream), "ungar_ocp_line_search_accept");
}

}  // extern "C"
