// ungar_amd :: batched solve of the SQP's quadratic subproblem for optimal-control structure.
//
// The reference hands every QP of its soft SQP to OSQP as one sparse problem (include/ungar/optimization/
// soft_sqp.hpp:143-158):   min_d 1/2 d^T H d + g^T d   s.t.  J_g d = -g(z),   H = hess f + J_h^T diag(b'') J_h + 1e-6 I.
// For a shooting problem with stage-wise cost and soft inequalities, z = [x_0..x_N | u_0..u_{N-1}],
//     g = [x_0 - x_m ; x_{k+1} - f(x_k, u_k)]   (example/mpc/quadrotor.example.cpp:246-266),
// H is block diagonal over the knots and J_g block bidiagonal, so the KKT system is solved EXACTLY by the discrete
// Riccati recursion -- O(N (nx + nu)^3) per instance, no fill-in, no iteration count to tune:
//     dx_0 given,  dx_{k+1} = A_k dx_k + B_k du_k + b_k                    (A|B = node Jacobian, b_k = f_k - x_{k+1})
//     P_N = W_N, p_N = w_N;   for k = N-1 .. 0, with AB = [A_k | B_k]:
//       H = W_k + AB^T P AB,   h = w_k + AB^T (P b_k + p)                  (n x n, n = nx + nu)
//       R = H_uu = L L^T,  [K | kff] = -R^-1 [H_ux | h_u],   P <- H_xx + H_ux^T K,   p <- h_x + H_ux^T kff
//     forward:  du_k = K_k dx_k + kff_k,  dx_{k+1} = AB [dx_k; du_k] + b_k.
// One 64-lane workgroup owns one MPC instance: every matrix of the recursion lives in LDS, the lanes share the entries of
// each product (Exec::ForEach) and meet at workgroup barriers; thousands of instances fill the device.
//
// The recursion is written once, generic over an execution policy: DeviceExec (ocp_riccati.hip) strides the lanes of a
// workgroup over the index range; a sequential host policy exists ONLY in tests/cpp/riccati_host.cpp so that the very same
// source can be pinned against a dense KKT solve without a GPU (test infrastructure, like tests/cpp/quad_sim.cpp).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UNGAR_HD __host__ __device__
#else
#define UNGAR_HD
#endif

#include <cmath>

namespace ungar_amd::kernels {

/// Strided view (instance, knot, element) of one operand, in doubles.
struct RiccatiView {
    double* base;
    long long bs, ks, es;
    UNGAR_HD double& at(long long b, long long k, long long e) const {
        return base[b * bs + k * ks + e * es];
    }
};

struct RiccatiArgs {
    int nx, nu, N;
    long long batch;
    RiccatiView jac;    // dense nx x (nx+nu) row-major block per node (b, k), k < N   -- what ungar_model_dense_jacobian writes
    RiccatiView b;      // nx per node: affine term of the linearised dynamics
    RiccatiView hess;   // (nx+nu)^2 per node, row-major; only entries with row <= col are read (upper triangle, function.hpp:232-235)
    RiccatiView grad;   // nx+nu per node
    RiccatiView hessN;  // terminal: nx^2 per instance (upper triangle read) -- base may be null: zero terminal cost
    RiccatiView gradN;  // terminal: nx per instance -- base may be null
    RiccatiView dx0;    // nx per instance
    RiccatiView dX;     // out: (N+1) nx per instance, element e of knot k at (b, k, e)
    RiccatiView dU;     // out: N nu per instance
    double* gains;      // workspace: batch * N * nu * (nx + 1) doubles (feedback gains and feed-forward terms)
    double regularization;  // added to the diagonal of every H and of P_N (the reference's 1e-6 I, soft_sqp.hpp:149-151)
    int* status;        // per instance: 0 ok, k+1 = R_k not positive definite at knot k (may be null)
};

/// Doubles of LDS (or host scratch) one instance needs.
UNGAR_HD inline int RiccatiScratchDoubles(int nx, int nu) {
    const int n = nx + nu;
    return nx * n /*AB*/ + n * n /*H*/ + 2 * nx * nx /*P, Pn*/ + nx * n /*PAB*/ + 2 * nx /*p, pn*/ + nx /*t*/ + n /*h*/ + nx /*bk*/ + nu * (nx + 1) /*K|kff*/ +
           nx + nu /*dx, du*/ + nx /*dxn*/;
}

/// The whole recursion for instance `inst`; `scratch` holds RiccatiScratchDoubles(nx, nu) doubles private to the workgroup.
template <class Exec>
UNGAR_HD void RiccatiInstance(const RiccatiArgs& a, long long inst, double* scratch, Exec& ex) {
    const int nx = a.nx, nu = a.nu, n = nx + nu, N = a.N, nk = nx + 1;
    double* AB = scratch;
    double* H = AB + nx * n;
    double* P = H + n * n;
    double* Pn = P + nx * nx;
    double* PAB = Pn + nx * nx;
    double* p = PAB + nx * n;
    double* pn = p + nx;
    double* t = pn + nx;
    double* h = t + nx;
    double* bk = h + n;
    double* K = bk + nx;  // nu x (nx + 1): [K | kff]
    double* dx = K + nu * nk;
    double* du = dx + nx;
    double* dxn = du + nu;
    double* gains = a.gains + inst * static_cast<long long>(N) * nu * nk;
    int failed = 0;

    // terminal cost-to-go
    ex.ForEach(nx * nx, [&](int idx) {
        const int i = idx / nx, j = idx % nx;
        double v = 0.0;
        if (a.hessN.base) v = i <= j ? a.hessN.at(inst, 0, i * nx + j) : a.hessN.at(inst, 0, j * nx + i);
        P[idx] = v + (i == j ? a.regularization : 0.0);
    });
    ex.ForEach(nx, [&](int i) { p[i] = a.gradN.base ? a.gradN.at(inst, 0, i) : 0.0; });

    for (int k = N - 1; k >= 0; --k) {
        ex.ForEach(nx * n, [&](int idx) { AB[idx] = a.jac.at(inst, k, idx); });
        ex.ForEach(nx, [&](int i) { bk[i] = a.b.at(inst, k, i); });
        // PAB = P AB;  t = P b + p
        ex.ForEach(nx * n, [&](int idx) {
            const int i = idx / n, c = idx % n;
            double acc = 0.0;
            for (int m = 0; m < nx; ++m) acc += P[i * nx + m] * AB[m * n + c];
            PAB[idx] = acc;
        });
        ex.ForEach(nx, [&](int i) {
            double acc = p[i];
            for (int m = 0; m < nx; ++m) acc += P[i * nx + m] * bk[m];
            t[i] = acc;
        });
        // H = W + AB^T PAB (symmetric: computed for r <= c, mirrored);  h = w + AB^T t
        ex.ForEach(n * n, [&](int idx) {
            const int r = idx / n, c = idx % n;
            if (r > c) return;
            double acc = a.hess.at(inst, k, r * n + c) + (r == c ? a.regularization : 0.0);
            for (int m = 0; m < nx; ++m) acc += AB[m * n + r] * PAB[m * n + c];
            H[r * n + c] = acc;
            H[c * n + r] = acc;
        });
        ex.ForEach(n, [&](int c) {
            double acc = a.grad.at(inst, k, c);
            for (int m = 0; m < nx; ++m) acc += AB[m * n + c] * t[m];
            h[c] = acc;
        });
        // Cholesky of R = H_uu in place (lower triangle of the uu block): column by column, rows in parallel
        for (int j = 0; j < nu; ++j) {
            ex.ForEach(1, [&](int) {
                double d = H[(nx + j) * n + nx + j];
                for (int m = 0; m < j; ++m) d -= H[(nx + j) * n + nx + m] * H[(nx + j) * n + nx + m];
                if (!(d > 0.0)) {
                    failed = failed ? failed : k + 1;
                    d = 1.0;
                }
                H[(nx + j) * n + nx + j] = sqrt(d);
            });
            ex.ForEach(nu - j - 1, [&](int q) {
                const int i = j + 1 + q;
                double s = H[(nx + i) * n + nx + j];
                for (int m = 0; m < j; ++m) s -= H[(nx + i) * n + nx + m] * H[(nx + j) * n + nx + m];
                H[(nx + i) * n + nx + j] = s / H[(nx + j) * n + nx + j];
            });
        }
        // [K | kff] = -R^-1 [H_ux | h_u]: one right-hand side per lane (forward then backward substitution)
        ex.ForEach(nk, [&](int c) {
            for (int i = 0; i < nu; ++i) {  // L y = rhs
                double s = c < nx ? -H[(nx + i) * n + c] : -h[nx + i];
                for (int m = 0; m < i; ++m) s -= H[(nx + i) * n + nx + m] * K[m * nk + c];
                K[i * nk + c] = s / H[(nx + i) * n + nx + i];
            }
            for (int i = nu - 1; i >= 0; --i) {  // L^T x = y
                double s = K[i * nk + c];
                for (int m = i + 1; m < nu; ++m) s -= H[(nx + m) * n + nx + i] * K[m * nk + c];
                K[i * nk + c] = s / H[(nx + i) * n + nx + i];
            }
        });
        ex.ForEach(nu * nk, [&](int idx) { gains[static_cast<long long>(k) * nu * nk + idx] = K[idx]; });
        // P <- H_xx + H_ux^T K (symmetrised),  p <- h_x + H_ux^T kff
        ex.ForEach(nx * nx, [&](int idx) {
            const int i = idx / nx, j = idx % nx;
            double s1 = H[i * n + j], s2 = H[j * n + i];
            for (int m = 0; m < nu; ++m) {
                s1 += H[(nx + m) * n + i] * K[m * nk + j];
                s2 += H[(nx + m) * n + j] * K[m * nk + i];
            }
            Pn[idx] = 0.5 * (s1 + s2);
        });
        ex.ForEach(nx, [&](int i) {
            double s = h[i];
            for (int m = 0; m < nu; ++m) s += H[(nx + m) * n + i] * K[m * nk + nx];
            pn[i] = s;
        });
        ex.ForEach(nx * nx, [&](int idx) { P[idx] = Pn[idx]; });
        ex.ForEach(nx, [&](int i) { p[i] = pn[i]; });
    }

    // forward pass
    ex.ForEach(nx, [&](int i) {
        dx[i] = a.dx0.at(inst, 0, i);
        a.dX.at(inst, 0, i) = dx[i];
    });
    for (int k = 0; k < N; ++k) {
        ex.ForEach(nx * n, [&](int idx) { AB[idx] = a.jac.at(inst, k, idx); });
        ex.ForEach(nu, [&](int i) {
            const double* g = gains + static_cast<long long>(k) * nu * nk + i * nk;
            double s = g[nx];
            for (int m = 0; m < nx; ++m) s += g[m] * dx[m];
            du[i] = s;
            a.dU.at(inst, k, i) = s;
        });
        ex.ForEach(nx, [&](int i) {
            double s = a.b.at(inst, k, i);
            for (int m = 0; m < nx; ++m) s += AB[i * n + m] * dx[m];
            for (int m = 0; m < nu; ++m) s += AB[i * n + nx + m] * du[m];
            dxn[i] = s;
            a.dX.at(inst, k + 1, i) = s;
        });
        ex.ForEach(nx, [&](int i) { dx[i] = dxn[i]; });
    }
    if (a.status) ex.ForEach(1, [&](int) { a.status[inst] = failed; });
}

}  // namespace ungar_amd::kernels
