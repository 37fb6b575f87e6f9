"""Batched soft SQP for shooting problems on the device.

Host-side mirror of `Ungar::SoftSQPOptimizer::Optimize` (reference include/ungar/optimization/soft_sqp.hpp:62-112) for a whole
batch of independent MPC instances with the structure of the reference's examples: dynamics constraint
g = [x_0 - x_m; x_{k+1} - f(x_k, u_k)] (example/mpc/quadrotor.example.cpp:246-266), stage-wise cost, stage-wise soft
inequalities h <= 0 behind a relaxed barrier.  Every step is a stream-ordered call of the C ABI (include/ungar_amd.h) on
device-resident data -- no host round trip inside an iteration:

    derivatives   ungar_model_dense_jacobian (dynamics, inequalities), ungar_model_sparse_hessian (stage cost)
    QP data       ungar_ocp_stage_qp          b_k, W_k = hess cost + J_h^T diag(b'') J_h, w_k = grad cost - J_h^T b'
    QP solve      ungar_ocp_riccati_solve     exact solution of the KKT system the reference hands to OSQP (:143-158)
    line search   ungar_ocp_trial_point / ungar_model_forward_zero / ungar_ocp_merit / ungar_ocp_line_search_accept
                  the three-way test of backtracking_line_search.hpp:116-151, alpha = 1, 1/2, ... >= alpha_min

torch only owns the device buffers.  Layout: states X (batch, N+1, nx), inputs U (batch, N, nu), node-major (what a
VariableMap buffer [X | U] holds, example/mpc/quadrotor.example.cpp:103-117); node outputs are node-major too.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import NodeModel, Operand, _check, _Operand, load_library


class _Barrier(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stiffness", ctypes.c_double), ("epsilon", ctypes.c_double)]


class _StageQpArgs(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int64), ("nu", ctypes.c_int64), ("horizon", ctypes.c_int64), ("batch", ctypes.c_int64), ("X", _Operand), ("xm", _Operand), ("f", _Operand),
                ("cost_grad", _Operand), ("cost_hes", _Operand), ("hes_rows", ctypes.c_void_p), ("hes_cols", ctypes.c_void_p), ("hes_nnz", ctypes.c_int64),
                ("nh", ctypes.c_int64), ("h", _Operand), ("h_jac", _Operand), ("barrier", _Barrier), ("b", _Operand), ("hess", _Operand), ("grad", _Operand),
                ("dx0", _Operand)]


class _Qp(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int64), ("nu", ctypes.c_int64), ("horizon", ctypes.c_int64), ("batch", ctypes.c_int64), ("jac", _Operand), ("b", _Operand),
                ("hess", _Operand), ("grad", _Operand), ("hess_terminal", _Operand), ("grad_terminal", _Operand), ("dx0", _Operand), ("dX", _Operand), ("dU", _Operand),
                ("workspace", ctypes.c_void_p), ("workspace_doubles", ctypes.c_int64), ("regularization", ctypes.c_double), ("status", ctypes.c_void_p),
                ("ne", ctypes.c_int64), ("eq", _Operand), ("eq_values", _Operand), ("hess_terminal_ld", ctypes.c_int64)]


class _MeritArgs(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int64), ("nu", ctypes.c_int64), ("horizon", ctypes.c_int64), ("batch", ctypes.c_int64), ("nh", ctypes.c_int64), ("X", _Operand),
                ("xm", _Operand), ("f", _Operand), ("cost", _Operand), ("cost_terminal", _Operand), ("h", _Operand), ("barrier", _Barrier),
                ("violation_multiplier", ctypes.c_double), ("cost_grad", _Operand), ("cost_grad_terminal", _Operand), ("dX", _Operand), ("dU", _Operand),
                ("theta", ctypes.c_void_p), ("phi", ctypes.c_void_p), ("slope", ctypes.c_void_p)]


class _LineSearchParameters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("alpha_min", "theta_min", "theta_max", "eta", "gamma_phi", "gamma_theta", "gamma_alpha")]


_NULL = _Operand(None, 0, 0, 0)


def _declare(lib):
    if getattr(lib, "_ungar_sqp_declared", False):
        return lib
    vp, op = ctypes.c_void_p, ctypes.POINTER(_Operand)
    lib.ungar_ocp_stage_qp.argtypes = [ctypes.POINTER(_StageQpArgs), vp]
    lib.ungar_ocp_riccati_workspace.argtypes = [ctypes.c_int64] * 4
    lib.ungar_ocp_riccati_workspace.restype = ctypes.c_int64
    lib.ungar_ocp_riccati_solve.argtypes = [ctypes.POINTER(_Qp), vp]
    lib.ungar_ocp_merit.argtypes = [ctypes.POINTER(_MeritArgs), vp]
    lib.ungar_ocp_trial_point.argtypes = [ctypes.c_int64] * 4 + [op] * 4 + [ctypes.c_double] + [op] * 2 + [vp]
    lib.ungar_ocp_merit_stacked.argtypes = [ctypes.POINTER(_MeritArgs), ctypes.c_int64, vp]
    lib.ungar_ocp_trial_points.argtypes = [ctypes.c_int64] * 4 + [op] * 4 + [ctypes.POINTER(ctypes.c_double), ctypes.c_int64] + [op] * 2 + [vp]
    lib.ungar_ocp_line_search_select.argtypes = ([ctypes.c_int64] * 4 + [ctypes.POINTER(_LineSearchParameters), ctypes.POINTER(ctypes.c_double), ctypes.c_int64] + [vp] * 6 +
                                                 [op] * 4 + [vp, vp])
    lib.ungar_ocp_line_search_accept.argtypes = [ctypes.c_int64] * 4 + [ctypes.POINTER(_LineSearchParameters), ctypes.c_double] + [vp] * 6 + [op] * 4 + [vp, vp]
    lib._ungar_sqp_declared = True
    return lib


def _node(t, elements, knots):
    """node-major tensor (batch, knots, elements) as an operand"""
    return Operand(t, instance_stride=knots * elements, knot_stride=elements, element_stride=1)


def _inst(t, elements):
    return Operand(t, instance_stride=elements, knot_stride=0, element_stride=1)


def riccati_solve(nx, nu, horizon, batch, jac, b, hess, grad, dx0, hess_terminal=None, grad_terminal=None, regularization=1e-6, stream=None):
    """Batched exact QP solve on node-major device tensors: jac (batch, N, nx, nx+nu), b (batch, N, nx), hess (batch, N, n, n)
    (upper triangle read), grad (batch, N, n), dx0 (batch, nx), optional terminal hess (batch, nx, nx) / grad (batch, nx).
    Returns (dX (batch, N+1, nx), dU (batch, N, nu), status (batch,) int32)."""
    import torch
    lib = _declare(load_library())
    n = nx + nu
    dX = torch.empty((batch, horizon + 1, nx), dtype=torch.float64, device="cuda")
    dU = torch.empty((batch, horizon, nu), dtype=torch.float64, device="cuda")
    status = torch.zeros((batch,), dtype=torch.int32, device="cuda")
    ws = torch.empty((max(1, lib.ungar_ocp_riccati_workspace(nx, nu, horizon, batch)),), dtype=torch.float64, device="cuda")
    q = _Qp(nx, nu, horizon, batch, _node(jac, nx * n, horizon)._c(), _node(b, nx, horizon)._c(), _node(hess, n * n, horizon)._c(), _node(grad, n, horizon)._c(),
            _inst(hess_terminal, nx * nx)._c() if hess_terminal is not None else _NULL, _inst(grad_terminal, nx)._c() if grad_terminal is not None else _NULL,
            _inst(dx0, nx)._c(), _node(dX, nx, horizon + 1)._c(), _node(dU, nu, horizon)._c(), ws.data_ptr(), ws.numel(), regularization, status.data_ptr())
    _check(lib.ungar_ocp_riccati_solve(ctypes.byref(q), NodeModel._stream(stream)))
    return dX, dU, status


@dataclass
class LineSearchParameters:
    """Defaults of the reference (backtracking_line_search.hpp:58-77)."""
    alpha_min: float = 1e-4
    theta_min: float = 1e-6
    theta_max: float = 1e-2
    eta: float = 1e-4
    gamma_phi: float = 1e-6
    gamma_theta: float = 1e-6
    gamma_alpha: float = 0.5


class BatchedSoftSqp:
    """Soft SQP iterations for `batch` independent instances of one shooting problem, entirely on the device.

    dynamics / cost / inequality are names of node models (ungar_model_open): x+ = f(x, u; w, p), scalar stage cost with
    gradient and upper Hessian, inequality rows h(x, u) <= 0 (optional).  Parameters follow SoftSQPOptimizer's constructor
    (soft_sqp.hpp:44-60): constraint-violation multiplier, stiffness / epsilon / type of the relaxed barrier."""

    def __init__(self, dynamics: str, cost: str, horizon: int, batch: int, inequality: str | None = None, constraint_violation_multiplier: float = 1.0,
                 stiffness: float = 100.0, epsilon: float = 2e-5, barrier: str = "poly", regularization: float = 1e-6,
                 line_search: LineSearchParameters | None = None, jacobian_in_place: bool = False):
        """jacobian_in_place: the Riccati recursion reads a wide unit-fastest [A|B] through its element stride instead of a node-major transpose
        (identical bits; measured slower, profiles/archive/r03f_jacobian_in_place_ab.log -- kept selectable for that comparison)."""
        import torch
        self.torch = torch
        self.lib = _declare(load_library())
        self.dyn, self.cost = NodeModel(dynamics), NodeModel(cost)
        self.ineq = NodeModel(inequality) if inequality else None
        self.nx, self.nu, self.N, self.batch = self.dyn.nx, self.dyn.nu, horizon, batch
        if (self.cost.nx, self.cost.nu, self.cost.ny) != (self.nx, self.nu, 1) or not self.cost.implements_hessian():
            raise ValueError("the cost model must be a scalar node model over the same (x, u)")
        if self.ineq is not None and (self.ineq.nx, self.ineq.nu) != (self.nx, self.nu):
            raise ValueError("the inequality model must take the same (x, u)")
        self.nh = self.ineq.ny if self.ineq is not None else 0
        self.multiplier, self.regularization = constraint_violation_multiplier, regularization
        self.barrier = _Barrier(1 if barrier == "log" else 0, 0, stiffness, epsilon)
        self.ls = line_search or LineSearchParameters()
        # the stacked search evaluates all candidates in one batch of at most 16; more candidates (gamma_alpha closer to 1, smaller alpha_min)
        # are walked one after the other -- decided here, not in the middle of an iteration
        self._stackable = len(self.candidate_steps()) <= 16
        self.dyn.prepare()
        rows, cols = self.cost.hessian_sparsity()
        self._hes_rows, self._hes_cols = np.ascontiguousarray(rows, dtype=np.int32), np.ascontiguousarray(cols, dtype=np.int32)
        n, nx, nu, N, B = self.nx + self.nu, self.nx, self.nu, horizon, batch
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device="cuda")  # noqa: E731
        self.f, self.J = z(B, N, nx), z(B, N, nx, n)
        self.c, self.cgrad, self.chess = z(B, N, 1), z(B, N, n), z(B, N, len(rows))
        self.h, self.hJ = (z(B, N, self.nh), z(B, N, self.nh, n)) if self.nh else (None, None)
        self.b, self.W, self.w, self.dx0 = z(B, N, nx), z(B, N, n, n), z(B, N, n), z(B, nx)
        self.dX, self.dU, self.Xt, self.Ut = z(B, N + 1, nx), z(B, N, nu), z(B, N + 1, nx), z(B, N, nu)
        self.theta0, self.phi0, self.slope, self.thetaT, self.phiT, self.accepted = (z(B) for _ in range(6))
        self.status = torch.zeros((B,), dtype=torch.int32, device="cuda")
        self.workspace = z(max(1, self.lib.ungar_ocp_riccati_workspace(nx, nu, N, B)))
        self._stack = None  # buffers of the stacked line search, allocated on first use
        self._transpose_jacobian = not jacobian_in_place
        self._wide = None   # unit-fastest scratch of a wide dynamics Jacobian (>= 1024 entries per node and no node parameters w)
        if nx * n >= 1024 and self.dyn.nw == 0:
            from .sharding import unit_fastest
            self._wide = (unit_fastest(nx, B * (N + 1), torch), unit_fastest(nu, B * N, torch), unit_fastest(nx, B * N, torch), unit_fastest(nx * n, B * N, torch))

    def _on(self, stream):
        """torch's own kernels (fills, copies) issued on the SAME stream as the C-ABI launches; the default stream is torch's current one."""
        import contextlib
        if not stream or (isinstance(stream, ctypes.c_void_p) and not stream.value):
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.torch.cuda.ExternalStream(stream.value if isinstance(stream, ctypes.c_void_p) else stream))

    # -- operands -----------------------------------------------------------------------------------------------------
    def _states(self, X):  # knots 0..N-1 of the (batch, N+1, nx) buffer as the x operand of the node models
        return Operand(X, instance_stride=(self.N + 1) * self.nx, knot_stride=self.nx, element_stride=1)

    def _evaluate_nodes(self, X, U, p_dyn, p_cost, p_ineq, w, derivatives: bool, stream):
        N, B, nx, nu, n = self.N, self.batch, self.nx, self.nu, self.nx + self.nu
        count = B * N
        xo, uo = self._states(X), _node(U, nu, N)
        wo = None if w is None else _node(w, w.shape[-1], N)
        par = lambda t, m: None if m.np == 0 else Operand.per_instance(t, m.np, shared=t.dim() == 1)  # noqa: E731
        wd = wo if self.dyn.nw else None
        if derivatives and self._wide is not None:
            # wide Jacobian (the lane-per-leg ANYmal kernel is 5.8x faster on unit-fastest operands): transpose (x, u) in, evaluate on
            # unit-fastest scratch, transpose (f, J) out into the node-major blocks the Riccati solve reads
            from . import transpose_nodes
            sx, su, sf, sJ = self._wide
            transpose_nodes(X, sx, B * (N + 1), nx, (nx, 1), (1, sx.stride(0)), stream=stream)  # all N + 1 knots: column b (N + 1) + k
            transpose_nodes(U, su, count, nu, (nu, 1), (1, su.stride(0)), stream=stream)
            st = sJ.stride(0)
            self.dyn.dense_jacobian(count, Operand(sx, instance_stride=N + 1, knot_stride=1, element_stride=sx.stride(0)), Operand.soa(su, su.stride(0), N), wd,
                                    par(p_dyn, self.dyn), Operand.soa(sf, st, N), Operand.soa(sJ, st, N), knots=N, stream=stream)
            transpose_nodes(sf, self.f, count, nx, (1, st), (nx, 1), stream=stream)
            # [A|B] is transposed into node-major blocks for the Riccati recursion (0.31 ms per 81 920 nodes).  Reading it IN PLACE through the
            # element stride (the recursion's LDS-DMA copies take 8-byte elements from any address; `jacobian_in_place=True`, identical
            # bits) was measured and is slower: every element of a knot then comes from its own cache line, shared only with the same
            # instance's seven neighbouring knots, which the 512 concurrent instances evict before they are used -- QP step 6.29 ms against
            # 4.70 ms through the transpose (profiles/archive/r03f_jacobian_in_place_ab.log).
            if self._transpose_jacobian:
                transpose_nodes(sJ, self.J, count, nx * n, (1, st), (nx * n, 1), stream=stream)
        elif derivatives:
            self.dyn.dense_jacobian(count, xo, uo, wd, par(p_dyn, self.dyn), _node(self.f, nx, N), _node(self.J, nx * n, N), knots=N, stream=stream)
        if derivatives:
            self.cost.sparse_hessian(count, xo, uo, wo if self.cost.nw else None, par(p_cost, self.cost), _node(self.c, 1, N), _node(self.cgrad, n, N),
                                     _node(self.chess, self.chess.shape[-1], N), knots=N, stream=stream)
            if self.ineq is not None:
                self.ineq.dense_jacobian(count, xo, uo, wo if self.ineq.nw else None, par(p_ineq, self.ineq), _node(self.h, self.nh, N), _node(self.hJ, self.nh * n, N),
                                         knots=N, stream=stream)
        else:
            self.dyn.forward_zero(count, xo, uo, wd, par(p_dyn, self.dyn), _node(self.f, nx, N), knots=N, stream=stream)
            self.cost.forward_zero(count, xo, uo, wo if self.cost.nw else None, par(p_cost, self.cost), _node(self.c, 1, N), knots=N, stream=stream)
            if self.ineq is not None:
                self.ineq.forward_zero(count, xo, uo, wo if self.ineq.nw else None, par(p_ineq, self.ineq), _node(self.h, self.nh, N), knots=N, stream=stream)

    def _merit(self, X, xm, theta, phi, with_slope, stream):
        nx, nu, N, B, n = self.nx, self.nu, self.N, self.batch, self.nx + self.nu
        a = _MeritArgs(nx, nu, N, B, self.nh, _node(X, nx, N + 1)._c(), _inst(xm, nx)._c(), _node(self.f, nx, N)._c(), _node(self.c, 1, N)._c(), _NULL,
                       _node(self.h, self.nh, N)._c() if self.nh else _NULL, self.barrier, self.multiplier,
                       _node(self.cgrad, n, N)._c() if with_slope else _NULL, _NULL, _node(self.dX, nx, N + 1)._c() if with_slope else _NULL,
                       _node(self.dU, nu, N)._c() if with_slope else _NULL, theta.data_ptr(), phi.data_ptr(), self.slope.data_ptr() if with_slope else None)
        _check(self.lib.ungar_ocp_merit(ctypes.byref(a), stream))

    # -- one iteration ------------------------------------------------------------------------------------------------
    def qp_step(self, X, U, xm, p_dyn=None, p_cost=None, p_ineq=None, w=None, stream=None):
        """Linearise at (X, U) and solve the QP: fills self.dX / self.dU (soft_sqp.hpp:143-158, SolveLocalQP)."""
        stream = NodeModel._stream(stream)
        nx, nu, N, B, n = self.nx, self.nu, self.N, self.batch, self.nx + self.nu
        self._evaluate_nodes(X, U, p_dyn, p_cost, p_ineq, w, True, stream)
        a = _StageQpArgs(nx, nu, N, B, _node(X, nx, N + 1)._c(), _inst(xm, nx)._c(), _node(self.f, nx, N)._c(), _node(self.cgrad, n, N)._c(),
                         _node(self.chess, self.chess.shape[-1], N)._c(), self._hes_rows.ctypes.data, self._hes_cols.ctypes.data, len(self._hes_rows), self.nh,
                         _node(self.h, self.nh, N)._c() if self.nh else _NULL, _node(self.hJ, self.nh * n, N)._c() if self.nh else _NULL, self.barrier,
                         _node(self.b, nx, N)._c(), _node(self.W, n * n, N)._c(), _node(self.w, n, N)._c(), _inst(self.dx0, nx)._c())
        _check(self.lib.ungar_ocp_stage_qp(ctypes.byref(a), stream))
        if self._wide is not None and not self._transpose_jacobian:
            sJ = self._wide[3]
            jac = Operand(sJ, instance_stride=N, knot_stride=1, element_stride=sJ.stride(0))  # node (b, k) = column b N + k of the unit-fastest scratch
        else:
            jac = _node(self.J, nx * n, N)
        q = _Qp(nx, nu, N, B, jac._c(), _node(self.b, nx, N)._c(), _node(self.W, n * n, N)._c(), _node(self.w, n, N)._c(), _NULL, _NULL,
                _inst(self.dx0, nx)._c(), _node(self.dX, nx, N + 1)._c(), _node(self.dU, nu, N)._c(), self.workspace.data_ptr(), self.workspace.numel(),
                self.regularization, self.status.data_ptr())
        _check(self.lib.ungar_ocp_riccati_solve(ctypes.byref(q), stream))

    def candidate_steps(self):
        """Step sizes the backtracking search tries, largest first: 1, gamma, gamma^2, ... >= alpha_min (backtracking_line_search.hpp:116-151)."""
        steps, alpha = [], 1.0
        while alpha >= self.ls.alpha_min:
            steps.append(alpha)
            alpha *= self.ls.gamma_alpha
        return steps

    def _stacked_search(self, X, U, xm, p_dyn, p_cost, p_ineq, w, s):
        """All candidates at once: stacked trial points -> one launch per node model -> one merit launch -> one selection launch."""
        torch, nx, nu, N, B, n = self.torch, self.nx, self.nu, self.N, self.batch, self.nx + self.nu
        steps = self.candidate_steps()
        K = len(steps)
        if self._stack is None or self._stack["K"] != K:
            z = lambda *shape: torch.empty(shape, dtype=torch.float64, device="cuda")  # noqa: E731
            self._stack = {"K": K, "Xt": z(K * B, N + 1, nx), "Ut": z(K * B, N, nu), "f": z(K * B, N, nx), "c": z(K * B, N, 1),
                           "h": z(K * B, N, self.nh) if self.nh else None, "theta": z(K * B), "phi": z(K * B), "rep": {}}
        st = self._stack
        alphas = (ctypes.c_double * K)(*steps)
        Xo, Uo, dXo, dUo = _node(X, nx, N + 1)._c(), _node(U, nu, N)._c(), _node(self.dX, nx, N + 1)._c(), _node(self.dU, nu, N)._c()
        Xt, Ut = _node(st["Xt"], nx, N + 1)._c(), _node(st["Ut"], nu, N)._c()
        _check(self.lib.ungar_ocp_trial_points(nx, nu, N, B, ctypes.byref(Xo), ctypes.byref(Uo), ctypes.byref(dXo), ctypes.byref(dUo), alphas, K, ctypes.byref(Xt),
                                               ctypes.byref(Ut), s))
        # node values of the K * B stacked instances: per-instance parameters / node parameters are repeated per candidate
        def rep(t):  # cached per source tensor: the copies outlive the launches that read them and are not reallocated every iteration
            if t is None or t.dim() == 1:
                return t
            key = (t.data_ptr(), tuple(t.shape))
            hit = st["rep"].get(key)
            if hit is None:
                if len(st["rep"]) >= 8:
                    st["rep"].clear()
                hit = (t, torch.empty((K * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device))
                st["rep"][key] = hit
            with self._on(s):
                hit[1].view((K,) + tuple(t.shape)).copy_(t.unsqueeze(0).expand((K,) + tuple(t.shape)))
            return hit[1]
        count = K * B * N
        xo = Operand(st["Xt"], instance_stride=(N + 1) * nx, knot_stride=nx, element_stride=1)
        uo = _node(st["Ut"], nu, N)
        ws = rep(w)
        wo = None if ws is None else _node(ws, ws.shape[-1], N)
        par = lambda t, m: None if m.np == 0 else Operand.per_instance(rep(t), m.np, shared=t.dim() == 1)  # noqa: E731
        self.dyn.forward_zero(count, xo, uo, wo if self.dyn.nw else None, par(p_dyn, self.dyn), _node(st["f"], nx, N), knots=N, stream=s)
        self.cost.forward_zero(count, xo, uo, wo if self.cost.nw else None, par(p_cost, self.cost), _node(st["c"], 1, N), knots=N, stream=s)
        if self.ineq is not None:
            self.ineq.forward_zero(count, xo, uo, wo if self.ineq.nw else None, par(p_ineq, self.ineq), _node(st["h"], self.nh, N), knots=N, stream=s)
        a = _MeritArgs(nx, nu, N, K * B, self.nh, Xt, _inst(xm, nx)._c(), _node(st["f"], nx, N)._c(), _node(st["c"], 1, N)._c(), _NULL,
                       _node(st["h"], self.nh, N)._c() if self.nh else _NULL, self.barrier, self.multiplier, _NULL, _NULL, _NULL, _NULL, st["theta"].data_ptr(),
                       st["phi"].data_ptr(), None)
        _check(self.lib.ungar_ocp_merit_stacked(ctypes.byref(a), B, s))
        params = _LineSearchParameters(self.ls.alpha_min, self.ls.theta_min, self.ls.theta_max, self.ls.eta, self.ls.gamma_phi, self.ls.gamma_theta, self.ls.gamma_alpha)
        _check(self.lib.ungar_ocp_line_search_select(nx, nu, N, B, ctypes.byref(params), alphas, K, self.theta0.data_ptr(), self.phi0.data_ptr(), self.slope.data_ptr(),
                                                     st["theta"].data_ptr(), st["phi"].data_ptr(), self.accepted.data_ptr(), ctypes.byref(Xo), ctypes.byref(Uo),
                                                     ctypes.byref(Xt), ctypes.byref(Ut), self.status.data_ptr(), s))
        return self.accepted

    def iterate(self, X, U, xm, p_dyn=None, p_cost=None, p_ineq=None, w=None, stream=None, stacked: bool = True):
        """One SQP iteration in place on (X, U): QP step, then the backtracking line search of the reference on
        phi = cost + barrier, theta = c |g| (soft_sqp.hpp:68-87).  Returns the per-instance accepted step sizes (device tensor,
        0 = no acceptable step: that instance was left unchanged, the reference's `break`).  An instance whose QP could not be solved
        (self.status != 0: reduced input Hessian not positive definite) takes no step either -- the reference asserts there.

        stacked (default): all candidate steps are evaluated as ONE stacked batch of candidates x batch trial points -- one launch per
        node model, one merit launch, one selection launch instead of six small launches per candidate; with thousands of instances
        every candidate is needed by somebody, so no work is added, only launches removed.  stacked=False walks the candidates one
        after the other (same result; candidates x less memory)."""
        s = NodeModel._stream(stream)
        nx, nu, N, B = self.nx, self.nu, self.N, self.batch
        self.qp_step(X, U, xm, p_dyn, p_cost, p_ineq, w, stream)
        self._merit(X, xm, self.theta0, self.phi0, True, s)  # node values at (X, U) are still in f / c / h
        if stacked and self._stackable:
            return self._stacked_search(X, U, xm, p_dyn, p_cost, p_ineq, w, s)
        with self._on(s):
            self.accepted.zero_()
        params = _LineSearchParameters(self.ls.alpha_min, self.ls.theta_min, self.ls.theta_max, self.ls.eta, self.ls.gamma_phi, self.ls.gamma_theta, self.ls.gamma_alpha)
        Xo, Uo, Xt, Ut = (_node(X, nx, N + 1)._c(), _node(U, nu, N)._c(), _node(self.Xt, nx, N + 1)._c(), _node(self.Ut, nu, N)._c())
        dXo, dUo = _node(self.dX, nx, N + 1)._c(), _node(self.dU, nu, N)._c()
        alpha = 1.0
        while alpha >= self.ls.alpha_min:
            _check(self.lib.ungar_ocp_trial_point(nx, nu, N, B, ctypes.byref(Xo), ctypes.byref(Uo), ctypes.byref(dXo), ctypes.byref(dUo), alpha, ctypes.byref(Xt),
                                                  ctypes.byref(Ut), s))
            self._evaluate_nodes(self.Xt, self.Ut, p_dyn, p_cost, p_ineq, w, False, s)
            self._merit(self.Xt, xm, self.thetaT, self.phiT, False, s)
            _check(self.lib.ungar_ocp_line_search_accept(nx, nu, N, B, ctypes.byref(params), alpha, self.theta0.data_ptr(), self.phi0.data_ptr(), self.slope.data_ptr(),
                                                         self.thetaT.data_ptr(), self.phiT.data_ptr(), self.accepted.data_ptr(), ctypes.byref(Xo), ctypes.byref(Uo),
                                                         ctypes.byref(Xt), ctypes.byref(Ut), self.status.data_ptr(), s))
            alpha *= self.ls.gamma_alpha
        return self.accepted
