"""Batched soft SQP on the device (reference include/ungar/optimization/soft_sqp.hpp:62-158,
186-264 and backtracking_line_search.hpp:80-165).

  CPU   the Riccati recursion of ungar_amd/csrc/kernels/ocp_riccati.hpp -- the source the gfx950 kernel runs -- executed by a
        sequential host policy (tests/cpp/riccati_host.cpp, test infrastructure) against a dense solve of the KKT system the
        reference hands to OSQP; indefinite reduced Hessians are reported, not silently used;
  GPU   the device kernel against the same dense solves; the stage-QP assembly and the merit terms against numpy on the
        node kernels' own outputs; ONE FULL SQP ITERATION (derivatives -> QP data -> Riccati -> backtracking line search)
        for 4096 instances of the quadrotor OCP (N = 30, rotor-speed bounds behind the POLY barrier) against an independent
        numpy restatement of SoftSQPOptimizer::Optimize built on the torch oracle's derivatives, to 1e-9; several iterations
        drive the constraint violation of every instance down; the same for the quadruped (SRBD) OCP with its friction-cone rows.
"""
import ctypes
import json
import os
import subprocess
import time

import numpy as np
import pytest

from oracle import ungar_oracle as O


# ------------------------------------------------------------------------------------------------ numpy reference pieces
def kkt_dense(nx, nu, N, AB, b, W, w, dx0, reg, WN=None, wN=None, E=None, e=None):
    """Dense solve of  min 1/2 d^T H d + g^T d  s.t.  dx_0 = dx0, dx_{k+1} = A dx_k + B du_k + b_k  (one instance); optional stage
    equality rows E_k [dx_k; du_k] + e_k = 0 (rows that are identically zero are dropped, as OSQP's l = A x = u leaves them inert)."""
    n, nz = nx + nu, (N + 1) * nx + N * nu
    H, g = np.zeros((nz, nz)), np.zeros(nz)
    xs = lambda k: np.arange(k * nx, (k + 1) * nx)  # noqa: E731
    us = lambda k: np.arange((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)  # noqa: E731
    for k in range(N):
        Wk = np.triu(W[k])
        Wk = Wk + Wk.T - np.diag(np.diag(Wk)) + reg * np.eye(n)
        idx = np.r_[xs(k), us(k)]
        H[np.ix_(idx, idx)] += Wk
        g[idx] += w[k]
    WNs = np.zeros((nx, nx)) if WN is None else np.triu(WN) + np.triu(WN, 1).T
    H[np.ix_(xs(N), xs(N))] += WNs + reg * np.eye(nx)
    if wN is not None:
        g[xs(N)] += wN
    m = (N + 1) * nx
    A, c = np.zeros((m, nz)), np.zeros(m)
    A[:nx, xs(0)] = np.eye(nx)
    c[:nx] = dx0
    for k in range(N):
        r = np.arange((k + 1) * nx, (k + 2) * nx)
        A[np.ix_(r, xs(k + 1))] = np.eye(nx)
        A[np.ix_(r, xs(k))] = -AB[k][:, :nx]
        A[np.ix_(r, us(k))] = -AB[k][:, nx:]
        c[r] = b[k]
    if E is not None:
        rows, vals = [], []
        for k in range(N):
            for j in range(E[k].shape[0]):
                if np.any(E[k][j] != 0.0):
                    row = np.zeros(nz)
                    row[np.r_[xs(k), us(k)]] = E[k][j]
                    rows.append(row)
                    vals.append(-e[k][j])
        if rows:
            A, c = np.vstack([A, np.array(rows)]), np.r_[c, vals]
            m = A.shape[0]
    sol = np.linalg.solve(np.block([[H, A.T], [A, np.zeros((m, m))]]), np.r_[-g, c])
    return sol[:(N + 1) * nx].reshape(N + 1, nx), sol[(N + 1) * nx:nz].reshape(N, nu)


def random_qp(rng, nx, nu, N, batch):
    n = nx + nu
    AB = rng.normal(size=(batch, N, nx, n)) * 0.3
    AB[:, :, :, :nx] += np.eye(nx)
    L = rng.normal(size=(batch, N, n, n))
    W = L @ np.swapaxes(L, -1, -2) * 0.1
    LN = rng.normal(size=(batch, nx, nx))
    return {"AB": np.ascontiguousarray(AB), "b": rng.normal(size=(batch, N, nx)) * 0.1, "W": np.ascontiguousarray(np.triu(W)), "Wfull": W,
            "w": rng.normal(size=(batch, N, n)), "WN": np.ascontiguousarray(np.triu(LN @ np.swapaxes(LN, -1, -2))), "wN": rng.normal(size=(batch, nx)),
            "dx0": rng.normal(size=(batch, nx))}


def poly_barrier(z, k, eps, order=0):
    a1, b1 = k, -0.5 * k * eps
    c1 = -(1.0 / 3.0) * (-b1 - a1 * eps) * eps - 0.5 * a1 * eps * eps - b1 * eps
    a2 = (-b1 - a1 * eps) / eps ** 2
    if order == 0:
        return np.where(z < 0, 0.5 * a1 * z * z + b1 * z + c1, np.where(z < eps, a2 * z ** 3 / 3 + 0.5 * a1 * z * z + b1 * z + c1, 0.0))
    if order == 1:
        return np.where(z < 0, a1 * z + b1, np.where(z < eps, a2 * z * z + a1 * z + b1, 0.0))
    return np.where(z < 0, a1, np.where(z < eps, 2 * a2 * z + a1, 0.0))


# ------------------------------------------------------------------------------------------------ CPU
@pytest.fixture(scope="module")
def host(repo_root):
    lib = os.path.join(repo_root, "build", "libriccati_host.so")
    src = os.path.join(repo_root, "tests", "cpp", "riccati_host.cpp")
    hdr = os.path.join(repo_root, "ungar_amd", "csrc", "kernels", "ocp_riccati.hpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.run(["g++", "-std=c++20", "-O2", "-shared", "-fPIC", "-o", lib, src], check=True, timeout=300)
    return ctypes.CDLL(lib)


def solve_host(lib, nx, nu, N, q, reg, terminal=True):
    batch = q["AB"].shape[0]
    dX, dU, st = np.zeros((batch, N + 1, nx)), np.zeros((batch, N, nu)), np.zeros(batch, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    lib.riccati_host_solve(nx, nu, N, ctypes.c_longlong(batch), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(q["WN"]) if terminal else None,
                           p(q["wN"]) if terminal else None, p(q["dx0"]), ctypes.c_double(reg), p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return dX, dU, st


@pytest.mark.parametrize("nx,nu,N", [(13, 4, 30), (6, 2, 20), (13, 24, 8), (3, 1, 5), (37, 12, 6), (2, 6, 7)])  # the last one: more inputs than nx^2
def test_riccati_recursion_equals_the_dense_kkt_solve(host, nx, nu, N):
    rng = np.random.default_rng(nx * 100 + nu)
    q = random_qp(rng, nx, nu, N, 3)
    for terminal in (True, False):
        dX, dU, st = solve_host(host, nx, nu, N, q, 1e-6, terminal)
        assert (st == 0).all()
        for i in range(3):
            rX, rU = kkt_dense(nx, nu, N, q["AB"][i], q["b"][i], q["Wfull"][i], q["w"][i], q["dx0"][i], 1e-6, q["WN"][i] if terminal else None,
                               q["wN"][i] if terminal else None)
            assert np.abs(dX[i] - rX).max() <= 1e-10 * max(1.0, np.abs(rX).max()) and np.abs(dU[i] - rU).max() <= 1e-10 * max(1.0, np.abs(rU).max())
            # the solution satisfies the linearised dynamics exactly (feasibility of the QP's equality constraints)
            for k in range(N):
                assert np.abs(dX[i, k + 1] - q["AB"][i, k] @ np.r_[dX[i, k], dU[i, k]] - q["b"][i, k]).max() < 1e-10 * max(1.0, np.abs(dX[i]).max())


def solve_host_eq(lib, variant, nx, nu, ne, N, q, reg, terminal_ld=0):
    batch = q["AB"].shape[0]
    dX, dU, st = np.zeros((batch, N + 1, nx)), np.zeros((batch, N, nu)), np.zeros(batch, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    WN = q["WN"]
    if terminal_ld:  # the terminal block as the leading nx x nx part of a wider row-major block
        wide = np.zeros((batch, nx, terminal_ld))
        wide[:, :, :nx] = WN
        WN = np.ascontiguousarray(wide)
    rc = lib.riccati_host_solve_eq(variant, nx, nu, ne, N, ctypes.c_longlong(batch), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(WN), terminal_ld, p(q["wN"]), p(q["dx0"]),
                                   p(q["E"]), p(q["e"]), ctypes.c_double(reg), p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return dX, dU, st


def random_eq_rows(rng, nx, nu, ne, N, batch, zero_fraction=0.3):
    """Stage equality rows with a state part and an input part of full row rank; a fraction of the rows is identically zero
    (an inactive contact: coefficient 0 times the Jacobian row, quadruped.example.cpp:301-302)."""
    n = nx + nu
    E = rng.normal(size=(batch, N, ne, n))
    e = rng.normal(size=(batch, N, ne)) * 0.1
    off = rng.random(size=(batch, N, ne)) < zero_fraction
    E[off] = 0.0
    e[off] = 0.0
    return np.ascontiguousarray(E), np.ascontiguousarray(e)


@pytest.mark.parametrize("nx,nu,ne,N", [(5, 4, 2, 6), (13, 24, 12, 5), (25, 24, 16, 4), (7, 3, 3, 9)])
def test_riccati_with_stage_equality_rows_equals_the_dense_kkt_solve(host, nx, nu, ne, N):
    """Stage equality rows E_k [dx; du] + e_k = 0 (the foot-contact rows of the reference's quadruped OCP, quadruped.example.cpp:279-304, are
    hard equalities of its QP, soft_sqp.hpp:155-157): the recursion eliminates the stage KKT block [R D^T; D 0] and must reproduce the dense
    solve of the whole KKT system, rows that are identically zero included; the terminal block may be the corner of a wider one."""
    rng = np.random.default_rng(1000 * nx + 10 * nu + ne)
    q = random_qp(rng, nx, nu, N, 3)
    q["E"], q["e"] = random_eq_rows(rng, nx, nu, ne, N, 3)
    for variant, ld in ((0, 0), (1, nx + nu)):
        dX, dU, st = solve_host_eq(host, variant, nx, nu, ne, N, q, 1e-6, ld)
        assert (st == 0).all()
        for i in range(3):
            rX, rU = kkt_dense(nx, nu, N, q["AB"][i], q["b"][i], q["Wfull"][i], q["w"][i], q["dx0"][i], 1e-6, q["WN"][i], q["wN"][i], q["E"][i], q["e"][i])
            scale = max(1.0, np.abs(rX).max(), np.abs(rU).max())
            assert np.abs(dX[i] - rX).max() <= 2e-9 * scale and np.abs(dU[i] - rU).max() <= 2e-9 * scale
            for k in range(N):  # the rows hold along the solution
                r = q["E"][i, k] @ np.r_[dX[i, k], dU[i, k]] + q["e"][i, k]
                assert np.abs(r).max() <= 1e-9 * scale


@pytest.mark.parametrize("nx,nu,ne", [(17, 4, 0), (8, 2, 0), (25, 24, 16), (25, 24, 0)])
def test_fixed_size_instantiations_with_carried_quantities(host, nx, nu, ne):
    """The instantiations for the reference's OCPs once the carried quantities (previous input, previous foot positions) are part of
    the stage state -- 17 + 4, 8 + 2, 25 + 24 with 16 contact rows -- agree with the run-time-sized recursion, also under both extreme
    schedules of the asynchronous operand copies."""
    rng = np.random.default_rng(77 + nx)
    N = 5
    q = random_qp(rng, nx, nu, N, 2)
    q["E"], q["e"] = random_eq_rows(rng, nx, nu, max(ne, 1), N, 2)  # (ne == 0: the arrays are not read)
    ref = solve_host_eq(host, 0, nx, nu, ne, N, q, 1e-6)
    assert (ref[2] == 0).all()
    fixed = solve_host_eq(host, 2, nx, nu, ne, N, q, 1e-6)
    scale = max(1.0, np.abs(ref[0]).max(), np.abs(ref[1]).max())
    assert np.abs(fixed[0] - ref[0]).max() <= 1e-9 * scale and np.abs(fixed[1] - ref[1]).max() <= 1e-9 * scale
    for variant in (3, 4):
        out = solve_host_eq(host, variant, nx, nu, ne, N, q, 1e-6)
        assert np.abs(out[0] - fixed[0]).max() <= 1e-9 * scale and np.abs(out[1] - fixed[1]).max() <= 1e-9 * scale
    a, b = solve_host_eq(host, 3, nx, nu, ne, N, q, 1e-6), solve_host_eq(host, 4, nx, nu, ne, N, q, 1e-6)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_prefetching_variant_is_bitwise_the_same_recursion(host):
    """The device kernel stages the next knot's operands in registers for small problems; the staged and the in-place variant
    of the shared source must produce identical bits."""
    rng = np.random.default_rng(9)
    nx, nu, N = 13, 4, 12
    q = random_qp(rng, nx, nu, N, 2)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    out = []
    for variant in (0, 1):
        dX, dU, st = np.zeros((2, N + 1, nx)), np.zeros((2, N, nu)), np.zeros(2, dtype=np.int32)
        host.riccati_host_solve_variant(variant, nx, nu, N, ctypes.c_longlong(2), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(q["WN"]), p(q["wN"]), p(q["dx0"]),
                                        ctypes.c_double(1e-6), p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        out.append((dX, dU))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("nx,nu", [(13, 4), (6, 2), (37, 12), (13, 24)])
def test_fixed_size_instantiation_matches_the_generic_recursion(host, nx, nu):
    """For the reference's OCP sizes the device launches an instantiation with (nx, nu) fixed at compile time, whose Cholesky of the
    nu x nu block runs in registers inside the solve phase (one phase instead of nu + 1; nu <= 8) and whose two large products use 2 x 4
    register tiles (nx >= 24): same recursion, same solution up to the
    rounding of the differently associated sums, and the indefinite-block report still works."""
    rng = np.random.default_rng(21)
    N = 9
    q = random_qp(rng, nx, nu, N, 3)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    out = []
    for variant in (0, 2):
        dX, dU, st = np.zeros((3, N + 1, nx)), np.zeros((3, N, nu)), np.zeros(3, dtype=np.int32)
        rc = host.riccati_host_solve_variant(variant, nx, nu, N, ctypes.c_longlong(3), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(q["WN"]), p(q["wN"]), p(q["dx0"]),
                                             ctypes.c_double(1e-6), p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert rc == 0 and (st == 0).all()
        out.append((dX, dU))
    scale = max(1.0, np.abs(out[0][0]).max(), np.abs(out[0][1]).max())
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * scale and np.abs(out[0][1] - out[1][1]).max() <= 1e-11 * scale
    q["W"][2, 5] = -np.eye(nx + nu) * 50.0  # knot 5 of instance 2: concave in the inputs
    dX, dU, st = np.zeros((3, N + 1, nx)), np.zeros((3, N, nu)), np.zeros(3, dtype=np.int32)
    host.riccati_host_solve_variant(2, nx, nu, N, ctypes.c_longlong(3), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(q["WN"]), p(q["wN"]), p(q["dx0"]), ctypes.c_double(1e-6),
                                    p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert st[0] == 0 and st[1] == 0 and st[2] == 6


@pytest.mark.parametrize("nx,nu", [(13, 4), (37, 12), (13, 24)])
def test_asynchronous_operand_copies_do_not_change_a_bit(host, nx, nu):
    """The four-wavefront device kernels request a knot's operands by LDS-DMA the moment their destination dies in the previous knot
    (the stage Hessian parked, folded, in the retired cost-to-go buffer) and run the forward pass three knots ahead.  Sequential
    stand-in for that protocol: copies that land at once and copies that land only when waited for must both reproduce the bits of
    the plain fixed-size recursion -- a destination that is still live, or read before its wait, would show."""
    rng = np.random.default_rng(33)
    N = 11
    q = random_qp(rng, nx, nu, N, 3)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    out = []
    for variant in (2, 3, 4):
        dX, dU, st = np.full((3, N + 1, nx), np.nan), np.full((3, N, nu), np.nan), np.zeros(3, dtype=np.int32)
        rc = host.riccati_host_solve_variant(variant, nx, nu, N, ctypes.c_longlong(3), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(q["WN"]), p(q["wN"]), p(q["dx0"]),
                                             ctypes.c_double(1e-6), p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert rc == 0 and (st == 0).all()
        out.append((dX, dU))
    for dX, dU in out[1:]:
        assert np.array_equal(out[0][0], dX) and np.array_equal(out[0][1], dU)


@pytest.mark.parametrize("nx,nu", [(13, 4), (6, 2)])
def test_counted_waits_of_the_single_wavefront_forward_pass(host, nx, nu):
    """The one-wavefront kernels run their forward pass three knots ahead on copies they issue themselves and await by COUNT (all but the
    copy instructions of the two younger knots).  Sequential stand-in: nothing lands before a wait demands it / everything lands at once;
    both must reproduce the bits of the plain fixed-size recursion -- a count that is off by one instruction would leave stale data."""
    rng = np.random.default_rng(44)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    for N in (1, 2, 3, 4, 9):  # shorter than, equal to and longer than the pipeline
        q = random_qp(rng, nx, nu, N, 2)
        out = []
        for variant in (2, 5, 6, 7, 8):  # 7 / 8: asynchronous copies in the backward pass as well (device variant "fixedq")
            dX, dU, st = np.full((2, N + 1, nx), np.nan), np.full((2, N, nu), np.nan), np.zeros(2, dtype=np.int32)
            rc = host.riccati_host_solve_variant(variant, nx, nu, N, ctypes.c_longlong(2), p(q["AB"]), p(q["b"]), p(q["W"]), p(q["w"]), p(q["WN"]), p(q["wN"]), p(q["dx0"]),
                                                 ctypes.c_double(1e-6), p(dX), p(dU), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
            assert rc == 0 and (st == 0).all()
            out.append((dX, dU))
        for dX, dU in out[1:]:
            assert np.array_equal(out[0][0], dX) and np.array_equal(out[0][1], dU), N


@pytest.mark.parametrize("n", [1, 2, 8, 17, 37, 49])
def test_folded_triangle_index_maps_are_inverse_bijections(host, n):
    """The upper triangle of a symmetric n x n block is parked in a ((n + 1) / 2) x (n + 1) rectangle (stage Hessian of the next knot,
    pairs of the cost-to-go update): every (r <= c) has its own cell, the source map inverts the index map, and the only cells without
    a source are the unused tail of the middle row when n is odd."""
    cells = ((n + 1) // 2) * (n + 1)
    seen = {}
    for r in range(n):
        for c in range(r, n):
            i = host.riccati_folded_index(n, r, c)
            assert 0 <= i < cells and i not in seen
            seen[i] = (r, c)
            assert host.riccati_folded_source(n, i) == r * n + c
    unused = [i for i in range(cells) if i not in seen]
    assert len(unused) == cells - n * (n + 1) // 2 == ((n + 1) // 2 if n % 2 else 0)
    assert all(host.riccati_folded_source(n, i) == -1 for i in unused)


def test_riccati_reports_an_indefinite_reduced_hessian(host):
    rng = np.random.default_rng(1)
    q = random_qp(rng, 4, 2, 6, 2)
    q["W"][1, 3] = -np.eye(6) * 50.0  # knot 3 of instance 1: concave in the inputs
    _, _, st = solve_host(host, 4, 2, 6, q, 1e-6)
    assert st[0] == 0 and st[1] == 4  # knot index + 1


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("nx,nu,N,batch", [(13, 4, 30, 257), (13, 24, 10, 33), (37, 12, 20, 9), (6, 2, 200, 16), (25, 24, 30, 17), (37, 12, 3, 1), (13, 24, 1, 65), (17, 4, 30, 70),
                                           (8, 2, 30, 257), (6, 2, 30, 131), (3, 1, 5, 7), (2, 6, 7, 5), (9, 2, 10, 6), (1, 1, 4, 3)])  # second line: the narrow blocks (LDS-resident kernels, run-time sizes)
def test_device_riccati_equals_the_dense_kkt_solve(nx, nu, N, batch):
    import torch
    from ungar_amd import sqp
    rng = np.random.default_rng(7)
    q = random_qp(rng, nx, nu, N, batch)
    dev = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
    dX, dU, st = sqp.riccati_solve(nx, nu, N, batch, dev(q["AB"]), dev(q["b"]), dev(q["W"]), dev(q["w"]), dev(q["dx0"]), dev(q["WN"]), dev(q["wN"]))
    torch.cuda.synchronize()
    assert (st == 0).all()
    dX, dU = dX.cpu().numpy(), dU.cpu().numpy()
    for i in sorted({0, batch // 2, batch - 1}):
        rX, rU = kkt_dense(nx, nu, N, q["AB"][i], q["b"][i], q["Wfull"][i], q["w"][i], q["dx0"][i], 1e-6, q["WN"][i], q["wN"][i])
        assert np.abs(dX[i] - rX).max() <= 1e-9 * max(1.0, np.abs(rX).max()) and np.abs(dU[i] - rU).max() <= 1e-9 * max(1.0, np.abs(rU).max())
    res = dX[:, 1:] - np.einsum("bkij,bkj->bki", q["AB"], np.concatenate((dX[:, :-1], dU), axis=2)) - q["b"]
    assert np.abs(res).max() <= 1e-9 * max(1.0, np.abs(dX).max())


@gpu
@pytest.mark.parametrize("nx,nu,N,batch", [(10, 3, 25, 130), (20, 9, 12, 67), (31, 30, 6, 9), (23, 17, 7, 33)])
def test_register_resident_riccati_for_sizes_the_library_was_not_compiled_for(nx, nu, N, batch, tmp_path, monkeypatch):
    """The reference's optimiser takes ANY problem (optimization/concepts.hpp:153-262); the register-resident recursion is a template over the stage sizes.  For sizes
    that are not compiled into the library the kernel factory instantiates it at run time (hipcc --genco, cached under the code-generation folder, occupancy picked
    from the compiled candidates): the route is reported as 2, the steps equal the dense KKT solve, the dynamics of the QP hold on the whole batch, and a second
    request finds the entry in the cache folder."""
    import ctypes
    import torch
    import ungar_amd
    from ungar_amd import sqp
    monkeypatch.setenv("UNGAR_CODEGEN_FOLDER", str(tmp_path))
    lib = ungar_amd.load_library()
    lib.ungar_ocp_riccati_route.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int32]
    assert lib.ungar_ocp_riccati_route(nx, nu, 0, 1) == 2, lib.ungar_last_error()
    entries = sorted(f for f in os.listdir(tmp_path / "ungar_amd_kernels") if f.startswith(f"riccati_wave_{nx}_{nu}_"))
    assert any(f.endswith(".hsaco") for f in entries) and any(f.endswith(".meta") for f in entries), entries
    waves, registers, scratch = (int(v) for v in open(tmp_path / "ungar_amd_kernels" / next(f for f in entries if f.endswith(".meta"))).read().split("\n")[1].split()[1:4])
    assert waves in (1, 2, 4) and scratch == 0 and registers * waves <= 512, (waves, registers, scratch)
    rng = np.random.default_rng(29)
    q = random_qp(rng, nx, nu, N, batch)
    dev = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
    dX, dU, st = sqp.riccati_solve(nx, nu, N, batch, dev(q["AB"]), dev(q["b"]), dev(q["W"]), dev(q["w"]), dev(q["dx0"]), dev(q["WN"]), dev(q["wN"]))
    torch.cuda.synchronize()
    assert (st == 0).all()
    dX, dU = dX.cpu().numpy(), dU.cpu().numpy()
    for i in sorted({0, batch // 2, batch - 1}):
        rX, rU = kkt_dense(nx, nu, N, q["AB"][i], q["b"][i], q["Wfull"][i], q["w"][i], q["dx0"][i], 1e-6, q["WN"][i], q["wN"][i])
        assert np.abs(dX[i] - rX).max() <= 1e-9 * max(1.0, np.abs(rX).max()) and np.abs(dU[i] - rU).max() <= 1e-9 * max(1.0, np.abs(rU).max())
    res = dX[:, 1:] - np.einsum("bkij,bkj->bki", q["AB"], np.concatenate((dX[:, :-1], dU), axis=2)) - q["b"]
    assert np.abs(res).max() <= 1e-9 * max(1.0, np.abs(dX).max())


def test_kernel_factory_builds_and_caches_without_a_gpu(tmp_path, monkeypatch):
    """The kernel factory on a build host (UNGAR_AMD_COMPILE_ONLY: hipcc cross-compiles gfx950 without a device): a miss compiles the occupancy candidates side by
    side and publishes {meta, code object}; the second request is a hit; sizes the template does not fit, or that the LDS-resident kernels serve better, are
    answered without compiling anything."""
    import ctypes
    import ungar_amd
    monkeypatch.setenv("UNGAR_CODEGEN_FOLDER", str(tmp_path))
    monkeypatch.setenv("UNGAR_AMD_COMPILE_ONLY", "1")
    lib = ungar_amd.load_library()
    lib.ungar_ocp_riccati_route.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int32]
    lib.ungar_shooting_assemble_route.argtypes = [ctypes.c_int64] * 4 + [ctypes.c_int32]
    assert lib.ungar_ocp_riccati_route(37, 12, 0, 1) == 1 and lib.ungar_ocp_riccati_route(6, 2, 0, 1) == 0 and lib.ungar_ocp_riccati_route(40, 30, 0, 1) == 0
    assert lib.ungar_ocp_riccati_route(13, 4, 3, 1) == 0  # equality rows inside the recursion: the LDS-resident kernels
    assert not os.path.exists(tmp_path / "ungar_amd_kernels")
    assert lib.ungar_ocp_riccati_route(11, 5, 0, 1) == 2, lib.ungar_last_error()
    first = sorted(os.listdir(tmp_path / "ungar_amd_kernels"))
    assert sum(f.endswith(".hsaco") for f in first) == 1 and not any(".tmp" in f for f in first), first
    stamp = os.path.getmtime(tmp_path / "ungar_amd_kernels" / next(f for f in first if f.endswith(".hsaco")))
    assert lib.ungar_ocp_riccati_route(11, 5, 0, 1) == 2
    assert sorted(os.listdir(tmp_path / "ungar_amd_kernels")) == first and os.path.getmtime(tmp_path / "ungar_amd_kernels" / next(f for f in first if f.endswith(".hsaco"))) == stamp
    assert lib.ungar_shooting_assemble_route(25, 24, 16, 12, 1) == 1 and lib.ungar_shooting_assemble_route(17, 4, 0, 8, 1) == 3 and lib.ungar_shooting_assemble_route(60, 30, 4, 0, 1) == 0
    assert lib.ungar_shooting_assemble_route(9, 4, 3, 2, 1) == 2, lib.ungar_last_error()
    assert any(f.startswith("shooting_assemble_wave_9_4_3_") and f.endswith(".hsaco") for f in os.listdir(tmp_path / "ungar_amd_kernels"))


@gpu
@pytest.mark.parametrize("nx,nu,N", [(37, 12, 20), (25, 24, 30), (13, 24, 30), (17, 4, 30), (13, 4, 30)])
def test_register_resident_riccati_kernels_agree_with_the_lds_resident_ones(nx, nu, N, monkeypatch, measurement_library):
    """The large blocks take the one-wavefront-per-instance kernels of ocp_riccati_wave.hip by default (every matrix of the recursion in registers, products
    chained on the FP64 matrix cores, homogeneous coordinates for the affine parts); UNGAR_AMD_RICCATI_VARIANT keeps the LDS-resident kernels of
    ocp_riccati.hip (the recursion that is pinned against the dense KKT solve on the host).  Same QPs, both routes: steps equal to 1e-11 of their scale, and
    the dynamics of the QP hold to rounding on the whole batch -- also through a strided view of [A|B] (the register-resident route needs contiguous knot
    blocks and must hand such a call over)."""
    import torch
    from ungar_amd import sqp
    batch = 64
    rng = np.random.default_rng(23)
    q = random_qp(rng, nx, nu, N, batch)
    dev = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
    args = [dev(q[k]) for k in ("AB", "b", "W", "w", "dx0", "WN", "wN")]
    monkeypatch.delenv("UNGAR_AMD_RICCATI_VARIANT", raising=False)
    dXw, dUw, stw = sqp.riccati_solve(nx, nu, N, batch, *args)
    monkeypatch.setenv("UNGAR_AMD_RICCATI_VARIANT", "fixed")
    dXl, dUl, stl = sqp.riccati_solve(nx, nu, N, batch, *args)
    torch.cuda.synchronize()
    assert (stw == 0).all() and (stl == 0).all()
    for w, l in ((dXw, dXl), (dUw, dUl)):
        assert float((w - l).abs().max()) <= 1e-11 * max(1.0, float(l.abs().max()))
    res = dXw[:, 1:].cpu().numpy() - np.einsum("bkij,bkj->bki", q["AB"], np.concatenate((dXw[:, :-1].cpu().numpy(), dUw.cpu().numpy()), axis=2)) - q["b"]
    assert np.abs(res).max() <= 1e-10 * max(1.0, float(dXw.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("nx,nu", [(13, 4), (13, 24), (37, 12), (4, 2), (25, 24), (17, 4)])
def test_device_riccati_reports_an_indefinite_reduced_hessian(nx, nu):
    """Every device instantiation (register Cholesky, L D L^T phases, run-time sizes; one and four wavefronts per instance) reports the
    knot whose input block is not positive definite for exactly the instance concerned, as the host policy does."""
    import torch
    from ungar_amd import sqp
    rng = np.random.default_rng(3)
    N, batch = 7, 5
    q = random_qp(rng, nx, nu, N, batch)
    q["W"][3, 4] = -np.eye(nx + nu) * 50.0  # knot 4 of instance 3: concave in the inputs
    dev = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
    _, _, st = sqp.riccati_solve(nx, nu, N, batch, dev(q["AB"]), dev(q["b"]), dev(q["W"]), dev(q["w"]), dev(q["dx0"]), dev(q["WN"]), dev(q["wN"]))
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [0, 0, 0, 5, 0]


def _quadrotor_problem(batch, N, seed, torch):
    """Random-but-reasonable instances of the quadrotor OCP: states near hover, references 1 m away, inputs around hover speed."""
    rng = np.random.default_rng(seed)
    hover = np.sqrt(1.5 * 9.80665 / (4 * 0.015))
    quat = rng.normal(size=(batch, N + 1, 4)) * 0.1 + np.array([0, 0, 0, 1.0])
    quat /= np.linalg.norm(quat, axis=2, keepdims=True)
    X = np.concatenate((rng.uniform(-0.5, 0.5, (batch, N + 1, 3)), quat, rng.uniform(-0.3, 0.3, (batch, N + 1, 6))), axis=2)
    U = hover * rng.uniform(0.7, 1.3, (batch, N, 4))
    U[:, ::5, 0] = 1.05 * 2.0 * hover  # some rotors beyond r_max = 2 hover: active barrier rows
    xm = X[:, 0] + rng.normal(size=(batch, 13)) * 0.02
    qref = rng.normal(size=(batch, 4)) * 0.1 + np.array([0, 0, 0, 1.0])
    p_cost = np.concatenate((rng.uniform(-1, 1, (batch, 3)), qref / np.linalg.norm(qref, axis=1, keepdims=True), np.zeros((batch, 6))), axis=1)
    p_dyn = np.tile(O.default_params("quadrotor"), (batch, 1))
    p_dyn[:, 0] = 1.0 / N
    p_ineq = np.full((batch, 1), 2.0 * hover)
    return X, U, xm, p_dyn, p_cost, p_ineq


def _c_checker_node_jacobian(name, X, U, p):
    """(f, dense [A|B]) of the nodes (X[k], U[k]) from the oracle's generated C (oracle/_gen; pinned against the torch oracle's golden vectors by
    tests/test_codegen_c.py): two orders of magnitude faster than torch autograd for the full-body model."""
    import ctypes
    from oracle import build_oracle
    path = build_oracle.lib_path("portable")
    if not os.path.exists(path):
        pytest.skip("oracle C library not built: run __graft_entry__.build()")
    clib = ctypes.CDLL(path)
    m, nx, nu = X.shape[0], X.shape[1], U.shape[1]
    nnz = ctypes.c_int.in_dll(clib, f"{name}_jac_nnz").value
    rows = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_row")).astype(np.int64)
    cols = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_col")).astype(np.int64)
    dp = ctypes.POINTER(ctypes.c_double)
    loop = getattr(clib, f"{name}_sparse_jacobian_batch_shared")
    loop.argtypes = [dp] * 6 + [ctypes.c_long, ctypes.c_long]
    loop.restype = None
    xh, uh, ph, w0 = np.ascontiguousarray(X), np.ascontiguousarray(U), np.ascontiguousarray(p), np.zeros(1)
    f, v = np.empty((m, nx)), np.empty((m, nnz))
    loop(xh.ctypes.data_as(dp), uh.ctypes.data_as(dp), w0.ctypes.data_as(dp), ph.ctypes.data_as(dp), f.ctypes.data_as(dp), v.ctypes.data_as(dp), 0, m)
    J = np.zeros((m, nx, nx + nu))
    J[:, rows, cols] = v
    return f, J


def _reference_iteration(X, U, xm, p_dyn, p_cost, p_ineq, dyn, cost, N, k_barrier=100.0, eps=2e-5, mult=1.0, c_checker_jacobian=False):
    """One iteration of SoftSQPOptimizer::Optimize for ONE instance, numpy + the torch oracle (independent of the product).
    c_checker_jacobian: node Jacobians from the oracle's generated C instead of torch autograd (the full-body model: 7 s -> 0.1 s per instance)."""
    nx, nu = X.shape[1], U.shape[1]
    n = nx + nu
    w0 = np.zeros((N, 0))

    def evaluate(Xc, Uc, derivatives):
        pd, pc = np.tile(p_dyn, (N, 1)), np.tile(p_cost, (N, 1))
        if derivatives and c_checker_jacobian:
            f, J = _c_checker_node_jacobian(dyn, Xc[:N], Uc, p_dyn)
            c, g, H = O.cost_value_gradient_hessian(Xc[:N], Uc, pc, name=cost)
        elif derivatives:
            f, J = O.node_jacobian(dyn, Xc[:N], Uc, w0, pd)
            c, g, H = O.cost_value_gradient_hessian(Xc[:N], Uc, pc, name=cost)
        else:
            f, J = O.node_value(dyn, Xc[:N], Uc, w0, pd), None
            c, g, H = O.cost_value_gradient_hessian(Xc[:N], Uc, pc, name=cost)[0], None, None
        if p_ineq is None:
            h = np.zeros((N, 0))  # no inequality rows (the full-body quadruped problem below)
        else:
            h = np.stack((Uc - p_ineq[0], -Uc), axis=2).reshape(N, 2 * nu)  # [r - r_max, -r] per rotor
        return f, J, c, g, H, h

    def merit(Xc, f, c, h):
        gres = np.concatenate((Xc[0] - xm, (Xc[1:] - f).reshape(-1)))
        return mult * np.sqrt((gres ** 2).sum()), c.sum() + poly_barrier(-h, k_barrier, eps).sum()

    f, J, c, g, H, h = evaluate(X, U, True)
    Jh = np.zeros((N, 2 * nu if p_ineq is not None else 0, n))
    for i in range(nu if p_ineq is not None else 0):
        Jh[:, 2 * i, nx + i] = 1.0
        Jh[:, 2 * i + 1, nx + i] = -1.0
    d1, d2 = poly_barrier(-h, k_barrier, eps, 1), poly_barrier(-h, k_barrier, eps, 2)
    W = H + np.einsum("kja,kj,kjb->kab", Jh, d2, Jh)
    w = g - np.einsum("kja,kj->ka", Jh, d1)
    dX, dU = kkt_dense(nx, nu, N, J, f - X[1:], W, w, xm - X[0], 1e-6)
    theta, phi = merit(X, f, c, h)
    slope = (g * np.concatenate((dX[:N], dU), axis=1)).sum()
    alpha = 1.0
    while alpha >= 1e-4:
        Xt, Ut = X + alpha * dX, U + alpha * dU
        ft, _, ct, _, _, ht = evaluate(Xt, Ut, False)
        tn, pn = merit(Xt, ft, ct, ht)
        if tn > 1e-2:
            ok = tn < (1 - 1e-6) * theta
        elif max(theta, tn) < 1e-6 and slope < 0:
            ok = pn < phi + 1e-4 * alpha * slope
        else:
            ok = pn < (1 - 1e-6) * phi or tn < (1 - 1e-6) * theta
        if ok:
            return dX, dU, alpha, Xt, Ut, (theta, phi, slope)
        alpha *= 0.5
    return dX, dU, 0.0, X, U, (theta, phi, slope)


@gpu
def test_one_sqp_iteration_quadrotor_4096_instances(repo_root):
    import torch
    from ungar_amd import sqp
    batch, N = 4096, 30
    X, U, xm, p_dyn, p_cost, p_ineq = _quadrotor_problem(batch, N, 3, torch)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    Xd, Ud, xmd, pd, pc, pi = dev(X), dev(U), dev(xm), dev(p_dyn), dev(p_cost), dev(p_ineq)
    solver = sqp.BatchedSoftSqp("quadrotor", "quadrotor_cost", N, batch, inequality="quadrotor_ineq")
    accepted = solver.iterate(Xd, Ud, xmd, pd, pc, pi)
    torch.cuda.synchronize()
    assert (solver.status == 0).all()
    acc = accepted.cpu().numpy()
    dXd, dUd, Xn, Un = solver.dX.cpu().numpy(), solver.dU.cpu().numpy(), Xd.cpu().numpy(), Ud.cpu().numpy()
    th0, ph0, sl = solver.theta0.cpu().numpy(), solver.phi0.cpu().numpy(), solver.slope.cpu().numpy()
    assert (acc > 0).mean() > 0.95  # nearly every instance finds an acceptable step
    for i in (0, 1, 777, 2048, 4095):
        dX, dU, alpha, Xr, Ur, (theta, phi, slope) = _reference_iteration(X[i], U[i], xm[i], p_dyn[i], p_cost[i], p_ineq[i], "quadrotor", "quadrotor_cost", N)
        scale = max(1.0, np.abs(dX).max(), np.abs(dU).max())
        assert np.abs(dXd[i] - dX).max() <= 1e-9 * scale and np.abs(dUd[i] - dU).max() <= 1e-9 * scale
        assert abs(th0[i] - theta) <= 1e-10 * max(1.0, theta) and abs(ph0[i] - phi) <= 1e-10 * max(1.0, abs(phi)) and abs(sl[i] - slope) <= 1e-9 * max(1.0, abs(slope))
        assert acc[i] == alpha
        assert np.abs(Xn[i] - Xr).max() <= 1e-9 * max(1.0, np.abs(Xr).max()) and np.abs(Un[i] - Ur).max() <= 1e-9 * max(1.0, np.abs(Ur).max())
    # the stacked line search (all candidates in one batch: the default) and the candidate-by-candidate loop give the same bits
    solver2 = sqp.BatchedSoftSqp("quadrotor", "quadrotor_cost", N, batch, inequality="quadrotor_ineq")
    X2, U2 = dev(X), dev(U)
    accepted2 = solver2.iterate(X2, U2, xmd, pd, pc, pi, stacked=False)
    torch.cuda.synchronize()
    assert torch.equal(accepted2, accepted) and torch.equal(X2, Xd) and torch.equal(U2, Ud)
    assert len(solver.candidate_steps()) == 14 and solver.candidate_steps()[-1] >= 1e-4
    # further iterations: the dynamics defect of every instance goes down by orders of magnitude
    theta_first = th0.copy()
    t0 = time.perf_counter()
    iterations = 6
    for _ in range(iterations):
        solver.iterate(Xd, Ud, xmd, pd, pc, pi)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iterations
    theta_last = solver.theta0.cpu().numpy()  # violation at the start of the last iteration
    # time of the QP part alone (derivatives + stage data + Riccati), stream-ordered
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        solver.qp_step(Xd, Ud, xmd, pd, pc, pi)
    torch.cuda.synchronize()
    qp_ms = (time.perf_counter() - t1) / 20 * 1e3
    os.makedirs(os.path.join(repo_root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(repo_root, "gpurun_out", "sqp_quadrotor_timing.json"), "w") as fh:
        json.dump({"workload": "quadrotor OCP N=30, 4096 instances, rotor bounds (POLY barrier)", "ms_per_sqp_iteration": dt * 1e3, "ms_per_qp_step": qp_ms,
                   "line_search_candidates": 14, "instances_per_s": batch / dt, "median_theta_first": float(np.median(theta_first)),
                   "median_theta_after_6_more_iterations": float(np.median(theta_last))}, fh)
    # the filter line search of the reference trades violation against cost, so the defect need not collapse in a few
    # iterations from a random (dynamically inconsistent) start; it must go down for (nearly) every instance
    assert torch.isfinite(Xd).all() and torch.isfinite(Ud).all()
    assert (theta_last < 0.6 * theta_first).mean() > 0.95


@gpu
def test_sqp_iterations_quadruped_srbd_with_friction_cones():
    """The single-rigid-body quadruped OCP (example/mpc/quadruped.example.cpp): dynamics 'srbd' with per-knot contact flags,
    stage cost 'srbd_cost', inequality rows 'srbd_ineq' -- 37 stage variables, 12 barrier rows per knot, 1024 instances."""
    import torch
    from ungar_amd import sqp
    batch, N = 1024, 30
    rng = np.random.default_rng(4)
    x, u, w, p = O.synthetic_inputs("srbd", batch * (N + 1), seed=11)
    X = x.reshape(batch, N + 1, 13)
    U = u[:batch * N].reshape(batch, N, 24)
    W = np.ascontiguousarray(w[:batch * N].reshape(batch, N, 4))
    p_dyn = np.tile(O.default_params("srbd"), (batch, 1))
    p_ineq = np.tile(O.default_params("srbd_ineq"), (batch, 1))
    _, _, ref = O.synthetic_cost_inputs(batch, seed=5, name="srbd_cost")
    xm = X[:, 0] + rng.normal(size=(batch, 13)) * 0.01
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    Xd, Ud, Wd, xmd, pd, pc, pi = dev(X), dev(U), dev(W), dev(xm), dev(p_dyn), dev(ref), dev(p_ineq)
    solver = sqp.BatchedSoftSqp("srbd", "srbd_cost", N, batch, inequality="srbd_ineq", constraint_violation_multiplier=1.0 / 30.0, stiffness=1.0, epsilon=1.0)
    solver.qp_step(Xd, Ud, xmd, pd, pc, pi, w=Wd)
    torch.cuda.synchronize()
    assert (solver.status == 0).all()
    # the step of instance 0 against the dense KKT solve of the QP data the kernels produced
    J, b, Wq, wq, dx0 = (t[0].cpu().numpy() for t in (solver.J, solver.b, solver.W, solver.w, solver.dx0))
    Wfull = np.triu(Wq) + np.triu(Wq, 1).transpose(0, 2, 1)
    rX, rU = kkt_dense(13, 24, N, J, b, Wfull, wq, dx0, 1e-6)
    assert np.abs(solver.dX[0].cpu().numpy() - rX).max() <= 1e-9 * max(1.0, np.abs(rX).max())
    assert np.abs(solver.dU[0].cpu().numpy() - rU).max() <= 1e-9 * max(1.0, np.abs(rU).max())
    # stage data against numpy on the node kernels' outputs: W = hess cost + J_h^T diag(b'') J_h, w = grad - J_h^T b'
    h, hJ, cg = solver.h[0].cpu().numpy(), solver.hJ[0].cpu().numpy(), solver.cgrad[0].cpu().numpy()
    rows, cols = solver.cost.hessian_sparsity()
    Hc = np.zeros((N, 37, 37))
    Hc[:, rows, cols] = solver.chess[0].cpu().numpy()
    ref_W = Hc + np.triu(np.einsum("kja,kj,kjb->kab", hJ, poly_barrier(-h, 1.0, 1.0, 2), hJ))
    assert np.abs(np.triu(Wq) - ref_W).max() <= 1e-10 * max(1.0, np.abs(ref_W).max())
    assert np.abs(wq - (cg - np.einsum("kja,kj->ka", hJ, poly_barrier(-h, 1.0, 1.0, 1)))).max() <= 1e-10 * max(1.0, np.abs(wq).max())
    first = None
    for it in range(6):
        solver.iterate(Xd, Ud, xmd, pd, pc, pi, w=Wd)
        if it == 0:
            first = solver.theta0.clone()
    torch.cuda.synchronize()
    assert torch.isfinite(Xd).all() and torch.isfinite(Ud).all()
    assert (solver.theta0 < first).float().mean().item() > 0.9 and solver.theta0.median().item() < 0.8 * first.median().item()


def _anymal_problem(batch, N, seed):
    """Full-body quadruped OCP instances (BASELINE config 4's model: nx = 37, nu = 12, N = 20): trajectories around random postures,
    tracking cost towards the first state of each instance (`anymal_cost`), no inequality rows."""
    rng = np.random.default_rng(seed)
    x, u, _, p = O.synthetic_inputs("anymal", batch * (N + 1), seed=seed)
    X = x.reshape(batch, N + 1, 37).copy()
    X[:, 1:] = X[:, :1] + 0.05 * (X[:, 1:] - X[:, :1])  # a trajectory is a small perturbation of its first state
    X[:, :, 3:7] /= np.linalg.norm(X[:, :, 3:7], axis=2, keepdims=True)
    U = u.reshape(batch, N + 1, 12)[:, :N].copy()
    xm = X[:, 0] + rng.normal(size=(batch, 37)) * 0.01
    ref = X[:, 0].copy()
    ref[:, 19:] = 0.0  # come to rest at the initial posture
    p_cost = np.concatenate((ref, np.tile([10.0, 10.0, 1.0, 0.1, 1e-3], (batch, 1))), axis=1)
    p_dyn = np.tile(p[0], (batch, 1))
    return X, U, xm, p_dyn, p_cost


@gpu
def test_one_sqp_iteration_full_body_quadruped():
    """BASELINE config 4's model and batch end to end: ANYmal node Jacobians (lane-per-leg kernel), `anymal_cost` value / gradient /
    Hessian, stage QP data, Riccati solve with (nx, nu) = (37, 12) and the stacked line search for 4096 instances -- the first AND the
    second batched SQP iteration equal the numpy + torch-oracle restatement of SoftSQPOptimizer::Optimize on eight sampled instances (three of them carried through the second iteration)."""
    import torch
    from ungar_amd import sqp
    batch, N = 4096, 20  # BASELINE config 4's batch
    X, U, xm, p_dyn, p_cost = _anymal_problem(batch, N, 5)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    Xd, Ud, xmd, pd, pc = dev(X), dev(U), dev(xm), dev(p_dyn), dev(p_cost)
    solver = sqp.BatchedSoftSqp("anymal", "anymal_cost", N, batch)
    sample = [0, 37, 511, 1024, 2047, 3000, 4000, 4095]
    Xc, Uc = {i: X[i].copy() for i in sample}, {i: U[i].copy() for i in sample}  # the reference iterates of the sampled instances
    th0 = None
    for iteration in (1, 2):  # the second iteration starts from the first one's (accepted) iterate on both sides
        accepted = solver.iterate(Xd, Ud, xmd, pd, pc)
        torch.cuda.synchronize()
        assert (solver.status == 0).all()
        acc = accepted.cpu().numpy()
        dXd, dUd, Xn, Un = solver.dX.cpu().numpy(), solver.dU.cpu().numpy(), Xd.cpu().numpy(), Ud.cpu().numpy()
        theta0, phi0 = solver.theta0.cpu().numpy(), solver.phi0.cpu().numpy()
        if th0 is None:
            th0 = theta0.copy()
        assert (acc > 0).mean() > 0.9
        for i in (sample if iteration == 1 else sample[:3]):
            # node Jacobians by torch autograd (~7 s per instance and iteration) for two instances of the first iteration and one of the second,
            # by the oracle's generated C for the others
            independent = i in (sample[:2] if iteration == 1 else sample[:1])
            dX, dU, alpha, Xr, Ur, (theta, phi, slope) = _reference_iteration(Xc[i], Uc[i], xm[i], p_dyn[i], p_cost[i], None, "anymal", "anymal_cost", N,
                                                                              c_checker_jacobian=not independent)
            scale = max(1.0, np.abs(dX).max(), np.abs(dU).max())
            assert np.abs(dXd[i] - dX).max() <= 1e-8 * scale and np.abs(dUd[i] - dU).max() <= 1e-8 * scale
            assert abs(theta0[i] - theta) <= 1e-9 * max(1.0, theta) and abs(phi0[i] - phi) <= 1e-9 * max(1.0, abs(phi))
            assert acc[i] == alpha
            assert np.abs(Xn[i] - Xr).max() <= 1e-8 * max(1.0, np.abs(Xr).max()) and np.abs(Un[i] - Ur).max() <= 1e-8 * max(1.0, np.abs(Ur).max())
            Xc[i], Uc[i] = Xr, Ur
    # a few more iterations: the dynamics defect keeps shrinking
    theta_first = th0.copy()
    for _ in range(2):
        solver.iterate(Xd, Ud, xmd, pd, pc)
    torch.cuda.synchronize()
    assert (solver.status == 0).all()
    assert np.median(solver.theta0.cpu().numpy() / theta_first) < 0.5
