"""Host-side optimisation layer of the C++ facade: build/optimization_test
checks derivatives, continuity, KKT residuals and the line-search branches itself; here the barrier values
are compared with an independent restatement of the reference's formulas
(include/ungar/optimization/soft_inequality_constraint.hpp:98-105, 182-190, 207-222) and the QP solutions
with a dense numpy solve of the same KKT systems."""
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def output(repo_root):
    exe = os.path.join(repo_root, "build", "optimization_test")
    if not os.path.exists(exe):
        pytest.skip("build/optimization_test missing: run __graft_entry__.build()")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout[-2000:]
    return r.stdout.splitlines()


def _poly(x, rhs, k, eps):
    a1, b1 = k, -0.5 * k * eps
    c1 = -1.0 / 3.0 * (-b1 - a1 * eps) * eps - 0.5 * a1 * eps ** 2 - b1 * eps
    a2, b2, c2, d2 = (-b1 - a1 * eps) / eps ** 2, a1, b1, c1
    x = x - rhs
    if x < 0.0:
        return 0.5 * a1 * x ** 2 + b1 * x + c1
    if x < eps:
        return 1.0 / 3.0 * a2 * x ** 3 + 0.5 * b2 * x ** 2 + c2 * x + d2
    return 0.0


def _log(x, rhs, mu, eps):
    x = x - rhs
    if x >= eps:
        return -mu * np.log(x)
    return mu / 2.0 * (((x - 2.0 * eps) / eps) ** 2 - 1.0) - mu * np.log(eps)


def test_barrier_values_match_the_reference_formulas(output):
    rows = {"POLY": [], "LOG": [], "BOUND": []}
    for line in output:
        t = line.split()
        if t and t[0] in rows:
            rows[t[0]].append((float(t[1]), float(t[2])))
    assert all(len(v) == 41 for v in rows.values())
    for x, v in rows["POLY"]:
        assert abs(v - _poly(x, 0.5, 100.0, 2e-2)) <= 1e-14 * (1 + abs(v))
    for x, v in rows["LOG"]:
        assert abs(v - _log(x, -1.0, 1e-2, 0.5)) <= 1e-14 * (1 + abs(v))
    eps = (2.0 - -1.0) * 1e-1  # SoftBoundConstraint: relative epsilon
    for x, v in rows["BOUND"]:
        assert abs(v - (_poly(x, -1.0, 10.0, eps) + _poly(-x, -2.0, 10.0, eps))) <= 1e-14 * (1 + abs(v))


def test_kkt_solver_matches_dense_solve(output):
    problems, cur = [], None
    for line in output:
        t = line.split()
        if not t:
            continue
        if t[0] == "QP":
            n, m = int(t[1]), int(t[2])
            cur = {"H": np.zeros((n, n)), "A": np.zeros((m, n)), "g": np.zeros(n), "b": np.zeros(m), "d": np.zeros(n)}
        elif t[0] == "ENDQP":
            problems.append(cur)
        elif cur is not None and t[0] in ("H", "A"):
            cur[t[0]][int(t[1]), int(t[2])] = float(t[3])
        elif cur is not None and t[0] in ("g", "b", "d"):
            cur[t[0]][int(t[1])] = float(t[2])
    assert len(problems) == 3
    for P in problems:
        H = P["H"] + np.triu(P["H"], 1).T
        n, m = H.shape[0], P["A"].shape[0]
        K = np.block([[H, P["A"].T], [P["A"], np.zeros((m, m))]])
        sol = np.linalg.solve(K, np.concatenate((-P["g"], P["b"])))
        assert np.abs(sol[:n] - P["d"]).max() <= 1e-9 * max(1.0, np.abs(sol[:n]).max())
