"""The tape engine's C dialect (oracle/_gen/*_cg.c = CPU baseline / checker, stand-in for the C that
CppADCodeGen generates for the reference) against the independent torch oracle's golden vectors.
This pins BOTH the checker used by bench.py's cpu_baseline leg and -- because the HIP bodies are
lowered from the very same tapes -- the recorder, the derivative transforms and the sparsity logic,
without needing a GPU."""
import ctypes
import os

import numpy as np
import pytest

MODELS = ("quadrotor", "rc_car", "srbd", "anymal", "anymal_ad")


@pytest.fixture(scope="module")
def clib(repo_root):
    from oracle import build_oracle
    path = build_oracle.lib_path("portable")
    if not os.path.exists(path):
        if not all(os.path.exists(os.path.join(repo_root, "oracle", "_gen", f"{m}_cg.c")) for m in MODELS):
            pytest.skip("oracle/_gen/*.c not generated: run __graft_entry__.build()")
        path = build_oracle.build("portable")
    return ctypes.CDLL(path)


@pytest.mark.parametrize("name,fixture", [(m, None) for m in MODELS] + [("anymal", "node_anymal_256.npz"), ("anymal_ad", "node_anymal_256.npz")] +
                         [("quadrotor", "node_quadrotor_edge.npz"), ("rc_car", "node_rc_car_edge.npz"), ("anymal", "node_anymal_edge.npz"), ("anymal_ad", "node_anymal_edge.npz")])
def test_generated_c_matches_golden(repo_root, clib, name, fixture):
    """(node_*_edge.npz: nodes ON the edge cases of the reference's helpers -- omega+ = 0 exactly for ApproximateExponentialMap, stored quaternions of length 1.3 / 0.7,
    v_x at the bounds with zero slip angles, the robot at rest: tests/golden/make_edge_cases.py.)
    (node_anymal_256.npz: 256 nodes of the headline model from the independent torch oracle, tests/golden/make_anymal_many.py -- the C checker that the
    every-node GPU tests compare with is itself pinned at hundreds of nodes, for both the structured and the taped-ABA program)"""
    g = np.load(f"{repo_root}/tests/golden/{fixture or 'node_' + name.replace('_ad', '') + '.npz'}")
    dims = (ctypes.c_int * 4).in_dll(clib, f"{name}_dims")
    nx, nu, nw, _ = list(dims)
    nnz = ctypes.c_int.in_dll(clib, f"{name}_jac_nnz").value
    rows = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_row"))
    cols = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_col"))
    key = rows.astype(np.int64) * (nx + nu) + cols
    assert (np.diff(key) > 0).all()
    dp = ctypes.POINTER(ctypes.c_double)
    jac_fn, val_fn = getattr(clib, f"{name}_sparse_jacobian"), getattr(clib, f"{name}_forward_zero")
    for b in range(g["x"].shape[0]):
        x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
        w = np.ascontiguousarray(g["w"][b]) if nw else np.zeros(1)
        f, f0, jac = np.zeros(nx), np.zeros(nx), np.zeros(nnz)
        ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
        jac_fn(ptr(x), ptr(u), ptr(w), ptr(p), ptr(f), ptr(jac))
        val_fn(ptr(x), ptr(u), ptr(w), ptr(p), ptr(f0))
        J = np.zeros((nx, nx + nu))
        J[rows, cols] = jac
        assert np.abs(f - g["f"][b]).max() <= 1e-11 * max(1.0, np.abs(g["f"][b]).max())
        assert np.abs(f0 - f).max() <= 1e-13 * max(1.0, np.abs(f).max())
        assert np.abs(J - g["J"][b]).max() <= 1e-10 * np.abs(g["J"][b]).max()
        # the structural pattern covers every numerically non-zero entry
        assert not ((g["J"][b] != 0) & (J == 0) & (np.abs(g["J"][b]) > 1e-12)).any()
