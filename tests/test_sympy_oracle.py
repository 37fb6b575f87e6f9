"""Closed-form (symbolic) Jacobians of the quadrotor / rc_car / single-rigid-body quadruped node functions
(oracle/sympy_oracle.py -> tests/golden/sympy_<model>.npz; VERDICT r01 "tighten the derivative pin") against
  CPU   the torch-autograd oracle, the product's derivative programs lowered to C, and the product's structural sparsity
        pattern (the symbolic Jacobian's non-zeros: 118 / 32 / 238 entries) -- entry for entry;
  GPU   the HIP kernels through the C ABI, both layouts, dense and sparse.
Three derivative engines of different kinds (symbolic differentiation, reverse-mode autograd on a dynamic graph, the
product's source-to-source tape transforms) must agree on the same restated functions to 1e-10 of the block scale."""
import ctypes
import os

import numpy as np
import pytest

from oracle import ungar_oracle as O

MODELS = ("quadrotor", "rc_car", "srbd")


def _fixture(repo_root, name):
    return np.load(os.path.join(repo_root, "tests", "golden", f"sympy_{name}.npz"))


@pytest.mark.parametrize("name", MODELS)
def test_symbolic_jacobians_match_the_autograd_oracle(repo_root, name):
    g = _fixture(repo_root, name)
    rf, rJ = O.node_jacobian(name, g["x"], g["u"], g["w"], g["p"])
    scale = np.abs(g["J"]).max(axis=(1, 2), keepdims=True)
    assert np.abs(rf - g["f"]).max() <= 1e-12 * max(1.0, np.abs(g["f"]).max())
    assert (np.abs(rJ - g["J"]) <= 1e-11 * scale).all()
    assert ((g["pattern"] == 0) <= (np.abs(rJ).max(axis=0) == 0)).all()  # structural zeros of the symbolic Jacobian are zeros


def test_fixture_regenerates(repo_root):
    from oracle import sympy_oracle as S
    g = _fixture(repo_root, "rc_car")
    value, jacobian, pattern = S.build("rc_car")
    assert (pattern == g["pattern"]).all()
    for i in (0, 5):
        assert np.abs(value(g["x"][i], g["u"][i], g["w"][i], g["p"][i]) - g["f"][i]).max() < 1e-13
        assert np.abs(jacobian(g["x"][i], g["u"][i], g["w"][i], g["p"][i]) - g["J"][i]).max() < 1e-12


@pytest.mark.parametrize("name", MODELS)
def test_product_programs_and_pattern_match_the_symbolic_jacobian(repo_root, name):
    from oracle import build_oracle
    path = build_oracle.lib_path("portable")
    if not os.path.exists(path):
        pytest.skip("oracle/_gen library not built: run __graft_entry__.build()")
    clib = ctypes.CDLL(path)
    g = _fixture(repo_root, name)
    nx, nu, nw, _ = list((ctypes.c_int * 4).in_dll(clib, f"{name}_dims"))
    nnz = ctypes.c_int.in_dll(clib, f"{name}_jac_nnz").value
    rows = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_row"))
    cols = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_col"))
    product_pattern = np.zeros((nx, nx + nu), dtype=np.int8)
    product_pattern[rows, cols] = 1
    assert (product_pattern == g["pattern"]).all(), "the tape's dependency sparsity must equal the symbolic structural pattern"
    dp = ctypes.POINTER(ctypes.c_double)
    ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    for b in range(g["x"].shape[0]):
        x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
        w = np.ascontiguousarray(g["w"][b]) if nw else np.zeros(1)
        f, jac = np.zeros(nx), np.zeros(nnz)
        getattr(clib, f"{name}_sparse_jacobian")(ptr(x), ptr(u), ptr(w), ptr(p), ptr(f), ptr(jac))
        J = np.zeros((nx, nx + nu))
        J[rows, cols] = jac
        assert np.abs(f - g["f"][b]).max() <= 1e-11 * max(1.0, np.abs(g["f"][b]).max())
        assert np.abs(J - g["J"][b]).max() <= 1e-10 * np.abs(g["J"][b]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_hip_kernels_match_the_symbolic_jacobian(repo_root, name, layout, mode):
    import ungar_amd
    g = _fixture(repo_root, name)
    m = ungar_amd.NodeModel(name)
    f, J = m.evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode=mode, layout=layout)
    scale = np.abs(g["J"]).max(axis=(1, 2), keepdims=True)
    assert np.abs(f - g["f"]).max() <= 1e-11 * max(1.0, np.abs(g["f"]).max())
    assert (np.abs(J - g["J"]) <= 1e-10 * scale).all()
    rows, cols = m.jacobian_sparsity()
    pattern = np.zeros_like(g["pattern"])
    pattern[rows, cols] = 1
    assert (pattern == g["pattern"]).all()
