"""Rigid-body quantities as batched node models (reference
include/ungar/rbd/quantities/{joint_torques, joint_space_inertia_matrix, joint_space_inertia_matrix_inverse, frames,
centroidal_momentum}.hpp:42-43): anymal_rnea / anymal_crba / anymal_minv / anymal_feet / anymal_centroidal.

  CPU   the product's tapes lowered to C (oracle/_gen/<model>_cg.c) against the committed oracle fixtures
        tests/golden/rbd_<model>.npz (values and Jacobians from RNEA, RNEA-column mass matrix, forward kinematics and a
        spatial-momentum sum with torch.autograd -- made by tests/golden/make_rbd_golden.py), plus identities between the
        quantities (M symmetric positive definite, M M^-1 = 1, tau = M a + tau(a = 0), orthonormal foot rotations) and the
        mass matrix of the formulation-independent Lagrangian oracle;
  GPU   the HIP kernels through the C ABI against the same fixtures, both device layouts, dense and sparse Jacobians.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import ungar_oracle as O

MODELS = ("anymal_rnea", "anymal_crba", "anymal_minv", "anymal_feet", "anymal_centroidal")


def _golden(repo_root, name):
    return np.load(os.path.join(repo_root, "tests", "golden", f"rbd_{name}.npz"))


@pytest.fixture(scope="module")
def clib(repo_root):
    from oracle import build_oracle
    if not all(os.path.exists(os.path.join(repo_root, "oracle", "_gen", f"{m}_cg.c")) for m in MODELS):
        pytest.skip("oracle/_gen/*.c not generated: run __graft_entry__.build()")
    return ctypes.CDLL(build_oracle.build("portable", models=build_oracle.RBD_MODELS))


def _eval_c(clib, name, x, u):
    nx, nu, ny = O.RBD_DIMS[name]
    nnz = ctypes.c_int.in_dll(clib, f"{name}_jac_nnz").value
    assert ctypes.c_int.in_dll(clib, f"{name}_ny").value == ny and list((ctypes.c_int * 4).in_dll(clib, f"{name}_dims")) == [nx, nu, 0, 0]
    rows = np.ctypeslib.as_array((ctypes.c_int * max(nnz, 1)).in_dll(clib, f"{name}_jac_row"))[:nnz]
    cols = np.ctypeslib.as_array((ctypes.c_int * max(nnz, 1)).in_dll(clib, f"{name}_jac_col"))[:nnz]
    dp = ctypes.POINTER(ctypes.c_double)
    ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    y, y0, jac, dummy = np.zeros(ny), np.zeros(ny), np.zeros(max(nnz, 1)), np.zeros(1)
    xx, uu = np.ascontiguousarray(x), (np.ascontiguousarray(u) if nu else dummy)
    getattr(clib, f"{name}_sparse_jacobian")(ptr(xx), ptr(uu), ptr(dummy), ptr(dummy), ptr(y), ptr(jac))
    getattr(clib, f"{name}_forward_zero")(ptr(xx), ptr(uu), ptr(dummy), ptr(dummy), ptr(y0))
    assert np.abs(y - y0).max() <= 1e-13 * max(1.0, np.abs(y).max())  # two separately compiled bodies (-ffast-math)
    J = np.zeros((ny, nx + nu))
    J[rows, cols] = jac[:nnz]
    return y, J, nnz


@pytest.mark.parametrize("name", MODELS)
def test_generated_c_matches_the_oracle_fixture(repo_root, clib, name):
    g = _golden(repo_root, name)
    for i in range(g["x"].shape[0]):
        y, J, nnz = _eval_c(clib, name, g["x"][i], g["u"][i])
        assert np.abs(y - g["y"][i]).max() <= 1e-11 * max(1.0, np.abs(g["y"][i]).max())
        if name == "anymal_minv":
            assert nnz == 0  # value-only model
            continue
        assert np.abs(J - g["J"][i]).max() <= 1e-10 * max(1.0, np.abs(g["J"][i]).max())
        assert not ((J == 0) & (np.abs(g["J"][i]) > 1e-11)).any()  # the structural pattern misses nothing


def test_identities_between_the_quantities(repo_root, clib):
    import torch
    from oracle import lagrange_oracle as L
    g = _golden(repo_root, "anymal_rnea")
    lag = L.LagrangeModel(L.load_fixture())
    for i in range(3):
        x, a = g["x"][i], g["u"][i]
        q = x[:19]
        M = _eval_c(clib, "anymal_crba", q, None)[0].reshape(18, 18)
        Minv = _eval_c(clib, "anymal_minv", q, None)[0].reshape(18, 18)
        assert np.abs(M - M.T).max() < 1e-13 and np.linalg.eigvalsh(M).min() > 0
        assert np.abs(M @ Minv - np.eye(18)).max() < 1e-10
        tau = _eval_c(clib, "anymal_rnea", x, a)[0]
        tau0 = _eval_c(clib, "anymal_rnea", x, np.zeros(18))[0]
        assert np.abs(tau - tau0 - M @ a).max() < 1e-10  # tau = M a + h(q, v)
        # the mass matrix of the energy-based oracle (no spatial algebra, un-lumped links read by the second URDF reader)
        M_lag = lag.mass_matrix(torch.zeros(18, dtype=torch.float64), torch.zeros(3, dtype=torch.float64), torch.eye(3, dtype=torch.float64), torch.as_tensor(q[7:])).numpy()
        assert np.abs(M - M_lag).max() < 1e-11
        feet = _eval_c(clib, "anymal_feet", q, None)[0].reshape(4, 12)
        for f in range(4):
            R = feet[f, 3:].reshape(3, 3)
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
        local = lag.frame_positions(np.concatenate((np.zeros(3), [0, 0, 0, 1], q[7:])), [f"{leg}_FOOT" for leg in L.LEGS])
        world = torch.as_tensor(q[:3]) + (L.quat_to_rot(torch.as_tensor(q[3:7])) @ local.T).T
        assert np.abs(feet[:, :3] - world.numpy()).max() < 1e-12
        h = _eval_c(clib, "anymal_centroidal", x, None)[0]
        Mlin = M[:3, :] @ x[19:]  # base-frame linear momentum = first three rows of M v; rotate to the world
        assert np.abs(L.quat_to_rot(torch.as_tensor(q[3:7])).numpy() @ Mlin - h[:3]).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("layout", ["soa", "aos"])
def test_hip_kernels_match_the_oracle_fixture(repo_root, name, layout):
    import ungar_amd
    g = _golden(repo_root, name)
    m = ungar_amd.NodeModel(name)
    nx, nu, ny = O.RBD_DIMS[name]
    assert (m.nx, m.nu, m.nw, m.np, m.ny) == (nx, nu, 0, 0, ny)
    count = g["x"].shape[0]
    none = np.zeros((count, 0))
    y, _ = m.evaluate_numpy(g["x"], g["u"], none, none, mode="value", layout=layout)
    assert np.abs(y - g["y"]).max() <= 1e-11 * max(1.0, np.abs(g["y"]).max())
    if name == "anymal_minv":
        assert not m.implements_jacobian()
        with pytest.raises(ungar_amd.UngarError, match="without a Jacobian"):
            m.evaluate_numpy(g["x"], g["u"], none, none, mode="dense", layout=layout)
        return
    for mode in ("dense", "sparse"):
        y, J = m.evaluate_numpy(g["x"], g["u"], none, none, mode=mode, layout=layout)
        assert np.isfinite(J).all()
        assert np.abs(y - g["y"]).max() <= 1e-11 * max(1.0, np.abs(g["y"]).max())
        assert np.abs(J - g["J"]).max() <= 1e-10 * max(1.0, np.abs(g["J"]).max())


@pytest.mark.gpu
def test_batched_quantities_at_scale():
    """65 536 configurations in one launch per quantity: identities that need no oracle (M M^-1 = 1, tau linear in a,
    orthonormal foot rotations), i.e. every lane of a large launch computes what the small launches compute."""
    import torch
    import ungar_amd
    count = 65536
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    r = lambda n: torch.rand((n, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1  # noqa: E731
    quat = torch.randn((4, count), generator=gen, device="cuda", dtype=torch.float64)
    q = torch.cat((r(3), quat / quat.norm(dim=0, keepdim=True), r(12)))
    Op = ungar_amd.Operand

    def run(name, x, u=None):
        m = ungar_amd.NodeModel(name)
        y = torch.full((m.ny, count), float("nan"), dtype=torch.float64, device="cuda")
        m.forward_zero(count, Op.soa(x, count), None if u is None else Op.soa(u, count), None, None, Op.soa(y, count))
        torch.cuda.synchronize()
        return y

    M = run("anymal_crba", q).t().reshape(count, 18, 18)
    Minv = run("anymal_minv", q).t().reshape(count, 18, 18)
    assert (torch.bmm(M, Minv) - torch.eye(18, dtype=torch.float64, device="cuda")).abs().max().item() < 1e-9
    v, a = r(18), r(18)
    x = torch.cat((q, v))
    tau, tau0 = run("anymal_rnea", x, a), run("anymal_rnea", x, torch.zeros_like(a))
    assert ((tau - tau0).t() - torch.bmm(M, a.t().unsqueeze(2)).squeeze(2)).abs().max().item() < 1e-9
    feet = run("anymal_feet", q).t().reshape(count, 4, 12)
    R = feet[:, :, 3:].reshape(count * 4, 3, 3)
    assert (torch.bmm(R, R.transpose(1, 2)) - torch.eye(3, dtype=torch.float64, device="cuda")).abs().max().item() < 1e-12


@pytest.mark.gpu
def test_joint_torque_jacobians_at_scale_lane_per_leg():
    """The Jacobian modes of 'anymal_rnea' run the lane-per-leg program (quad_rnea_kernel.hpp): 65 531 configurations (a ragged last wavefront)
    in one launch, dense block and CSR values; identities that need no oracle -- d tau / d a = M(q) from the CRBA kernel, the values equal the
    value-only (lane-per-node) kernel's, the CSR values are the dense block gathered through the pattern with exact zeros elsewhere,
    and J (dq, dv) matches central differences of the value kernel along a random direction at EVERY configuration."""
    import torch
    import ungar_amd
    count = 65531
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    r = lambda n: torch.rand((n, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1  # noqa: E731
    quat = torch.randn((4, count), generator=gen, device="cuda", dtype=torch.float64)
    q = torch.cat((r(3), quat / quat.norm(dim=0, keepdim=True), r(12)))
    x, a = torch.cat((q, r(18))), r(18)
    Op = ungar_amd.Operand
    m = ungar_amd.NodeModel("anymal_rnea")
    rows, cols = (torch.as_tensor(t.astype(np.int64), device="cuda") for t in m.jacobian_sparsity())
    y0 = torch.full((18, count), float("nan"), dtype=torch.float64, device="cuda")
    m.forward_zero(count, Op.soa(x, count), Op.soa(a, count), None, None, Op.soa(y0, count))
    yd, Jd = torch.full_like(y0, float("nan")), torch.full((18 * 55, count), float("nan"), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, Op.soa(x, count), Op.soa(a, count), None, None, Op.soa(yd, count), Op.soa(Jd, count))
    ys, Js = torch.full_like(y0, float("nan")), torch.full((m.jac_nnz, count), float("nan"), dtype=torch.float64, device="cuda")
    m.sparse_jacobian(count, Op.soa(x, count), Op.soa(a, count), None, None, Op.soa(ys, count), Op.soa(Js, count))
    torch.cuda.synchronize()
    assert torch.isfinite(Jd).all() and torch.isfinite(Js).all()
    scale = y0.abs().max().item()
    assert (yd - y0).abs().max().item() <= 1e-12 * scale and (ys - y0).abs().max().item() <= 1e-12 * scale
    J = Jd.t().reshape(count, 18, 55)
    # (two instantiations of the same body: the compiler contracts multiply-adds differently around the sinks that differ, so not bit for bit)
    assert (Js.t() - J[:, rows, cols]).abs().max().item() <= 1e-12 * J.abs().max().item()
    mask = torch.ones((18, 55), dtype=torch.bool, device="cuda")
    mask[rows, cols] = False
    assert (J[:, mask] == 0.0).all()
    crba = ungar_amd.NodeModel("anymal_crba")
    M = torch.empty((324, count), dtype=torch.float64, device="cuda")
    crba.forward_zero(count, Op.soa(q, count), None, None, None, Op.soa(M, count))
    torch.cuda.synchronize()
    assert (J[:, :, 37:] - M.t().reshape(count, 18, 18)).abs().max().item() <= 1e-10 * M.abs().max().item()
    # directional derivative along (dq, dv): the quaternion moves along a direction of R^4 (the node differentiates with respect to its four entries)
    d = r(37)
    h = 1e-6
    yp, ym = torch.empty_like(y0), torch.empty_like(y0)
    m.forward_zero(count, Op.soa(x + h * d, count), Op.soa(a, count), None, None, Op.soa(yp, count))
    m.forward_zero(count, Op.soa(x - h * d, count), Op.soa(a, count), None, None, Op.soa(ym, count))
    torch.cuda.synchronize()
    fd = ((yp - ym) / (2 * h)).t()
    jd = torch.bmm(J[:, :, :37], d.t().unsqueeze(2)).squeeze(2)
    assert (fd - jd).abs().max().item() <= 2e-6 * max(1.0, jd.abs().max().item())


@pytest.mark.gpu
def test_inertia_matrix_jacobian_at_scale_lane_per_leg():
    """The CSR Jacobian mode of 'anymal_crba' runs the lane-per-leg program (quad_crba_kernel.hpp): 65 531 configurations in one launch; M equals the
    value-only (lane-per-node) kernel's, and d M / d q applied to a random direction matches central differences of the value kernel at EVERY configuration."""
    import torch
    import ungar_amd
    count = 65531
    gen = torch.Generator(device="cuda")
    gen.manual_seed(12)
    r = lambda n: torch.rand((n, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1  # noqa: E731
    quat = torch.randn((4, count), generator=gen, device="cuda", dtype=torch.float64)
    q = torch.cat((r(3), quat / quat.norm(dim=0, keepdim=True), r(12)))
    Op = ungar_amd.Operand
    m = ungar_amd.NodeModel("anymal_crba")
    rows, cols = (torch.as_tensor(t.astype(np.int64), device="cuda") for t in m.jacobian_sparsity())
    M0 = torch.full((324, count), float("nan"), dtype=torch.float64, device="cuda")
    m.forward_zero(count, Op.soa(q, count), None, None, None, Op.soa(M0, count))
    M1, Js = torch.full_like(M0, float("nan")), torch.full((m.jac_nnz, count), float("nan"), dtype=torch.float64, device="cuda")
    m.sparse_jacobian(count, Op.soa(q, count), None, None, None, Op.soa(M1, count), Op.soa(Js, count))
    torch.cuda.synchronize()
    assert torch.isfinite(Js).all()
    assert (M1 - M0).abs().max().item() <= 1e-12 * M0.abs().max().item()
    d = r(19)
    h = 1e-6
    Mp, Mm = torch.empty_like(M0), torch.empty_like(M0)
    m.forward_zero(count, Op.soa(q + h * d, count), None, None, None, Op.soa(Mp, count))
    m.forward_zero(count, Op.soa(q - h * d, count), None, None, None, Op.soa(Mm, count))
    torch.cuda.synchronize()
    fd = (Mp - Mm) / (2 * h)                       # (324, count)
    jd = torch.zeros_like(fd)
    jd.index_add_(0, rows, Js * d[cols])           # sum over the pattern entries of every row
    assert (fd - jd).abs().max().item() <= 2e-6 * max(1.0, jd.abs().max().item())


@pytest.mark.gpu
def test_centroidal_momentum_jacobian_at_scale_lane_per_leg():
    """The Jacobian modes of 'anymal_centroidal' run the lane-per-leg program: 65 531 configurations, dense block and CSR values; the values equal
    the value-only (lane-per-node) kernel's, and J (dq, dv) matches central differences of the value kernel along a random direction -- the raw
    quaternion entries included -- at EVERY configuration."""
    import torch
    import ungar_amd
    count = 65531
    gen = torch.Generator(device="cuda")
    gen.manual_seed(13)
    r = lambda n: torch.rand((n, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1  # noqa: E731
    quat = torch.randn((4, count), generator=gen, device="cuda", dtype=torch.float64)
    x = torch.cat((r(3), quat / quat.norm(dim=0, keepdim=True), r(12), r(18)))
    Op = ungar_amd.Operand
    m = ungar_amd.NodeModel("anymal_centroidal")
    rows, cols = (torch.as_tensor(t.astype(np.int64), device="cuda") for t in m.jacobian_sparsity())
    y0 = torch.full((6, count), float("nan"), dtype=torch.float64, device="cuda")
    m.forward_zero(count, Op.soa(x, count), None, None, None, Op.soa(y0, count))
    yd, Jd = torch.full_like(y0, float("nan")), torch.full((6 * 37, count), float("nan"), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, Op.soa(x, count), None, None, None, Op.soa(yd, count), Op.soa(Jd, count))
    ys, Js = torch.full_like(y0, float("nan")), torch.full((m.jac_nnz, count), float("nan"), dtype=torch.float64, device="cuda")
    m.sparse_jacobian(count, Op.soa(x, count), None, None, None, Op.soa(ys, count), Op.soa(Js, count))
    torch.cuda.synchronize()
    assert torch.isfinite(Jd).all() and torch.isfinite(Js).all()
    scale = y0.abs().max().item()
    assert (yd - y0).abs().max().item() <= 1e-12 * scale and (ys - y0).abs().max().item() <= 1e-12 * scale
    J = Jd.t().reshape(count, 6, 37)
    assert (Js.t() - J[:, rows, cols]).abs().max().item() <= 1e-12 * J.abs().max().item()
    d = r(37)
    h = 1e-6
    yp, ym = torch.empty_like(y0), torch.empty_like(y0)
    m.forward_zero(count, Op.soa(x + h * d, count), None, None, None, Op.soa(yp, count))
    m.forward_zero(count, Op.soa(x - h * d, count), None, None, None, Op.soa(ym, count))
    torch.cuda.synchronize()
    fd = ((yp - ym) / (2 * h)).t()
    jd = torch.bmm(J, d.t().unsqueeze(2)).squeeze(2)
    assert (fd - jd).abs().max().item() <= 2e-6 * max(1.0, jd.abs().max().item())
