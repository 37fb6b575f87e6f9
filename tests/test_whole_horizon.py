"""GPU test: the reference's whole-horizon quadrotor OCP (objective / equality / inequality tapes of
example/mpc/quadrotor.example.cpp:196-316) built with ungar_amd's C++ facade and evaluated through
Ungar::Autodiff::Function on the MI355X, checked BLOCK BY BLOCK against the per-shooting-node kernel
(rows nx+k*nx.. are x_{k+1} - f(x_k,u_k): d/dx_{k+1} = I,
d/dx_k = -A_k, d/du_k = -B_k) and against an independent torch model of the objective."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import ungar_oracle as O

pytestmark = pytest.mark.gpu

N, NX, NU = 30, 13, 4
DEC, PAR = 523, 437


def _parse(path):
    out, lines, i = {}, open(path).read().split("\n"), 0
    while i < len(lines):
        t = lines[i].split()
        if not t:
            i += 1
            continue
        if len(t) == 2:  # vector
            n = int(t[1])
            out[t[0]] = np.array([float(v) for v in lines[i + 1:i + 1 + n]])
            i += 1 + n
        else:  # sparse: tag rows cols nnz
            r, c, nnz = int(t[1]), int(t[2]), int(t[3])
            M = np.zeros((r, c))
            mask = np.zeros((r, c), dtype=bool)
            for ln in lines[i + 1:i + 1 + nnz]:
                a, b, v = ln.split()
                M[int(a), int(b)] = float(v)
                mask[int(a), int(b)] = True
            out[t[0]], out[t[0] + "_MASK"] = M, mask
            i += 1 + nnz
    return out


@pytest.fixture(scope="module")
def dump(repo_root, tmp_path_factory):
    exe = os.path.join(repo_root, "build", "quadrotor_ocp_test")
    assert os.path.exists(exe), "build/quadrotor_ocp_test missing: run __graft_entry__.build()"
    d = tmp_path_factory.mktemp("ocp")
    r = subprocess.run([exe, str(d / "codegen"), str(d / "dump.txt")], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "DUMPED" in r.stdout
    return _parse(str(d / "dump.txt"))


def test_equality_constraints_block_structure(dump):
    import ungar_amd
    v = dump["INPUT"]
    assert v.size == DEC + PAR
    X = v[:(N + 1) * NX].reshape(N + 1, NX)
    U = v[(N + 1) * NX:DEC].reshape(N, NU)
    par = v[DEC:]
    p = np.tile(par[:20], (N, 1))            # node parameter block = parameters[0:20]
    xm = par[-13:]                           # measured_state is the last parameter
    f, J = ungar_amd.NodeModel("quadrotor").evaluate_numpy(X[:N], U, np.zeros((N, 0)), p, mode="dense", layout="aos")
    fo, Jo = O.node_jacobian("quadrotor", X[:N], U, np.zeros((N, 0)), p)
    assert np.abs(f - fo).max() < 1e-12 and np.abs(J - Jo).max() < 1e-10

    eq, Jeq, mask = dump["EQ"], dump["EQ_JAC"], dump["EQ_JAC_MASK"]
    assert eq.shape == (403,) and Jeq.shape == (403, DEC)  # m x n: parameter columns trimmed
    assert np.abs(eq[:NX] - (X[0] - xm)).max() < 1e-14
    expected = np.zeros((403, DEC))
    expected[:NX, :NX] = np.eye(NX)
    for k in range(N):
        rows = slice(NX + k * NX, NX + (k + 1) * NX)
        assert np.abs(eq[rows] - (X[k + 1] - f[k])).max() < 1e-12
        expected[rows, k * NX:(k + 1) * NX] = -J[k][:, :NX]
        expected[rows, (k + 1) * NX:(k + 2) * NX] += np.eye(NX)
        expected[rows, (N + 1) * NX + k * NU:(N + 1) * NX + (k + 1) * NU] = -J[k][:, NX:]
    assert np.abs(Jeq - expected).max() <= 1e-10 * np.abs(expected).max()
    assert not (Jeq != 0)[~mask].any()
    # structural pattern = identity blocks + the node pattern (118 nnz per knot),
    # except where -A_k and +I overlap on no entry (A_k couples x_k, I couples x_{k+1})
    assert mask.sum() == NX + N * (118 + NX)


def test_batched_assembly_from_node_blocks_matches_whole_horizon_function(dump):
    """Row N1: the block-bidiagonal equality Jacobian of a BATCH of instances assembled on the GPU from the
    node kernel's dense blocks (ungar_ocp_assemble_equality) -- instance 0 must reproduce, entry for entry
    and in the same CSR order, what the whole-horizon Ungar::Autodiff::Function returned."""
    import ungar_amd
    v = dump["INPUT"]
    m = ungar_amd.NodeModel("quadrotor")
    starts, cols = m.ocp_equality_sparsity(N)
    mask = dump["EQ_JAC_MASK"]
    assert starts[-1] == mask.sum() == len(cols)
    ref_rows, ref_cols = np.nonzero(mask)                      # row-major order = CSR order
    assert np.array_equal(cols, ref_cols) and np.array_equal(np.diff(starts), mask.sum(axis=1))

    batch, count = 5, 5 * N
    rng = np.random.default_rng(0)
    Xs = np.stack([v[:(N + 1) * NX].reshape(N + 1, NX)] + [v[:(N + 1) * NX].reshape(N + 1, NX) + 0.05 * rng.normal(size=(N + 1, NX)) for _ in range(batch - 1)])
    Us = np.stack([v[(N + 1) * NX:DEC].reshape(N, NU)] + [v[(N + 1) * NX:DEC].reshape(N, NU) * (1 + 0.05 * rng.normal(size=(N, NU))) for _ in range(batch - 1)])
    xm = np.tile(v[DEC:][-13:], (batch, 1))
    dev = torch.device("cuda")
    Xd = torch.as_tensor(Xs, device=dev).contiguous()           # (batch, N+1, nx): VariableMap-style state block
    Ud = torch.as_tensor(Us, device=dev).contiguous()
    pd = torch.as_tensor(v[DEC:DEC + 20], device=dev)
    xmd = torch.as_tensor(xm, device=dev)
    f = torch.empty((NX, count), dtype=torch.float64, device=dev)
    J = torch.empty((NX * (NX + NU), count), dtype=torch.float64, device=dev)
    Op = ungar_amd.Operand
    m.dense_jacobian(count, Op(Xd, (N + 1) * NX, NX, 1), Op(Ud, N * NU, NU, 1), None, Op.per_instance(pd, 20, shared=True), Op.soa(f, count, N),
                     Op.soa(J, count, N), knots=N)
    nnz = len(cols)
    g = torch.full((batch, (N + 1) * NX), float("nan"), dtype=torch.float64, device=dev)
    vals = torch.full((batch, nnz), float("nan"), dtype=torch.float64, device=dev)
    m.ocp_assemble_equality(N, batch, Op(Xd, (N + 1) * NX, NX, 1), Op(xmd, NX, 0, 1), Op.soa(f, count, N), Op.soa(J, count, N), Op(g, (N + 1) * NX, 0, 1),
                            Op(vals, nnz, 0, 1))
    torch.cuda.synchronize()
    g, vals, fh, Jh = g.cpu().numpy(), vals.cpu().numpy(), f.cpu().numpy().T, J.cpu().numpy().T.reshape(count, NX, NX + NU)
    assert np.isfinite(g).all() and np.isfinite(vals).all()
    # instance 0 == the whole-horizon Function
    assert np.abs(g[0] - dump["EQ"]).max() < 1e-12
    assert np.abs(vals[0] - dump["EQ_JAC"][ref_rows, ref_cols]).max() <= 1e-10 * np.abs(dump["EQ_JAC"]).max()
    # every instance == the block formula built on the host from the node outputs
    for b in range(batch):
        dense = np.zeros(((N + 1) * NX, DEC))
        dense[ref_rows, ref_cols] = vals[b]
        want = np.zeros_like(dense)
        want[:NX, :NX] = np.eye(NX)
        for k in range(N):
            rows = slice(NX + k * NX, NX + (k + 1) * NX)
            Jk = Jh[b * N + k]
            want[rows, k * NX:(k + 1) * NX] = -Jk[:, :NX]
            want[rows, (k + 1) * NX:(k + 2) * NX] += np.eye(NX)
            want[rows, (N + 1) * NX + k * NU:(N + 1) * NX + (k + 1) * NU] = -Jk[:, NX:]
            assert np.abs(g[b, rows] - (Xs[b, k + 1] - fh[b * N + k])).max() < 1e-13
        assert np.array_equal(dense, want)
        assert np.abs(g[b, :NX] - (Xs[b, 0] - xm[b])).max() < 1e-15


def test_inequality_constraints(dump):
    v = dump["INPUT"]
    U = v[(N + 1) * NX:DEC].reshape(N, NU)
    rmax = v[DEC + 20]
    h, Jh = dump["INEQ"], dump["INEQ_JAC"]
    assert h.shape == (240,) and Jh.shape == (240, DEC) and dump["INEQ_JAC_MASK"].sum() == 240
    want = np.stack((U - rmax, -U), axis=-1).reshape(-1)   # k-major, rotor-minor pairs [r - rmax, -r]
    assert np.abs(h - want).max() < 1e-13
    cols = (N + 1) * NX + np.repeat(np.arange(N * NU), 2)
    expected = np.zeros((240, DEC))
    expected[np.arange(240), cols] = np.tile([1.0, -1.0], N * NU)
    assert np.array_equal(Jh, expected)


def _objective_torch(z, par):
    X = z[:(N + 1) * NX].reshape(N + 1, NX)
    U = z[(N + 1) * NX:].reshape(N, NU)
    o = 21
    ref_p = par[o:o + 3 * (N + 1)].reshape(N + 1, 3)
    o += 3 * (N + 1)
    ref_q = par[o:o + 4 * (N + 1)].reshape(N + 1, 4)
    o += 4 * (N + 1)
    ref_v = par[o:o + 3 * (N + 1)].reshape(N + 1, 3)
    o += 3 * (N + 1)
    ref_w = par[o:o + 3 * (N + 1)].reshape(N + 1, 3)
    val = torch.zeros((), dtype=torch.float64)
    for k in range(N + 1):
        q = X[k, 3:7]
        val = val + ((X[k, 0:3] - ref_p[k]) ** 2).sum() + torch.minimum(((q - ref_q[k]) ** 2).sum(), ((q + ref_q[k]) ** 2).sum()) \
            + ((X[k, 7:10] - ref_v[k]) ** 2).sum() + ((X[k, 10:13] - ref_w[k]) ** 2).sum()
        if 0 < k < N:
            val = val + 1e-6 * ((U[k] - U[k - 1]) ** 2).sum()
        if k < N:
            val = val + 1e-6 * (U[k] ** 2).sum()
    return val


def test_objective_value_gradient_and_upper_triangular_hessian(dump):
    v = dump["INPUT"]
    z = torch.tensor(v[:DEC], requires_grad=True)
    par = torch.tensor(v[DEC:])
    val = _objective_torch(z, par)
    (grad,) = torch.autograd.grad(val, z, create_graph=False)
    H = torch.autograd.functional.hessian(lambda zz: _objective_torch(zz, par), torch.tensor(v[:DEC])).numpy()
    assert abs(dump["OBJ"][0] - val.item()) <= 1e-12 * abs(val.item())
    g = dump["OBJ_JAC"]
    assert g.shape == (1, DEC) and np.abs(g[0] - grad.numpy()).max() <= 1e-12 * np.abs(grad.numpy()).max()
    Hd, Hm = dump["OBJ_HES"], dump["OBJ_HES_MASK"]
    assert Hd.shape == (DEC, DEC)
    assert not Hm[np.tril_indices(DEC, -1)].any(), "Hessian must be upper-triangular only (function.hpp:232-235)"
    assert np.abs(Hd - np.triu(H)).max() <= 1e-12 * np.abs(H).max()


# ---- the quadruped OCP as written: 883 equality rows incl. the 480 foot-contact rows (quadruped.example.cpp:246-304) and its objective (:209-245) --------
@pytest.fixture(scope="module")
def quadruped_dump(repo_root, tmp_path_factory, shared_codegen):
    exe = os.path.join(repo_root, "build", "batched_quadruped_test")
    assert os.path.exists(exe), "build/batched_quadruped_test missing: run __graft_entry__.build()"
    d = tmp_path_factory.mktemp("quadruped_ocp")
    r = subprocess.run([exe, str(shared_codegen("batched_quadruped")), "4", "-1", str(d / "dump.txt")], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "DUMPED" in r.stdout
    return _parse(str(d / "dump.txt"))


def test_quadruped_whole_horizon_functions_against_the_independent_oracle(quadruped_dump):
    """The whole-horizon equality function (x_0 - x_m, 30 dynamics defects, 480 foot-contact rows that couple knots k and k - 1 through the world positions of the
    feet) and the objective of the reference's quadruped OCP, evaluated through the facade's Ungar::Autodiff::Function on the GPU at an instance with a random
    gait (stance, swing, touch-down and lift-off rows all occur), against oracle.ungar_oracle.quadruped_whole_horizon -- a torch restatement that shares no code
    with the product -- and its autograd Jacobian / gradient: values to 1e-12, every Jacobian entry to 1e-10 of the largest, structural zeros exact.
    (The batched SQP tests compare the device iteration with the facade's optimiser on THESE functions.)"""
    dump = quadruped_dump
    dec, par_n = 1123, 948
    v = dump["INPUT"]
    assert v.size == dec + par_n
    z, par = torch.tensor(v[:dec]), torch.tensor(v[dec:])
    value, eq = O.quadruped_whole_horizon(z, par)
    assert eq.shape == (883,) and dump["EQ"].shape == (883,)
    assert abs(dump["OBJ"][0] - value.item()) <= 1e-12 * abs(value.item())
    assert np.abs(dump["EQ"] - eq.numpy()).max() <= 1e-12 * max(1.0, np.abs(eq.numpy()).max())
    contact_rows = eq.numpy()[13 + 390:]
    assert (contact_rows != 0).sum() > 40 and (contact_rows == 0).sum() > 40  # active and inactive contact rows both occur
    zg = z.clone().requires_grad_(True)
    (grad,) = torch.autograd.grad(O.quadruped_whole_horizon(zg, par)[0], zg)
    g = dump["OBJ_JAC"]
    assert g.shape == (1, dec) and np.abs(g[0] - grad.numpy()).max() <= 1e-12 * np.abs(grad.numpy()).max()
    J = torch.autograd.functional.jacobian(lambda zz: O.quadruped_whole_horizon(zz, par)[1], z, vectorize=True).numpy()  # (one batched backward pass: 4 s; row by row 100-180 s, same entries)
    Jd, mask = dump["EQ_JAC"], dump["EQ_JAC_MASK"]
    assert Jd.shape == (883, dec)
    assert np.abs(Jd - J).max() <= 1e-10 * np.abs(J).max()
    assert not (J != 0)[~mask].any(), "an entry the oracle differentiates to non-zero is missing from the structural pattern"
    # the foot-contact rows couple knots: a stance row of knot k >= 1 has entries in x_k, u_k AND in x_(k-1), u_(k-1)
    coupled = 0
    for k in range(1, 30):
        for i in range(4):
            row = 13 + 390 + (k * 4 + i) * 4 + 1  # first of the three "sPrev s (pFoot - pFootPrev)" rows
            if np.any(J[row] != 0):
                assert np.any(J[row, (k - 1) * 13:k * 13] != 0) and np.any(J[row, k * 13:(k + 1) * 13] != 0)
                coupled += 1
    assert coupled > 20


def _check_function(dump, tag, value, jacobian, tol_value=1e-12, tol_jac=1e-10):
    """Facade values / sparse Jacobian (dump[tag], dump[tag + "_JAC"]) against an oracle value vector and dense Jacobian: entries to tol_jac of the largest, and
    every entry the oracle differentiates to non-zero present in the structural pattern."""
    assert dump[tag].shape == value.shape
    assert np.abs(dump[tag] - value).max() <= tol_value * max(1.0, np.abs(value).max())
    J, mask = dump[tag + "_JAC"], dump[tag + "_JAC_MASK"]
    assert J.shape == jacobian.shape
    assert np.abs(J - jacobian).max() <= tol_jac * max(np.abs(jacobian).max(), 1e-300)
    assert not (jacobian != 0)[~mask].any(), "an entry the oracle differentiates to non-zero is missing from the structural pattern"


def test_quadruped_inequality_rows_and_objective_hessian_against_the_independent_oracle(quadruped_dump):
    """The other half of the reference's quadruped OCP (quadruped.example.cpp:209-250, 306-338): the 360 inequality rows (unilateral force, friction cone and leg
    reach through Utils::ApproximateNorm, switched by the reference contact state) with their 360 x 1123 Jacobian, and the UPPER-TRIANGULAR 1123 x 1123 Hessian of
    the objective (quaternion `min` term included), through the facade's Ungar::Autodiff::Function on the GPU against torch autograd on
    oracle.ungar_oracle -- a restatement that shares no code with the product."""
    dump = quadruped_dump
    dec = 1123
    v = dump["INPUT"]
    z, par = torch.tensor(v[:dec]), torch.tensor(v[dec:])
    h = O.quadruped_whole_horizon_inequalities(z, par)
    assert h.shape == (360,)
    Jh = torch.autograd.functional.jacobian(lambda zz: O.quadruped_whole_horizon_inequalities(zz, par), z, vectorize=True).numpy()
    _check_function(dump, "INEQ", h.numpy(), Jh)
    swing = (np.abs(dump["INEQ_JAC"]).reshape(120, 3, dec).sum(axis=(1, 2)) == 0) | (np.abs(Jh).reshape(120, 3, dec)[:, 0].sum(axis=1) == 0)
    assert swing.any() and (~swing).any()  # legs in swing (s = 0: the force rows vanish) and in stance both occur
    H = torch.autograd.functional.hessian(lambda zz: O.quadruped_whole_horizon(zz, par)[0], z, vectorize=True).numpy()
    Hd, Hm = dump["OBJ_HES"], dump["OBJ_HES_MASK"]
    assert Hd.shape == (dec, dec)
    assert not Hm[np.tril_indices(dec, -1)].any(), "Hessian must be upper-triangular only (function.hpp:232-235)"
    assert np.abs(Hd - np.triu(H)).max() <= 1e-12 * np.abs(H).max()
    assert not (np.triu(H) != 0)[~Hm].any()


@pytest.fixture(scope="module")
def rc_car_dump(repo_root, tmp_path_factory, shared_codegen):
    exe = os.path.join(repo_root, "build", "batched_rc_car_test")
    assert os.path.exists(exe), "build/batched_rc_car_test missing: run __graft_entry__.build()"
    d = tmp_path_factory.mktemp("rc_car_ocp")
    r = subprocess.run([exe, str(shared_codegen("batched_rc_car")), "4", "-1", str(d / "dump.txt")], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "DUMPED" in r.stdout
    return _parse(str(d / "dump.txt"))


def test_rc_car_whole_horizon_functions_against_the_independent_oracle(rc_car_dump):
    """The reference's RC-car OCP as written (rc_car.example.cpp:191-285; Pacejka tyre model, atan / sin chains, Utils::Abs on the inputs, the minimum-velocity row)
    through the facade's Ungar::Autodiff::Function on the GPU: objective value / gradient / upper-triangular Hessian, the 186 equality rows and the 90 inequality rows
    with their Jacobians, against torch autograd on oracle.ungar_oracle.rc_car_whole_horizon at an instance that crawls at the 0.3 m/s bound."""
    dump = rc_car_dump
    dec, par_n = 246, 83
    v = dump["INPUT"]
    assert v.size == dec + par_n
    z, par = torch.tensor(v[:dec]), torch.tensor(v[dec:])
    value, eq, ineq = O.rc_car_whole_horizon(z, par)
    assert eq.shape == (186,) and ineq.shape == (90,)
    assert abs(dump["OBJ"][0] - value.item()) <= 1e-12 * abs(value.item())
    zg = z.clone().requires_grad_(True)
    (grad,) = torch.autograd.grad(O.rc_car_whole_horizon(zg, par)[0], zg)
    g = dump["OBJ_JAC"]
    assert g.shape == (1, dec) and np.abs(g[0] - grad.numpy()).max() <= 1e-12 * np.abs(grad.numpy()).max()
    H = torch.autograd.functional.hessian(lambda zz: O.rc_car_whole_horizon(zz, par)[0], z, vectorize=True).numpy()
    Hd, Hm = dump["OBJ_HES"], dump["OBJ_HES_MASK"]
    assert not Hm[np.tril_indices(dec, -1)].any() and np.abs(Hd - np.triu(H)).max() <= 1e-12 * np.abs(H).max()
    Je = torch.autograd.functional.jacobian(lambda zz: O.rc_car_whole_horizon(zz, par)[1], z, vectorize=True).numpy()
    _check_function(dump, "EQ", eq.numpy(), Je)
    Ji = torch.autograd.functional.jacobian(lambda zz: O.rc_car_whole_horizon(zz, par)[2], z, vectorize=True).numpy()
    _check_function(dump, "INEQ", ineq.numpy(), Ji)
    assert (np.abs(ineq.numpy()[2::3]) < 0.1).any(), "the instance is meant to sit near the minimum-velocity bound"
    assert (v[(30 + 1) * 6:dec:2] > 0).any() and (v[(30 + 1) * 6:dec:2] < 0).any()  # both branches of Utils::Abs on the duty cycle
