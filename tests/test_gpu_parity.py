"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  (a) the committed golden vectors (independent torch-autograd oracle, tests/golden/node_*.npz),
  (b) the oracle evaluated live on the same seeded inputs,
  (c) size-independent properties at BASELINE.json's full sizes.
Tolerance for Jacobian entries: BASELINE.json north_star "<=1e-6 relative"; we assert a far tighter
1e-9 relative to the block's max magnitude (FP64 end to end), and 1e-6 *entrywise* relative on every
entry above 1e-6 of the block scale.
"""
import numpy as np
import pytest

from oracle import ungar_oracle as O

from helpers import same_kernel_results

pytestmark = pytest.mark.gpu

MODELS = ("quadrotor", "rc_car", "srbd", "anymal", "anymal_ad", "anymal_reg")


def _oracle_name(name):
    """'anymal_ad' is the same function as 'anymal' with derivatives by taped ABA."""
    return name.replace("_ad", "").replace("_reg", "")


def _assert_close(name, got_f, got_J, ref_f, ref_J):
    assert np.isfinite(got_f).all() and np.isfinite(got_J).all(), f"{name}: non-finite output"
    scale_f = max(1.0, np.abs(ref_f).max())
    assert np.abs(got_f - ref_f).max() <= 1e-10 * scale_f, f"{name}: value mismatch {np.abs(got_f - ref_f).max()}"
    scale = np.abs(ref_J).max(axis=(1, 2), keepdims=True)
    err = np.abs(got_J - ref_J)
    assert (err <= 1e-9 * scale).all(), f"{name}: Jacobian mismatch {err.max()} (scale {scale.max()})"
    big = np.abs(ref_J) > 1e-6 * scale
    rel = err[big] / np.abs(ref_J[big])
    assert rel.max() <= 1e-6, f"{name}: entrywise relative error {rel.max()} > 1e-6"


@pytest.fixture(scope="module")
def ua():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ungar_amd
    ungar_amd.load_library()  # fails loudly if the HIP extension is missing
    return ungar_amd


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_golden_vectors(ua, repo_root, name, layout, mode):
    g = np.load(f"{repo_root}/tests/golden/node_{_oracle_name(name)}.npz")
    m = ua.NodeModel(name)
    f, J = m.evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode=mode, layout=layout)
    _assert_close(name, f, J, g["f"], g["J"])


@pytest.mark.parametrize("name", ["anymal", "anymal_ad", "anymal_reg"])
@pytest.mark.parametrize("layout,mode", [("soa", "dense"), ("soa", "sparse"), ("aos", "dense"), ("aos", "sparse")])
def test_golden_vectors_256_anymal_nodes(ua, repo_root, name, layout, mode):
    """The headline kernel (lane per leg) and the two comparison kernels against 256 nodes of the INDEPENDENT oracle (torch autograd over a spatial-algebra
    restatement; tests/golden/make_anymal_many.py, generated in the build container): every entry of f and of the 37 x 49 block within 1e-9 of the block's
    scale and 1e-6 relative where it is not negligible, all layouts and both Jacobian forms."""
    g = np.load(f"{repo_root}/tests/golden/node_anymal_256.npz")
    assert g["x"].shape[0] == 256
    f, J = ua.NodeModel(name).evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode=mode, layout=layout)
    _assert_close(name, f, J, g["f"], g["J"])


@pytest.mark.parametrize("name", ["quadrotor", "rc_car", "anymal", "anymal_ad", "anymal_reg"])
@pytest.mark.parametrize("layout,mode", [("soa", "dense"), ("aos", "sparse")])
def test_golden_vectors_on_the_edge_cases_of_the_helpers(ua, repo_root, name, layout, mode):
    """Nodes that sit ON the switching / degenerate points of the reference's helper functions (tests/golden/make_edge_cases.py, generated in the build container from
    the independent torch oracle): body angular velocity and net torque exactly 0 so that Utils::ApproximateExponentialMap is evaluated at the zero vector
    (utils.hpp:731-749; test/autodiff/function.test.cpp:40-58 pins that point for the bare helper only), stored quaternions of length 1.3 and 0.7 (the Lie-group
    integrator does not normalise, quadrotor.example.cpp:184-187), the RC car at v_x = 0.5 / 0.3 with zero slip-angle arguments (rc_car.example.cpp:158-161), ANYmal at
    rest with the joints at 0 -- through every kernel of the model."""
    g = np.load(f"{repo_root}/tests/golden/node_{_oracle_name(name)}_edge.npz")
    f, J = ua.NodeModel(name).evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode=mode, layout=layout)
    _assert_close(name, f, J, g["f"], g["J"])


@pytest.mark.parametrize("name", MODELS)
def test_forward_zero_matches_golden(ua, repo_root, name):
    g = np.load(f"{repo_root}/tests/golden/node_{_oracle_name(name)}.npz")
    f, _ = ua.NodeModel(name).evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode="value")
    assert np.abs(f - g["f"]).max() <= 1e-10 * max(1.0, np.abs(g["f"]).max())


@pytest.mark.parametrize("name,count", [("quadrotor", 200), ("rc_car", 300), ("srbd", 100), ("anymal", 32), ("anymal_ad", 12), ("anymal_reg", 12)])
def test_live_oracle_seeded(ua, name, count):
    """Ragged count (not a multiple of the wavefront/block size) on fresh seeded inputs."""
    x, u, w, p = O.synthetic_inputs(_oracle_name(name), count, seed=123)
    rf, rJ = O.node_jacobian(_oracle_name(name), x, u, w, p)
    f, J = ua.NodeModel(name).evaluate_numpy(x, u, w, p, mode="dense", layout="soa")
    _assert_close(name, f, J, rf, rJ)


def test_sparsity_matches_reference_probe_counts(ua):
    """Structural nnz of the per-node [A|B] blocks (quadrotor 118, rc_car 32,
    SRBD 238) and canonical CSR ordering."""
    for name, nnz in (("quadrotor", 118), ("rc_car", 32), ("srbd", 238)):
        m = ua.NodeModel(name)
        assert m.jac_nnz == nnz
        rows, cols = m.jacobian_sparsity()
        key = rows.astype(np.int64) * (m.nx + m.nu) + cols
        assert (np.diff(key) > 0).all()
        starts, outer = m.jacobian_csr()
        assert starts[0] == 0 and starts[-1] == nnz and len(starts) == m.ny + 1


def test_empty_batch_and_errors(ua):
    import torch
    m = ua.NodeModel("rc_car")
    z = torch.zeros(1, dtype=torch.float64, device="cuda")
    op = ua.Operand.aos(z, 1)
    m.dense_jacobian(0, op, op, None, op, op, op)  # count == 0 is a no-op
    with pytest.raises(ua.UngarError):
        ua.NodeModel("no_such_model")
    with pytest.raises(ua.UngarError):
        m.dense_jacobian(4, op, op, None, op, op, None)  # missing jac operand
    with pytest.raises(ua.UngarError):
        m.dense_jacobian(5, op, op, None, op, op, op, knots=2)  # count not a multiple of knots


# ----------------------------------------------------------------------- full-size properties
FULL = {  # BASELINE.json configs[1..3]
    "quadrotor": (4096, 128),
    "rc_car": (16384, 200),
    "anymal": (4096, 20),
    "anymal_ad": (4096, 20),
    "anymal_reg": (4096, 20),
}


def _device_inputs(name, count, seed):
    import torch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    name = _oracle_name(name)
    nx, nu, nw, npar = O.DIMS[name]
    x = torch.rand((nx, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    u = torch.rand((nu, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    if name in ("quadrotor", "anymal"):
        q = x[3:7]
        x[3:7] = q / q.norm(dim=0, keepdim=True)
    if name == "quadrotor":
        u = (u * 0.5 + 1.0) * 15.0
    if name == "rc_car":
        x[3] = x[3] * 0.75 + 1.25
        u[1] = u[1] * 0.3
    if name == "anymal":
        u = u * 20.0
    p = torch.as_tensor(O.default_params(name), device="cuda")
    return x, u, p


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_properties(ua, name):
    """At the BASELINE sizes: (1) unit-fastest (SoA) and node-major (AoS) device layouts agree
    bit-for-bit; (2) sparse CSR values equal the dense block at the pattern and the dense block is
    exactly zero elsewhere; (3) evaluating the batch in two halves reproduces the whole bit-for-bit;
    (4) directional derivative J*delta matches a central difference of f (linearisation property);
    (5) the exponential-map update keeps quaternions unit (where the state has one)."""
    import torch
    batch, N = FULL[name]
    count = batch * N
    m = ua.NodeModel(name)
    nx, nu, ncols = m.nx, m.nu, m.nx + m.nu
    x, u, p = _device_inputs(name, count, seed=11)
    P = ua.Operand.per_instance(p, m.np, shared=True)
    f = torch.empty((nx, count), dtype=torch.float64, device="cuda")
    J = torch.empty((nx * ncols, count), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, P, ua.Operand.soa(f, count), ua.Operand.soa(J, count))
    torch.cuda.synchronize()
    assert torch.isfinite(f).all() and torch.isfinite(J).all()

    # (2) sparse vs dense
    Js = torch.empty((m.jac_nnz, count), dtype=torch.float64, device="cuda")
    f2 = torch.empty_like(f)
    m.sparse_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, P, ua.Operand.soa(f2, count), ua.Operand.soa(Js, count))
    rows, cols = m.jacobian_sparsity()
    idx = torch.as_tensor(rows.astype(np.int64) * ncols + cols, device="cuda")
    if name == "anymal":  # dense block and CSR values: two instantiations of the lane-per-leg program (FMA contraction may differ)
        assert (f - f2).abs().max().item() < 1e-10
        assert ((J[idx] - Js).abs() / J.abs().amax(dim=0, keepdim=True)).max().item() < 1e-9
    else:
        assert torch.equal(J[idx], Js) and torch.equal(f, f2)
    mask = torch.ones(nx * ncols, dtype=torch.bool, device="cuda")
    mask[idx] = False
    assert (J[mask] == 0).all()

    # (1) AoS layout on a slice (keeps the transposed copies small)
    sl = min(count, 64 * 1024)
    xa, ua_ = x[:, :sl].t().contiguous(), u[:, :sl].t().contiguous()
    fa = torch.empty((sl, nx), dtype=torch.float64, device="cuda")
    Ja = torch.empty((sl, nx * ncols), dtype=torch.float64, device="cuda")
    m.dense_jacobian(sl, ua.Operand.aos(xa, nx), ua.Operand.aos(ua_, nu), None, P, ua.Operand.aos(fa, nx), ua.Operand.aos(Ja, nx * ncols))
    assert same_kernel_results(fa.t(), f[:, :sl], "value") and same_kernel_results(Ja.t(), J[:, :sl], "Jacobian")

    # (3) halves: second half evaluated alone, through offset views
    h = count // 2
    fh = torch.empty((nx, count - h), dtype=torch.float64, device="cuda")
    Jh = torch.empty((nx * ncols, count - h), dtype=torch.float64, device="cuda")
    xh, uh = x[:, h:].contiguous(), u[:, h:].contiguous()
    m.dense_jacobian(count - h, ua.Operand.soa(xh, count - h), ua.Operand.soa(uh, count - h), None, P, ua.Operand.soa(fh, count - h),
                     ua.Operand.soa(Jh, count - h))
    assert same_kernel_results(fh, f[:, h:], "value") and same_kernel_results(Jh, J[:, h:], "Jacobian")

    # (4) linearisation: f(z + e d) - f(z - e d) = 2 e J d + O(e^3)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    d = torch.rand((ncols, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    eps = 1e-6
    fp, fm = torch.empty_like(f), torch.empty_like(f)
    for sign, out in ((1.0, fp), (-1.0, fm)):
        xx, uu = x + sign * eps * d[:nx], u + sign * eps * d[nx:]
        m.forward_zero(count, ua.Operand.soa(xx, count), ua.Operand.soa(uu, count), None, P, ua.Operand.soa(out, count))
    fd = (fp - fm) / (2 * eps)
    Jd = torch.einsum("rcn,cn->rn", J.view(nx, ncols, count), d)
    scale = Jd.abs().max().clamp(min=1.0)
    assert ((fd - Jd).abs().max() / scale).item() < 1e-6

    # (5) unit quaternion preserved by q+ = q * exp~(dt w+)
    if name in ("quadrotor", "anymal", "anymal_ad", "anymal_reg"):
        assert (f[3:7].norm(dim=0) - 1).abs().max().item() < 1e-12


@pytest.mark.parametrize("name", ["anymal", "quadrotor", "rc_car"])
def test_full_size_launch_against_the_c_checker_on_every_node(ua, repo_root, name):
    """BASELINE-size launch (anymal: 81 920 nodes = config 4; quadrotor: 524 288 = config 1; rc_car: 3 276 800 = config 2), EVERY node compared with the oracle's
    generated C (oracle/_gen/*_cg.c, compiled by __graft_entry__.build(); pinned against the independent torch oracle's golden vectors
    by tests/test_codegen_c.py).  The C body is lowered from the same tape as the taped kernels but shares nothing with the launch
    geometry, the lane-per-leg program, the operand addressing or the store path of the kernel under test (tests/c_checker.py)."""
    import torch
    from c_checker import compare_launch_with_c_checker
    batch, N = FULL[name]
    count = batch * N
    m = ua.NodeModel(name)
    nx, ncols = m.nx, m.nx + m.nu
    x, u, p = _device_inputs(name, count, seed=23)
    P = ua.Operand.per_instance(p, m.np, shared=True)
    f = torch.empty((nx, count), dtype=torch.float64, device="cuda")
    J = torch.empty((nx * ncols, count), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, P, ua.Operand.soa(f, count), ua.Operand.soa(J, count))
    torch.cuda.synchronize()
    compare_launch_with_c_checker(name, x, u, p, f, J, count)
    # The C above is lowered from the product's own tape: a launch-geometry / store-path check.  Against the INDEPENDENT oracle (torch, oracle/ungar_oracle.py) the
    # same launch is held on 512 nodes spread over it: first and last wavefronts, both sides of the wavefront (64 nodes / 16 for the lane-per-leg kernel), pair
    # partner, generation (256 CUs x 4 x 64 lanes) and tile boundaries, and random ones.
    rng = np.random.default_rng(11)
    edges = np.array([0, 1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 1023, 1024, 1025, 16383, 16384, 16385, 65535, 65536, 65537, count - 65, count - 64, count - 17, count - 16, count - 2, count - 1])
    sample = np.unique(np.concatenate([edges[edges < count], np.arange(64), count - 1 - np.arange(64), rng.integers(0, count, 600)]))
    sample = np.sort(rng.permutation(sample)[:512])
    xs, us = x[:, sample].t().contiguous().cpu().numpy(), u[:, sample].t().contiguous().cpu().numpy()
    rf, rJ = O.node_jacobian_batched(_oracle_name(name), xs, us, np.zeros((len(sample), O.DIMS[_oracle_name(name)][2])), np.tile(p.cpu().numpy(), (len(sample), 1)))
    _assert_close(name, f[:, sample].t().cpu().numpy(), J[:, sample].t().cpu().numpy().reshape(len(sample), nx, ncols), rf, rJ)


def test_anymal_value_only_program_layouts_and_ragged_counts(ua):
    """forward_zero of the 'anymal' model runs the lane-per-leg VALUE program (16 nodes per wavefront): a count that is not a multiple of
    16, both device layouts, knots > 1 -- bit-identical among themselves and equal to the values the value + Jacobian kernel writes up to
    the rounding of differently contracted multiply-adds (the two bodies are compiled separately; in the 4-lane simulator they agree bit
    for bit, tests/test_quad_program.py)."""
    import torch
    m = ua.NodeModel("anymal")
    batch, N = 331, 3
    count = batch * N  # 993 = 62 * 16 + 1
    x, u, p = _device_inputs("anymal", count, seed=31)
    P = ua.Operand.per_instance(p, m.np, shared=True)
    f_ref = torch.empty((m.nx, count), dtype=torch.float64, device="cuda")
    J = torch.empty((m.nx * (m.nx + m.nu), count), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, P, ua.Operand.soa(f_ref, count), ua.Operand.soa(J, count))
    f = torch.full((m.nx, count + 7), float("nan"), dtype=torch.float64, device="cuda")
    m.forward_zero(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, P, ua.Operand.soa(f, count + 7))
    assert ((f[:, :count] - f_ref).abs().max() <= 1e-11 * f_ref.abs().max().clamp(min=1.0)).item()
    assert torch.isnan(f[:, count:]).all()  # nothing written past the last node
    f_ref = f[:, :count].clone()
    xa, ua_ = x.t().contiguous(), u.t().contiguous()
    fa = torch.empty((count, m.nx), dtype=torch.float64, device="cuda")
    m.forward_zero(count, ua.Operand.aos(xa, m.nx), ua.Operand.aos(ua_, m.nu), None, P, ua.Operand.aos(fa, m.nx))
    assert torch.equal(fa.t(), f_ref)
    fk = torch.empty((batch, N, m.nx), dtype=torch.float64, device="cuda")  # (instance, knot, element) with explicit knot strides
    m.forward_zero(count, ua.Operand(xa, instance_stride=N * m.nx, knot_stride=m.nx, element_stride=1), ua.Operand(ua_, instance_stride=N * m.nu, knot_stride=m.nu, element_stride=1),
                   None, P, ua.Operand(fk, instance_stride=N * m.nx, knot_stride=m.nx, element_stride=1), knots=N)
    assert torch.equal(fk.reshape(count, m.nx).t(), f_ref)


def test_structured_and_taped_aba_kernels_agree_at_full_size(ua):
    """'anymal' (implicit differentiation: CRBA + RNEA tangents + U D U^T solves) and 'anymal_ad'
    (derivatives by taping ABA, the reference's route) are different algorithms for the same
    Jacobian: they must agree to rounding on the whole BASELINE batch."""
    import torch
    count = 4096 * 20
    x, u, p = _device_inputs("anymal", count, seed=21)
    out = {}
    for name in ("anymal", "anymal_ad", "anymal_reg"):
        m = ua.NodeModel(name)
        f = torch.empty((37, count), dtype=torch.float64, device="cuda")
        J = torch.empty((37 * 49, count), dtype=torch.float64, device="cuda")
        m.dense_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, ua.Operand.per_instance(p, 1, shared=True),
                         ua.Operand.soa(f, count), ua.Operand.soa(J, count))
        out[name] = (f, J)
    torch.cuda.synchronize()
    f1, J1 = out["anymal_ad"]
    scale = J1.abs().amax(dim=0, keepdim=True)
    for other in ("anymal", "anymal_reg"):
        f0, J0 = out[other]
        assert (f0 - f1).abs().max().item() < 1e-10
        assert ((J0 - J1).abs() / scale).max().item() < 1e-9


def test_anymal_trajectory_layout_with_knots(ua):
    """Per-instance VariableMap-style buffers [X | U] (instance / knot strides, knots > 1) through the
    lane-per-leg kernel, against the oracle; ragged node count (not a multiple of 16 nodes per wavefront)."""
    import torch
    N, batch = 5, 7
    count = N * batch
    x, u, w, p = O.synthetic_inputs("anymal", count, seed=77)
    rf, rJ = O.node_jacobian("anymal", x, u, w, p)
    dev = torch.device("cuda")
    # instance-major trajectory block: (N+1) states (the last one unused by the nodes) then N inputs
    traj = torch.zeros((batch, (N + 1) * 37 + N * 12), dtype=torch.float64, device=dev)
    traj[:, :N * 37] = torch.as_tensor(x, device=dev).reshape(batch, N * 37)
    traj[:, (N + 1) * 37:] = torch.as_tensor(u, device=dev).reshape(batch, N * 12)
    stride = traj.shape[1]
    f = torch.full((count, 37), float("nan"), dtype=torch.float64, device=dev)
    J = torch.full((count, 37 * 49), float("nan"), dtype=torch.float64, device=dev)
    pd = torch.as_tensor(p[0], device=dev)
    m = ua.NodeModel("anymal")
    m.dense_jacobian(count, ua.Operand(traj, stride, 37, 1), ua.Operand(traj[:, (N + 1) * 37:], stride, 12, 1), None,
                     ua.Operand.per_instance(pd, 1, shared=True), ua.Operand.aos(f, 37, N), ua.Operand.aos(J, 37 * 49, N), knots=N)
    torch.cuda.synchronize()
    _assert_close("anymal", f.cpu().numpy(), J.cpu().numpy().reshape(count, 37, 49), rf, rJ)


def test_gn_hessian_mfma(ua):
    """J^T diag(d) J on the FP64 matrix cores vs torch (rows=37, cols=49 is the ANYmal block; also a
    ragged small case and the identity-weight path)."""
    import torch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    for rows, cols, count, weighted in ((37, 49, 1000, True), (13, 17, 257, True), (6, 8, 5, False), (37, 49, 3, False)):
        J = torch.rand((count, rows, cols), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
        d = torch.rand((count, rows), generator=gen, device="cuda", dtype=torch.float64) if weighted else None
        G = torch.full((count, cols, cols), float("nan"), dtype=torch.float64, device="cuda")
        ua.gn_hessian(J, d, G, rows, cols, count)
        torch.cuda.synchronize()
        ref = torch.einsum("nra,nr,nrb->nab", J, d if weighted else torch.ones((count, rows), device="cuda", dtype=torch.float64), J)
        assert torch.isfinite(G).all()
        assert (G - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
        assert torch.equal(G, G.transpose(1, 2)) or (G - G.transpose(1, 2)).abs().max().item() < 1e-13
        # upper-triangular variant: identical values on and above the diagonal, nothing written below it
        U = torch.full((count, cols, cols), float("nan"), dtype=torch.float64, device="cuda")
        ua.gn_hessian(J, d, U, rows, cols, count, upper_only=True)
        torch.cuda.synchronize()
        upper = torch.triu(torch.ones((cols, cols), dtype=torch.bool, device="cuda"))
        assert torch.equal(U[:, upper], G[:, upper])
        assert torch.isnan(U[:, ~upper]).all()
        # the same contraction fed with the unit-fastest layout the node kernels write
        S = torch.full((count, cols, cols), float("nan"), dtype=torch.float64, device="cuda")
        Jt = J.reshape(count, rows * cols).t().contiguous()
        ua.gn_hessian_unit_fastest(Jt, d.t().contiguous() if weighted else None, S, rows, cols, count)
        torch.cuda.synchronize()
        assert (S[:, upper] - ref[:, upper]).abs().max().item() <= 1e-12 * ref.abs().max().item()
        assert torch.isnan(S[:, ~upper]).all()


@pytest.mark.parametrize("kernel", ["gn_hessian_lanes", "gn_hessian_tiles"])
def test_gn_hessian_lane_per_node(ua, kernel):
    """The FP64-vector contractions for unit-fastest Jacobians -- one lane per node (gn_hessian_lanes.hip) and one lane per
    (node, 7 x 7 block) with LDS-streamed rows (gn_hessian_tiles.hip; widths without a compiled instance are forwarded to the
    former): ragged node counts (not a multiple of 64 / 16), column counts that are not a multiple of the 7 x 7 tile, row
    counts that are not a multiple of the staging depth, weighted and unweighted, both output layouts; entries below the
    diagonal are never written."""
    import torch
    contract = getattr(ua, kernel)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    for rows, cols, count, weighted in ((37, 49, 1000, True), (12, 37, 130, True), (13, 17, 257, False), (6, 8, 5, True), (8, 17, 64, True), (1, 49, 17, False),
                                        (5, 49, 31, True), (9, 23, 40, True), (37, 49, 1002, False), (8, 49, 48, True), (17, 49, 4098, True), (100, 49, 18, True),
                                        (37, 49, 2, True), (3, 37, 34, False), (29, 17, 1026, True)):
        J = torch.rand((rows * cols, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
        d = torch.rand((rows, count), generator=gen, device="cuda", dtype=torch.float64) if weighted else None
        Jv = J.view(rows, cols, count)
        ref = torch.einsum("ran,rn,rbn->nab", Jv, d if weighted else torch.ones((rows, count), device="cuda", dtype=torch.float64), Jv)
        upper = torch.triu(torch.ones((cols, cols), dtype=torch.bool, device="cuda"))
        Guf = torch.full((cols * cols, count), float("nan"), dtype=torch.float64, device="cuda")
        contract(J, d, Guf, rows, cols, count, True)
        Gnm = torch.full((count, cols, cols), float("nan"), dtype=torch.float64, device="cuda")
        contract(J, d, Gnm, rows, cols, count, False)
        torch.cuda.synchronize()
        got = Guf.view(cols, cols, count).permute(2, 0, 1)
        assert (got[:, upper] - ref[:, upper]).abs().max().item() <= 1e-12 * ref.abs().max().item()
        assert torch.isnan(got[:, ~upper]).all()
        assert torch.equal(Gnm[:, upper], got[:, upper]) and torch.isnan(Gnm[:, ~upper]).all()


def test_full_size_chain_node_jacobians_to_gauss_newton_term(ua):
    """BASELINE config 4 at full size (4096 instances x 20 knots = 81 920 nodes): ANYmal node Jacobians (unit-fastest) ->
    G = J^T diag(d) J with no transposition, by ALL THREE contractions (FP64 vector lane per node, lane per (node, block) with
    LDS-streamed rows, LDS-staged MFMA); they agree with each other on every node and with torch on a slice;
    trace(G) = sum_r d_r |J_r|^2 on every node."""
    import torch
    from ungar_amd import workloads as W
    N, batch = 20, 4096
    count = N * batch
    m = ua.NodeModel("anymal")
    rows, cols = m.nx, m.nx + m.nu
    x, u, _, p = W.synth_device_inputs("anymal", count, 2, torch)
    f = torch.empty((rows, count), dtype=torch.float64, device="cuda")
    J = torch.empty((rows * cols, count), dtype=torch.float64, device="cuda")
    Op = ua.Operand
    m.dense_jacobian(count, Op.soa(x, count, N), Op.soa(u, count, N), None, Op.per_instance(p, 1, shared=True), Op.soa(f, count, N), Op.soa(J, count, N), knots=N)
    d = torch.rand((rows, count), device="cuda", dtype=torch.float64) + 0.1
    Guf = torch.zeros((cols * cols, count), dtype=torch.float64, device="cuda")
    ua.gn_hessian_lanes(J, d, Guf, rows, cols, count, True)
    Gnm = torch.zeros((count, cols, cols), dtype=torch.float64, device="cuda")
    ua.gn_hessian_unit_fastest(J, d, Gnm, rows, cols, count)
    Gt = torch.zeros((cols * cols, count), dtype=torch.float64, device="cuda")
    ua.gn_hessian_tiles(J, d, Gt, rows, cols, count, True)
    torch.cuda.synchronize()
    upper = torch.triu(torch.ones((cols, cols), dtype=torch.bool, device="cuda"))
    got = Guf.view(cols, cols, count).permute(2, 0, 1)
    scale = got.abs().amax(dim=(1, 2), keepdim=True)
    assert ((got - Gt.view(cols, cols, count).permute(2, 0, 1))[:, upper].abs() <= 1e-13 * scale.expand(-1, cols, cols)[:, upper]).all()  # same products, same order
    assert ((got - Gnm)[:, upper].abs() <= 1e-11 * scale.expand(-1, cols, cols)[:, upper]).all()  # vector lanes == matrix cores
    Jv = J.view(rows, cols, count)
    sl = slice(count - 2048, count)
    ref = torch.einsum("ran,rn,rbn->nab", Jv[:, :, sl], d[:, sl], Jv[:, :, sl])
    assert ((got[sl] - ref)[:, upper].abs().max() <= 1e-12 * ref.abs().max()).item()
    trace = torch.einsum("naa->n", got)
    assert torch.allclose(trace, (d.unsqueeze(1) * Jv * Jv).sum(dim=(0, 1)), rtol=1e-12, atol=0)
    assert (torch.diagonal(got, dim1=1, dim2=2) >= 0).all()


@pytest.mark.parametrize("name,fixture,nu,npar", [("quadrotor_cost", "cost_quadrotor.npz", 4, 13), ("srbd_cost", "cost_srbd.npz", 24, 25), ("rc_car_cost", "cost_rc_car.npz", 2, 2), ("anymal_cost", "cost_anymal.npz", 12, 42)])
def test_stage_cost_value_gradient_hessian(ua, repo_root, name, fixture, nu, npar):
    """Scalar node models 'quadrotor_cost' / 'srbd_cost' (SURVEY.md section 8(f) N2): value, gradient and
    upper-triangular Hessian w.r.t. (x, u) against torch.autograd on the oracle's restatement of the
    reference's stage costs, in both operand layouts; inputs exercise both branches of the quaternion min."""
    import torch
    g = np.load(f"{repo_root}/tests/golden/{fixture}")
    m = ua.NodeModel(name)
    nx = g["x"].shape[1]
    ncols = nx + nu
    assert (m.nx, m.nu, m.np, m.ny) == (nx, nu, npar, 1) and m.implements_hessian()
    grad_rows, grad_cols = m.jacobian_sparsity()  # rc_car_cost does not depend on every state: its gradient is sparse
    rows, cols = m.hessian_sparsity()
    assert (rows <= cols).all(), "upper triangle only (function.hpp:236-274)"
    count = g["x"].shape[0]
    dev = "cuda"
    for layout in ("soa", "aos"):
        t = (lambda a: torch.as_tensor(a.T.copy(), device=dev)) if layout == "soa" else (lambda a: torch.as_tensor(a.copy(), device=dev))
        shape = (lambda n: (n, count)) if layout == "soa" else (lambda n: (count, n))
        mk = (lambda ten, n: ua.Operand.soa(ten, count)) if layout == "soa" else (lambda ten, n: ua.Operand.aos(ten, n))
        x, u, p = t(g["x"]), t(g["u"]), t(g["p"])
        y = torch.full(shape(1), float("nan"), dtype=torch.float64, device=dev)
        grad = torch.full(shape(ncols), float("nan"), dtype=torch.float64, device=dev)
        hes = torch.full(shape(len(rows)), float("nan"), dtype=torch.float64, device=dev)
        m.sparse_hessian(count, mk(x, nx), mk(u, nu), None, mk(p, npar), mk(y, 1), mk(grad, ncols), mk(hes, len(rows)))
        torch.cuda.synchronize()
        Y, G, H = (a.cpu().numpy().T if layout == "soa" else a.cpu().numpy() for a in (y, grad, hes))
        assert np.abs(Y[:, 0] - g["y"]).max() <= 1e-12 * np.abs(g["y"]).max()
        assert np.abs(G[:, grad_cols] - g["g"][:, grad_cols]).max() <= 1e-12 * np.abs(g["g"]).max()
        off = np.ones(ncols, dtype=bool)
        off[grad_cols] = False
        assert np.abs(g["g"][:, off]).max(initial=0.0) == 0.0 and (np.isnan(G[:, off]) | (G[:, off] == 0.0)).all()  # structural zeros of the gradient
        assert np.abs(H - g["H"][:, rows, cols]).max() <= 1e-12
        dense = np.zeros_like(g["H"])
        dense[:, rows, cols] = H
        assert np.abs(np.triu(g["H"]) - dense).max() <= 1e-12, "entries outside the pattern must be structural zeros"
    # value-only and gradient-only entry points of the same model
    x, u, p = (torch.as_tensor(g[k].T.copy(), device=dev) for k in ("x", "u", "p"))
    y2 = torch.empty((1, count), dtype=torch.float64, device=dev)
    m.forward_zero(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, ua.Operand.soa(p, count), ua.Operand.soa(y2, count))
    g2 = torch.empty((ncols, count), dtype=torch.float64, device=dev)
    m.dense_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, ua.Operand.soa(p, count), ua.Operand.soa(y2, count), ua.Operand.soa(g2, count))
    torch.cuda.synchronize()
    assert np.abs(y2.cpu().numpy()[0] - g["y"]).max() <= 1e-12 * np.abs(g["y"]).max()
    assert np.abs(g2.cpu().numpy().T[:, grad_cols] - g["g"][:, grad_cols]).max() <= 1e-12 * np.abs(g["g"]).max()
    # vector-valued models refuse the Hessian entry point
    with pytest.raises(ua.UngarError):
        ua.NodeModel("quadrotor").hessian_sparsity()


def test_pipeline_node_jacobian_to_gn_term(ua):
    """BASELINE config 3 chain: ANYmal node Jacobians written by the lane-per-leg kernel in the unit-fastest
    layout are consumed by the unit-fastest Gauss-Newton kernel with no transpose in between; the upper
    triangle of J^T diag(d) J must equal the contraction of the very same device Jacobians done by torch."""
    import torch
    m = ua.NodeModel("anymal")
    nx, ncols, count = m.nx, m.nx + m.nu, 1000  # ragged: not a multiple of 16
    x, u, p = _device_inputs("anymal", count, seed=21)
    f = torch.empty((nx, count), dtype=torch.float64, device="cuda")
    J = torch.empty((nx * ncols, count), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, ua.Operand.soa(x, count), ua.Operand.soa(u, count), None, ua.Operand.per_instance(p, m.np, shared=True),
                     ua.Operand.soa(f, count), ua.Operand.soa(J, count))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    d = torch.rand((nx, count), generator=gen, device="cuda", dtype=torch.float64)
    G = torch.full((count, ncols, ncols), float("nan"), dtype=torch.float64, device="cuda")
    ua.gn_hessian_unit_fastest(J, d, G, nx, ncols, count)
    torch.cuda.synchronize()
    Jn = J.view(nx, ncols, count)
    ref = torch.einsum("ran,rn,rbn->nab", Jn, d, Jn)
    upper = torch.triu(torch.ones((ncols, ncols), dtype=torch.bool, device="cuda"))
    assert (G[:, upper] - ref[:, upper]).abs().max().item() <= 1e-12 * ref.abs().max().item()
    assert torch.isnan(G[:, ~upper]).all()


@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_srbd_inequality_node_golden(ua, repo_root, layout, mode):
    """'srbd_ineq': 12 inequality rows per knot (unilateral force, friction cone, leg reach) of the quadruped
    OCP and their Jacobian; an output count different from nx (ny = 12)."""
    g = np.load(f"{repo_root}/tests/golden/node_srbd_ineq.npz")
    m = ua.NodeModel("srbd_ineq")
    assert (m.nx, m.nu, m.nw, m.np, m.ny) == (13, 24, 4, 14, 12)
    f, J = m.evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode=mode, layout=layout)
    assert f.shape == g["f"].shape and J.shape == g["J"].shape
    assert np.abs(f - g["f"]).max() <= 1e-12 * max(1.0, np.abs(g["f"]).max())
    assert np.abs(J - g["J"]).max() <= 1e-12 * np.abs(g["J"]).max()
    assert (g["f"] > 0).any() and (g["f"] < 0).any(), "the fixture must contain violated and satisfied rows"


@pytest.mark.parametrize("name,dims", [("quadrotor_ineq", (13, 4, 0, 1, 8)), ("rc_car_ineq", (6, 2, 0, 0, 3)), ("srbd_feet", (13, 24, 0, 0, 12))])
@pytest.mark.parametrize("layout", ["soa", "aos"])
@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_remaining_ocp_rows_golden(ua, repo_root, name, dims, layout, mode):
    """The remaining per-knot rows of the reference's OCPs as node kernels: rotor-speed bounds (quadrotor.example.cpp:280-288),
    RC-car input bounds through Utils::Abs and minimum forward velocity (rc_car.example.cpp:271-282: both sides of the kink of
    |.| are in the fixture), world foot positions p + q * r_i of the quadruped's contact rows (quadruped.example.cpp:288-291)."""
    g = np.load(f"{repo_root}/tests/golden/node_{name}.npz")
    m = ua.NodeModel(name)
    assert (m.nx, m.nu, m.nw, m.np, m.ny) == dims
    f, J = m.evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode=mode, layout=layout)
    assert f.shape == g["f"].shape and J.shape == g["J"].shape
    assert np.abs(f - g["f"]).max() <= 1e-12 * max(1.0, np.abs(g["f"]).max())
    assert np.abs(J - g["J"]).max() <= 1e-12 * np.abs(g["J"]).max()
    if name != "srbd_feet":
        assert (g["f"] > 0).any() and (g["f"] < 0).any(), "the fixture must contain violated and satisfied rows"


def test_barrier_gauss_newton_term_on_device(ua, repo_root):
    """The Gauss-Newton barrier term of the soft SQP, J_h^T diag(b''(-h)) J_h (soft_sqp.hpp:257-264), per knot and
    entirely on the device: inequality values and Jacobians from the 'srbd_ineq' node kernel, second derivatives
    of the relaxed POLY barrier (soft_inequality_constraint.hpp:135-205) applied elementwise, contraction on the
    FP64 matrix cores; checked against numpy on the oracle's h and J."""
    import torch
    count = 1000
    x, u, w, p = O.synthetic_inputs("srbd_ineq", count, seed=3)
    rh, rJ = O.node_jacobian("srbd_ineq", x[:64], u[:64], w[:64], p[:64])  # oracle on a slice (autograd is slow)
    m = ua.NodeModel("srbd_ineq")
    ny, ncols = m.ny, m.nx + m.nu
    dev = "cuda"
    tx, tu, tw = (torch.as_tensor(a.T.copy(), device=dev) for a in (x, u, w))
    tp = torch.as_tensor(p[0].copy(), device=dev)
    h = torch.empty((ny, count), dtype=torch.float64, device=dev)
    J = torch.empty((ny * ncols, count), dtype=torch.float64, device=dev)
    Op = ua.Operand
    m.dense_jacobian(count, Op.soa(tx, count), Op.soa(tu, count), Op.soa(tw, count), Op.per_instance(tp, m.np, shared=True), Op.soa(h, count), Op.soa(J, count))
    # b''(z) of RelaxedPolyBarrierFunction{rhs 0, stiffness k, epsilon e} at z = -h (the SQP's defaults: 100, 2e-5)
    k, e = 100.0, 2e-5
    a1, b1 = k, -0.5 * k * e
    a2 = (-b1 - a1 * e) / e ** 2
    z = -h
    d2 = torch.where(z < 0.0, torch.full_like(z, a1), torch.where(z < e, 2.0 * a2 * z + a1, torch.zeros_like(z)))
    G = torch.full((count, ncols, ncols), float("nan"), dtype=torch.float64, device=dev)
    ua.gn_hessian_unit_fastest(J, d2, G, ny, ncols, count)
    torch.cuda.synchronize()
    Gh = G[:64].cpu().numpy()
    zr = -rh
    d2r = np.where(zr < 0.0, a1, np.where(zr < e, 2.0 * a2 * zr + a1, 0.0))
    ref = np.einsum("nra,nr,nrb->nab", rJ, d2r, rJ)
    iu = np.triu_indices(ncols)
    assert np.abs(Gh[:, iu[0], iu[1]] - ref[:, iu[0], iu[1]]).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(ref).max() > 1.0, "active barrier rows expected in the fixture"


def test_transpose_nodes_between_the_two_layouts(ua):
    """ungar_transpose_nodes: unit-fastest <-> instance-major, bit-exact, ragged sizes (not multiples of the 64 x 64 tile),
    padded element strides and leading dimensions; cells outside (count, elements) are left untouched."""
    import torch
    from ungar_amd.sharding import padded_stride
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    for count, elements in ((1000, 1813), (64, 64), (1, 1), (130, 49), (4099, 37)):
        st, ld = padded_stride(count), elements + 3
        soa = torch.full((elements, st), float("nan"), dtype=torch.float64, device="cuda")
        soa[:, :count] = torch.rand((elements, count), generator=gen, device="cuda", dtype=torch.float64)
        aos = torch.full((count, ld), float("nan"), dtype=torch.float64, device="cuda")
        ua.transpose_nodes(soa, aos, count, elements, (1, st), (ld, 1))
        torch.cuda.synchronize()
        assert torch.equal(aos[:, :elements], soa[:, :count].t()) and torch.isnan(aos[:, elements:]).all()
        back = torch.full((elements, st), float("nan"), dtype=torch.float64, device="cuda")
        ua.transpose_nodes(aos, back, count, elements, (ld, 1), (1, st))
        torch.cuda.synchronize()
        assert torch.equal(back[:, :count], soa[:, :count]) and torch.isnan(back[:, count:]).all()


def test_instance_major_caller_through_transposes_matches_direct_launch(ua):
    """The recipe of INTEGRATION.md section 4 for instance-major callers of the wide ANYmal Jacobian: transpose (x, u) in, run the
    unit-fastest kernel, transpose (f, J) out -- same bits as launching the kernel on the instance-major operands directly."""
    import torch
    from ungar_amd import workloads as W
    from ungar_amd.sharding import unit_fastest
    count = 2048
    m = ua.NodeModel("anymal")
    ncols = m.nx + m.nu
    xs, us, _, p = W.synth_device_inputs("anymal", count, 6, torch)
    x_aos, u_aos = xs.t().contiguous(), us.t().contiguous()
    Op = ua.Operand
    f_ref = torch.empty((count, m.nx), dtype=torch.float64, device="cuda")
    J_ref = torch.empty((count, m.nx * ncols), dtype=torch.float64, device="cuda")
    m.dense_jacobian(count, Op.aos(x_aos, m.nx), Op.aos(u_aos, m.nu), None, Op.per_instance(p, m.np, shared=True), Op.aos(f_ref, m.nx), Op.aos(J_ref, m.nx * ncols))
    x, u, f, J = (unit_fastest(r, count, torch) for r in (m.nx, m.nu, m.nx, m.nx * ncols))
    st = x.stride(0)
    ua.transpose_nodes(x_aos, x, count, m.nx, (m.nx, 1), (1, st))
    ua.transpose_nodes(u_aos, u, count, m.nu, (m.nu, 1), (1, st))
    m.dense_jacobian(count, Op.soa(x, st), Op.soa(u, st), None, Op.per_instance(p, m.np, shared=True), Op.soa(f, st), Op.soa(J, st))
    f_out, J_out = torch.empty_like(f_ref), torch.empty_like(J_ref)
    ua.transpose_nodes(f, f_out, count, m.nx, (1, st), (m.nx, 1))
    ua.transpose_nodes(J, J_out, count, m.nx * ncols, (1, st), (m.nx * ncols, 1))
    torch.cuda.synchronize()
    assert torch.equal(f_out, f_ref) and torch.equal(J_out, J_ref)
