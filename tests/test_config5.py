"""GPU tests (-m gpu) of BASELINE.json configs[4]: the ANYmal-class quadruped, N = 20, a FIXED batch of 65 536 instances
partitioned over the GPUs of a node.  One GPU is enough to pin the path:

  * the shard a rank owns at G = 8 (8 192 instances x 20 = 163 840 nodes) and the WHOLE batch on one device
    (1 310 720 nodes, 19.9 GB of Jacobians) are evaluated through the C ABI and checked against the oracle on a slice,
    against the same nodes evaluated in a small launch (bit-exact: the result of a node does not depend on the launch it
    is part of), and through the checksum-of-checksums property bench.py reduces over ranks (SUM over the shards of
    shard_range == whole-batch checksum, for G = 2, 4, 8);
  * unit-fastest operands of more than 2.37 M nodes (element offsets beyond 32 bits) stay on the lane-per-leg kernel
    (64-bit-offset variant, quad_anymal_wide.hip): bit-identical to the 32-bit-offset kernel on the same nodes.
"""
import numpy as np
import pytest

from oracle import ungar_oracle as O

from helpers import same_kernel_results

pytestmark = pytest.mark.gpu

N = 20


@pytest.fixture(scope="module")
def ua():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ungar_amd
    ungar_amd.load_library()
    return ungar_amd


def _evaluate(ua, torch, x, u, p, count, mode="dense", knots=N):
    m = ua.NodeModel("anymal")
    f = torch.full((m.nx, count), float("nan"), dtype=torch.float64, device="cuda")
    J = torch.full((m.nx * (m.nx + m.nu) if mode == "dense" else m.jac_nnz, count), float("nan"), dtype=torch.float64, device="cuda")
    Op = ua.Operand
    call = m.dense_jacobian if mode == "dense" else m.sparse_jacobian
    call(count, Op.soa(x, count, knots), Op.soa(u, count, knots), None, Op.per_instance(p, m.np, shared=True), Op.soa(f, count, knots), Op.soa(J, count, knots),
         knots=knots)
    torch.cuda.synchronize()
    return f, J


def _oracle_slice(x, u, p, idx):
    xs, us = x[:, idx].t().cpu().numpy(), u[:, idx].t().cpu().numpy()
    ps = np.tile(p.cpu().numpy(), (len(idx), 1))
    return O.node_jacobian("anymal", xs, us, np.zeros((len(idx), 0)), ps)


@pytest.mark.parametrize("instances", [8192, 65536])
def test_config5_shard_and_whole_batch(ua, instances):
    import torch
    from ungar_amd import workloads as W
    from ungar_amd.sharding import shard_range
    count = instances * N
    x, u, _, p = W.synth_device_inputs("anymal", count, 5, torch)
    f, J = _evaluate(ua, torch, x, u, p, count)
    assert torch.isfinite(f).all() and torch.isfinite(J).all()
    # (1) oracle on a slice spread over the whole launch (first / last wavefronts included)
    # (the whole batch is checked node by node in (1b): its torch-oracle slice is the two end wavefronts and a few nodes in between)
    idx = np.unique(np.concatenate((np.arange(8), np.linspace(0, count - 1, 24 if instances == 8192 else 8).astype(np.int64), np.arange(count - 8, count))))
    rf, rJ = _oracle_slice(x, u, p, idx)
    gf = f[:, idx].t().cpu().numpy()
    gJ = J[:, idx].t().cpu().numpy().reshape(len(idx), 37, 49)
    scale = np.abs(rJ).max(axis=(1, 2), keepdims=True)
    assert np.abs(gf - rf).max() <= 1e-10 * max(1.0, np.abs(rf).max())
    assert (np.abs(gJ - rJ) <= 1e-9 * scale).all()
    # (1b) the whole batch: EVERY one of its 1 310 720 nodes against the oracle's generated C, 81 920 nodes at a time
    if instances == 65536:
        from c_checker import compare_launch_with_c_checker
        compare_launch_with_c_checker("anymal", x, u, p, f, J, count, chunk=81920)
    # (2) a node's result does not depend on the launch it is part of: re-evaluate 4 096 nodes from the middle alone
    lo = (count // 2 // N) * N
    sub = slice(lo, lo + 4096 * 1)
    sub_count = 4096
    f2, J2 = _evaluate(ua, torch, x[:, sub].contiguous(), u[:, sub].contiguous(), p, sub_count, knots=1)
    assert same_kernel_results(f2, f[:, sub], "value") and same_kernel_results(J2, J[:, sub], "Jacobian")
    # (3) checksum of checksums: what bench.py reduces with SUM over ranks equals the whole-batch checksum
    whole = f.sum(dtype=torch.float64) + J.sum(dtype=torch.float64)
    absum = float(f.abs().sum() + J.abs().sum())
    for world in (2, 4, 8):
        parts = 0.0
        nodes = 0
        for rank in range(world):
            b, e = shard_range(instances, world, rank)
            parts += float(f[:, b * N:e * N].sum() + J[:, b * N:e * N].sum())
            nodes += (e - b) * N
        assert nodes == count
        assert abs(parts - float(whole)) <= 1e-12 * absum
    # (4) exponential-map update keeps the base quaternion unit
    assert torch.allclose(f[3:7].norm(dim=0), torch.ones(count, dtype=torch.float64, device="cuda"), atol=1e-9)


@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_operands_beyond_32bit_offsets_keep_the_quad_kernel(ua, mode):
    """2 400 000 nodes in one unit-fastest operand: (row * 49 + col) * stride exceeds 2^32 (the r01 launcher fell back to
    the lane-per-node kernel at 12 % of the roofline there).  The 64-bit-offset variant must agree BIT FOR BIT with the
    32-bit-offset kernel evaluated on slices of the same nodes, at both ends of the operand and in the middle."""
    import torch
    from ungar_amd import workloads as W
    count = 2_400_000
    assert count * 1813 >= 2 ** 32
    x, u, _, p = W.synth_device_inputs("anymal", count, 6, torch)
    f, J = _evaluate(ua, torch, x, u, p, count, mode=mode, knots=1)
    assert torch.isfinite(f).all()
    for lo in (0, count // 2 - 37, count - 5000):
        sl = slice(lo, lo + 5000)
        f2, J2 = _evaluate(ua, torch, x[:, sl].contiguous(), u[:, sl].contiguous(), p, 5000, mode=mode, knots=1)
        assert same_kernel_results(f2, f[:, sl], "value") and same_kernel_results(J2, J[:, sl], "Jacobian")
    assert torch.isfinite(J).all()
    del J
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["dense", "sparse"])
def test_padded_element_stride_is_bit_identical(ua, mode):
    """bench.py stores every unit-fastest operand with ungar_amd.sharding.padded_stride (element stride rotated off the
    2^17-byte channel stride): the result of a node must not depend on the stride -- same bits as with stride = nodes, at the
    headline size (4096 instances x 20 knots), with the padding columns of the outputs left untouched."""
    import torch
    from ungar_amd import workloads as W
    from ungar_amd.sharding import padded_stride
    count = 4096 * N
    st = padded_stride(count)
    assert st % 16 == 0 and st >= count + 32 and (st // 16) % 4 == 3
    m = ua.NodeModel("anymal")
    x, u, _, p = W.synth_device_inputs("anymal", count, 4, torch)
    f0, J0 = _evaluate(ua, torch, x, u, p, count, mode)

    def padded(rows, src=None):
        t = torch.full((rows, st), float("nan"), dtype=torch.float64, device="cuda")
        if src is not None:
            t[:, :count] = src
        return t
    xp, up = padded(m.nx, x), padded(m.nu, u)
    f, J = padded(m.nx), padded(J0.shape[0])
    Op = ua.Operand
    call = m.dense_jacobian if mode == "dense" else m.sparse_jacobian
    call(count, Op.soa(xp, st, N), Op.soa(up, st, N), None, Op.per_instance(p, m.np, shared=True), Op.soa(f, st, N), Op.soa(J, st, N), knots=N)
    torch.cuda.synchronize()
    assert same_kernel_results(f[:, :count], f0, "value") and same_kernel_results(J[:, :count], J0, "Jacobian")
    assert torch.isnan(f[:, count:]).all() and torch.isnan(J[:, count:]).all()
    if mode == "dense":  # the Gauss-Newton contraction reads the padded Jacobian and writes a padded G: same bits as from the unpadded operands
        rows, cols = m.nx, m.nx + m.nu
        d0 = torch.rand((rows, count), dtype=torch.float64, device="cuda")
        d = padded(rows, d0)
        G0 = torch.full((cols * cols, count), float("nan"), dtype=torch.float64, device="cuda")
        G = padded(cols * cols)
        ua.gn_hessian_tiles(J0, d0, G0, rows, cols, count, True)
        ua.gn_hessian_tiles(J, d, G, rows, cols, count, True)
        torch.cuda.synchronize()
        upper = torch.triu(torch.ones((cols, cols), dtype=torch.bool, device="cuda")).reshape(-1)
        assert torch.equal(G[upper][:, :count], G0[upper]) and torch.isnan(G[:, count:]).all() and torch.isnan(G[~upper]).all()


@pytest.mark.parametrize("count", [4096 * N, 4090, 34])
def test_paired_stores_are_bit_identical_to_the_eight_byte_kernel(ua, count):
    """Where consecutive nodes lie at consecutive addresses and their number is even, the dense ANYmal kernel writes two entries of a column per store instruction
    (partner nodes exchange one value each with v_permlane16_swap).  An ODD number of nodes takes the kernel with 8-byte stores: the
    same nodes evaluated as part of an even launch (the headline size; an even count that is not a multiple of a wavefront's 16 nodes; two wavefronts' worth) and as
    an odd launch must agree bit for bit, and no entry may be left unwritten."""
    import torch
    from ungar_amd import workloads as W
    x, u, _, p = W.synth_device_inputs("anymal", count, 9, torch)
    f, J = _evaluate(ua, torch, x, u, p, count, knots=1)
    assert torch.isfinite(f).all() and torch.isfinite(J).all()  # (the outputs start as NaN)
    odd = count - 1
    f2, J2 = _evaluate(ua, torch, x[:, :odd].contiguous(), u[:, :odd].contiguous(), p, odd, knots=1)
    assert same_kernel_results(f2, f[:, :odd], "value") and same_kernel_results(J2, J[:, :odd], "Jacobian")
    f3, J3 = _evaluate(ua, torch, x[:, 1:].contiguous(), u[:, 1:].contiguous(), p, odd, knots=1)  # (shifted by one node: other partners)
    assert same_kernel_results(f3, f[:, 1:], "value") and same_kernel_results(J3, J[:, 1:], "Jacobian")
