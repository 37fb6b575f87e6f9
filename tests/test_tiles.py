"""The wave-tile layout of the 'anymal' dense block (include/ungar_amd.h: ungar_tile_layout; csrc/kernels/quad_tile_kernel.hpp).
CPU: the layout table is a bijection onto the dense block, the operand size formula, error behaviour for models without a tile program.
GPU (-m gpu): the tile kernel through the C ABI against the independent torch oracle (golden fixtures, edge cases, live oracle on ragged
counts), against the unit-fastest kernel on every entry, at the BASELINE size against the C checker on every node and against the torch
oracle on a spread sample; the device gather against the host reading of the same operand."""
import numpy as np
import pytest

from oracle import ungar_oracle as O


@pytest.fixture(scope="module")
def lib():
    import ungar_amd
    ungar_amd.load_library()
    return ungar_amd


def test_tile_layout_is_a_bijection_onto_the_dense_block(lib):
    m = lib.NodeModel("anymal")
    l = m.tile_layout()
    assert (l["nodes_per_tile"], l["band_tiles"], l["unit_doubles"], l["entries"]) == (16, 64, 128, 37 * 49)
    assert l["images"] % 2 == 0 and l["unit_doubles"] == 2 * 4 * l["nodes_per_tile"]
    table = l["entry_of_slot"]
    assert table.shape == (4 * l["images"],)
    used = table[table >= 0]
    assert sorted(used.tolist()) == list(range(37 * 49)), "every (row, col) of the dense block exactly once"
    assert (table < 0).sum() == 4 * l["images"] - 37 * 49 == 3, "padding slots"
    # the operand is padded to whole bands of 64 tiles
    per_tile = (l["images"] // 2) * l["unit_doubles"]
    for count, bands in ((1, 1), (16, 1), (1024, 1), (1025, 2), (81920, 80)):
        assert m.tile_doubles(count) == bands * 64 * per_tile
    assert m.tile_doubles(0) == 0
    # bytes written per node: 1816 slots of 8 bytes against 1813 entries (1.0017 x the dense block)
    assert 8 * 4 * l["images"] == 14528


def test_models_without_a_tile_program_say_so(lib):
    for name in ("quadrotor", "rc_car", "srbd", "anymal_ad"):
        m = lib.NodeModel(name)
        with pytest.raises(lib.UngarError, match="no wave-tile program"):
            m.tile_layout()
        with pytest.raises(lib.UngarError):
            m.tile_doubles(16)


def test_untile_reads_what_the_offset_formula_of_the_header_says(lib):
    """Host-side reading (NodeModel.untile_numpy) against a direct transcription of the formula in include/ungar_amd.h, on a synthetic operand."""
    m = lib.NodeModel("anymal")
    l = m.tile_layout()
    count = 64 * 16 + 37  # two bands, ragged last tile
    tiles = np.arange(m.tile_doubles(count), dtype=np.float64)
    J = m.untile_numpy(tiles, count).reshape(count, -1)
    rng = np.random.default_rng(0)
    for _ in range(200):
        i, slot = int(rng.integers(count)), int(rng.integers(4 * l["images"]))
        e = int(l["entry_of_slot"][slot])
        if e < 0:
            continue
        t, n, image, leg = i // 16, i % 16, slot // 4, slot % 4
        lane = 16 * (n // 4) + 4 * leg + n % 4
        off = (((t // 64) * (l["images"] // 2) + image // 2) * 64 + t % 64) * 128 + 2 * lane + image % 2
        assert J[i, e] == tiles[off]


# ------------------------------------------------------------------------------------------------------------------ GPU
def _assert_close(got_f, got_J, ref_f, ref_J):
    assert np.isfinite(got_f).all() and np.isfinite(got_J).all(), "non-finite output"
    assert np.abs(got_f - ref_f).max() <= 1e-10 * max(1.0, np.abs(ref_f).max())
    scale = np.abs(ref_J).max(axis=(1, 2), keepdims=True)
    err = np.abs(got_J - ref_J)
    assert (err <= 1e-9 * scale).all(), f"Jacobian mismatch {err.max()} (scale {scale.max()})"
    big = np.abs(ref_J) > 1e-6 * scale
    assert (err[big] / np.abs(ref_J[big])).max() <= 1e-6  # BASELINE.json north_star: <= 1e-6 relative


@pytest.fixture(scope="module")
def ua(lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["tiles", "tiles_gather"])
@pytest.mark.parametrize("fixture", ["node_anymal.npz", "node_anymal_256.npz", "node_anymal_edge.npz"])
def test_tile_kernel_against_the_golden_vectors(ua, repo_root, layout, fixture):
    """The tile kernel against the INDEPENDENT torch oracle's fixtures (the same ones the unit-fastest kernel is held to), read on the host through the
    layout table alone ("tiles") and through the device gather ("tiles_gather")."""
    g = np.load(f"{repo_root}/tests/golden/{fixture}")
    f, J = ua.NodeModel("anymal").evaluate_numpy(g["x"], g["u"], g["w"], g["p"], mode="dense", layout=layout)
    _assert_close(f, J, g["f"], g["J"])


@pytest.mark.gpu
@pytest.mark.parametrize("count", [1, 3, 15, 16, 17, 100, 1024, 1030])
def test_tile_kernel_on_ragged_counts_against_the_live_oracle_and_the_unit_fastest_kernel(ua, count):
    """Counts that are not a multiple of the tile (16 nodes) or of the band (1024 nodes): against the oracle evaluated live on the same seeded inputs (every node)
    and against the unit-fastest kernel on every entry (two compilations of one program: last-bit differences only);
    slots of nodes past the end stay padding (the operand is not read there)."""
    x, u, w, p = O.synthetic_inputs("anymal", count, seed=100 + count)
    m = ua.NodeModel("anymal")
    f, J = m.evaluate_numpy(x, u, w, p, mode="dense", layout="tiles")
    f0, J0 = m.evaluate_numpy(x, u, w, p, mode="dense", layout="soa")
    assert np.array_equal(f, f0) or np.abs(f - f0).max() <= 1e-13 * max(1.0, np.abs(f0).max())
    assert np.abs(J - J0).max() <= 1e-12 * np.abs(J0).max()
    assert np.array_equal(J == 0.0, J0 == 0.0), "structural zeros are exact zeros in both layouts"
    rf, rJ = O.node_jacobian_batched("anymal", x, u, w, p)
    _assert_close(f, J, rf, rJ)


@pytest.mark.gpu
def test_tile_kernel_with_knots_and_strided_operands(ua):
    """Trajectory operands (instance stride, knot stride) in, tiles out: node i = instance * knots + knot; the gather writes node-major blocks with a
    leading dimension and unit-fastest blocks, both equal to the host reading."""
    import torch
    batch, knots = 37, 5
    count = batch * knots
    x, u, w, p = O.synthetic_inputs("anymal", count, seed=7)
    m = ua.NodeModel("anymal")
    dev = torch.device("cuda", 0)
    ld = 64  # [x | u | pad] per node, node-major, as one VariableMap-style buffer per instance
    buf = torch.zeros((batch, knots, ld), dtype=torch.float64, device=dev)
    buf[:, :, :37] = torch.as_tensor(x).reshape(batch, knots, 37).to(dev)
    buf[:, :, 37:49] = torch.as_tensor(u).reshape(batch, knots, 12).to(dev)
    pt = torch.as_tensor(p).to(dev)
    X = ua.Operand(buf, instance_stride=knots * ld, knot_stride=ld, element_stride=1)
    U = ua.Operand(buf[:, :, 37:], instance_stride=knots * ld, knot_stride=ld, element_stride=1)
    f = torch.full((count, 37), float("nan"), dtype=torch.float64, device=dev)
    tiles = torch.full((m.tile_doubles(count),), float("nan"), dtype=torch.float64, device=dev)
    m.dense_jacobian_tiles(count, X, U, None, ua.Operand.per_instance(pt, m.np, shared=True), ua.Operand.aos(f, 37, knots), tiles, knots=knots)
    torch.cuda.synchronize()
    J = m.untile_numpy(tiles.cpu().numpy(), count)
    f0, J0 = m.evaluate_numpy(x, u, w, p, mode="dense", layout="soa")
    assert np.abs(f.cpu().numpy() - f0).max() <= 1e-13 * max(1.0, np.abs(f0).max())
    assert np.abs(J - J0).max() <= 1e-12 * np.abs(J0).max()
    # gather: node-major with a leading dimension, and unit-fastest
    ldj = 37 * 49 + 11
    aos = torch.full((count, ldj), float("nan"), dtype=torch.float64, device=dev)
    m.tiles_gather(count, tiles, ua.Operand.aos(aos, 37 * 49, knots, ld=ldj), knots=knots)
    soa = torch.full((37 * 49, count), float("nan"), dtype=torch.float64, device=dev)
    m.tiles_gather(count, tiles, ua.Operand.soa(soa, count, knots), knots=knots)
    torch.cuda.synchronize()
    assert np.array_equal(aos.cpu().numpy()[:, : 37 * 49].reshape(count, 37, 49), J)
    assert np.isnan(aos.cpu().numpy()[:, 37 * 49:]).all(), "the gather writes the block only"
    assert np.array_equal(soa.cpu().numpy().T.reshape(count, 37, 49), J)


def _device_inputs(count, seed):
    import torch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    x = torch.rand((37, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    u = (torch.rand((12, count), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1) * 20.0
    x[3:7] = x[3:7] / x[3:7].norm(dim=0, keepdim=True)
    return x, u, torch.as_tensor(O.default_params("anymal"), device="cuda")


@pytest.mark.gpu
def test_full_size_tile_launch_every_node_and_a_spread_sample_against_the_independent_oracle(ua):
    """BASELINE config 4 (4096 instances x N = 20 = 81 920 nodes, what bench.py times): the tile operand gathered on the device, then
    (1) EVERY node against the oracle's generated C (tests/c_checker.py: launch geometry, lane program, store path, gather);
    (2) 512 nodes SPREAD over the launch -- first and last wavefronts, both sides of every kind of boundary (tile, band, wavefront generation), random ones --
        against the INDEPENDENT torch oracle (oracle.ungar_oracle.node_jacobian), read through the host table, not the gather."""
    import torch
    from c_checker import compare_launch_with_c_checker
    batch, knots = 4096, 20
    count = batch * knots
    m = ua.NodeModel("anymal")
    x, u, p = _device_inputs(count, seed=23)
    P = ua.Operand.per_instance(p, m.np, shared=True)
    f = torch.empty((37, count), dtype=torch.float64, device="cuda")
    tiles = torch.full((m.tile_doubles(count),), float("nan"), dtype=torch.float64, device="cuda")
    m.dense_jacobian_tiles(count, ua.Operand.soa(x, count, knots), ua.Operand.soa(u, count, knots), None, P, ua.Operand.soa(f, count, knots), tiles, knots=knots)
    J = torch.full((37 * 49, count), float("nan"), dtype=torch.float64, device="cuda")
    m.tiles_gather(count, tiles, ua.Operand.soa(J, count, knots), knots=knots)
    torch.cuda.synchronize()
    assert not torch.isnan(J).any(), "every entry of every node is written"
    compare_launch_with_c_checker("anymal", x, u, p, f, J, count)
    # (2) spread sample against the torch oracle, host reading of the tile operand
    rng = np.random.default_rng(5)
    edges = [0, 1, 15, 16, 17, 1023, 1024, 1025, 16383, 16384, 16385, count - 17, count - 16, count - 1]
    sample = np.unique(np.concatenate([np.array(edges), np.arange(64), count - 1 - np.arange(64), rng.integers(0, count, 512)]))[:512]
    xs, us = x[:, sample].t().contiguous().cpu().numpy(), u[:, sample].t().contiguous().cpu().numpy()
    rf, rJ = O.node_jacobian_batched("anymal", xs, us, np.zeros((len(sample), 0)), np.tile(p.cpu().numpy(), (len(sample), 1)))
    Jh = m.untile_numpy(tiles.cpu().numpy(), count)[sample]
    _assert_close(f[:, sample].t().cpu().numpy(), Jh, rf, rJ)


@pytest.mark.gpu
def test_tile_kernel_rejects_what_it_cannot_do(ua):
    import torch
    m = ua.NodeModel("anymal")
    t = torch.zeros((64,), dtype=torch.float64, device="cuda")
    op = ua.Operand.soa(torch.zeros((49, 16), dtype=torch.float64, device="cuda"), 16)
    with pytest.raises(ua.UngarError, match="doubles for"):
        m.dense_jacobian_tiles(16, op, op, None, op, op, t)  # operand too small
    q = ua.NodeModel("quadrotor")
    with pytest.raises(ua.UngarError, match="no wave-tile program"):
        q.tile_doubles(16)
