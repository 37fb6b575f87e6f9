"""The lane-per-leg SPMD program (ungar_amd/csrc/codegen/quad_leg_program.hpp) on the CPU: the generated
body is generic over the value type, so the very text that is compiled for gfx950 runs here in a
4-lane simulator (tests/cpp/quad_sim.cpp: T = one value per lane of a quad, quad_sum / quad_rot as
loops) and is checked against the oracle's golden vectors -- block-arrow factorisation, row/column
ownership and rotations are validated without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def sim(repo_root, tmp_path_factory):
    gen = os.path.join(repo_root, "ungar_amd", "csrc", "gen", "anymal_quad_gen.hpp")
    if not os.path.exists(gen):
        pytest.skip("generated quad program missing: run __graft_entry__.build()")
    lib = os.path.join(repo_root, "build", "libquad_sim.so")
    src = os.path.join(repo_root, "tests", "cpp", "quad_sim.cpp")
    tiles = os.path.join(os.path.dirname(gen), "anymal_tiles_gen.hpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(gen), os.path.getmtime(tiles), os.path.getmtime(src)):
        lib = str(tmp_path_factory.mktemp("quad") / "libquad_sim.so")
        subprocess.run(["g++", "-std=c++20", "-O0", "-shared", "-fPIC", "-I", os.path.dirname(gen), "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


def test_quad_program_matches_golden(repo_root, sim):
    g = np.load(f"{repo_root}/tests/golden/node_anymal.npz")
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(g["x"].shape[0]):
        x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
        f, J = np.zeros(37), np.zeros((37, 49))
        sim.anymal_quad_sim(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f.ctypes.data_as(dp), J.ctypes.data_as(dp))
        assert not np.isnan(f).any() and not np.isnan(J).any(), "every entry of f and of the dense block must be written by some lane"
        assert np.abs(f - g["f"][b]).max() <= 1e-11 * max(1.0, np.abs(g["f"][b]).max())
        assert np.abs(J - g["J"][b]).max() <= 1e-10 * np.abs(g["J"][b]).max()


def test_tile_program_writes_every_entry_once_and_the_same_bits_as_the_unit_fastest_program(repo_root, sim):
    """The tile program (register images + slot table, phases in another order, literal columns pooled) in the 4-lane simulator: every entry of the dense block
    written exactly once (the simulator aborts on a second write), equal to the golden vectors, and BIT-IDENTICAL to the unit-fastest program -- the simulator is
    compiled without contraction, and the two programs are the same expressions: neither the phase order nor the store code changes a rounding there.  (On the GPU
    the compiler's contraction may pick fma(x, y, z w) in one kernel and fma(z, w, x y) in another: the kernels agree to the last bit or two, tests/test_tiles.py.)"""
    g = np.load(f"{repo_root}/tests/golden/node_anymal.npz")
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(g["x"].shape[0]):
        x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
        f, J, f2, J2 = np.zeros(37), np.zeros((37, 49)), np.zeros(37), np.zeros((37, 49))
        sim.anymal_quad_sim_tiles(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f.ctypes.data_as(dp), J.ctypes.data_as(dp))
        sim.anymal_quad_sim(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f2.ctypes.data_as(dp), J2.ctypes.data_as(dp))
        assert not np.isnan(f).any() and not np.isnan(J).any()
        assert np.abs(J - g["J"][b]).max() <= 1e-10 * np.abs(g["J"][b]).max()
        assert np.array_equal(f, f2) and np.array_equal(J, J2)


def test_value_only_quad_program_matches_golden_and_the_full_program(repo_root, sim):
    """forward_zero of the 'anymal' model runs the value sinks of the same recording (2.2 k statements per lane): every value written,
    equal to the golden vectors and bit-identical to the values the value + Jacobian program produces."""
    g = np.load(f"{repo_root}/tests/golden/node_anymal.npz")
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(g["x"].shape[0]):
        x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
        f, f2, J = np.zeros(37), np.zeros(37), np.zeros((37, 49))
        sim.anymal_quad_sim_value(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f.ctypes.data_as(dp))
        sim.anymal_quad_sim(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f2.ctypes.data_as(dp), J.ctypes.data_as(dp))
        assert not np.isnan(f).any()
        assert np.abs(f - g["f"][b]).max() <= 1e-11 * max(1.0, np.abs(g["f"][b]).max())
        assert np.array_equal(f, f2)


def test_quad_program_sparse_addressing(repo_root, sim):
    """Every Jacobian sink also carries its CSR index per lane; the values collected that way must be the
    dense block gathered through the committed sparsity pattern, with every pattern entry written."""
    import ungar_amd
    g = np.load(f"{repo_root}/tests/golden/node_anymal.npz")
    rows, cols = (a.astype(int) for a in ungar_amd.NodeModel("anymal").jacobian_sparsity())  # host-side table of the C ABI (no GPU needed)
    nnz = cols.size
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(min(4, g["x"].shape[0])):
        x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
        f, J, Js = np.zeros(37), np.zeros((37, 49)), np.zeros(nnz)
        sim.anymal_quad_sim_sparse(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f.ctypes.data_as(dp), J.ctypes.data_as(dp),
                                   Js.ctypes.data_as(dp), ctypes.c_int(nnz))
        assert not np.isnan(Js).any(), "a pattern entry was never written"
        assert np.array_equal(Js, J[rows, cols])
        mask = np.zeros((37, 49), dtype=bool)
        mask[rows, cols] = True
        assert np.all(J[~mask] == 0.0), "entries outside the pattern must be structural zeros"


@pytest.fixture(scope="module")
def rnea_sim(repo_root, tmp_path_factory):
    gen = os.path.join(repo_root, "ungar_amd", "csrc", "gen", "anymal_rnea_quad_gen.hpp")
    if not os.path.exists(gen):
        pytest.skip("generated joint-torque quad program missing: run __graft_entry__.build()")
    src = os.path.join(repo_root, "tests", "cpp", "quad_rnea_sim.cpp")
    lib = str(tmp_path_factory.mktemp("quad_rnea") / "libquad_rnea_sim.so")
    subprocess.run(["g++", "-std=c++20", "-O0", "-shared", "-fPIC", "-I", os.path.dirname(gen), "-o", lib, src], check=True)
    return ctypes.CDLL(lib)


def test_joint_torque_quad_program_matches_golden(repo_root, rnea_sim):
    """The lane-per-leg program of tau = RNEA(q, v, a) and d tau / d (q, v, a) (csrc/codegen/quad_rnea_program.hpp) in the 4-lane simulator:
    every value and every entry of the dense 18 x 55 block written, equal to the oracle's fixture; the CSR values collected through the per-lane
    indices of the sinks are the dense block gathered through the model's pattern, and everything outside the pattern is an exact zero."""
    import ungar_amd
    g = np.load(f"{repo_root}/tests/golden/rbd_anymal_rnea.npz")
    rows, cols = (a.astype(int) for a in ungar_amd.NodeModel("anymal_rnea").jacobian_sparsity())  # host-side table of the C ABI (no GPU needed)
    nnz = cols.size
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(g["x"].shape[0]):
        x, u = (np.ascontiguousarray(g[k][b]) for k in ("x", "u"))
        y, J, Js = np.zeros(18), np.zeros((18, 55)), np.zeros(nnz)
        rnea_sim.anymal_rnea_quad_sim(x.ctypes.data_as(dp), u.ctypes.data_as(dp), y.ctypes.data_as(dp), J.ctypes.data_as(dp), Js.ctypes.data_as(dp), ctypes.c_int(nnz))
        assert not np.isnan(y).any() and not np.isnan(J).any(), "every value and every entry of the dense block must be written by some lane"
        assert np.abs(y - g["y"][b]).max() <= 1e-12 * max(1.0, np.abs(g["y"][b]).max())
        assert np.abs(J - g["J"][b]).max() <= 1e-12 * np.abs(g["J"][b]).max()
        assert not np.isnan(Js).any(), "a pattern entry was never written"
        assert np.array_equal(Js, J[rows, cols])
        mask = np.zeros((18, 55), dtype=bool)
        mask[rows, cols] = True
        assert np.all(J[~mask] == 0.0)


def test_inertia_matrix_quad_program_matches_golden(repo_root, tmp_path_factory):
    """The lane-per-leg program of M(q) and d M / d q (csrc/codegen/quad_crba_program.hpp) in the 4-lane simulator: all 324 entries of M written
    (zeros between different legs included) and equal to the oracle's fixture; the CSR values of d M / d q, scattered through the per-leg indices of
    the sinks, cover the model's pattern completely and equal the fixture's dense block gathered through it."""
    import ungar_amd
    gen = os.path.join(repo_root, "ungar_amd", "csrc", "gen", "anymal_crba_quad_gen.hpp")
    if not os.path.exists(gen):
        pytest.skip("generated inertia-matrix quad program missing: run __graft_entry__.build()")
    lib = str(tmp_path_factory.mktemp("quad_crba") / "libquad_crba_sim.so")
    subprocess.run(["g++", "-std=c++20", "-O0", "-shared", "-fPIC", "-I", os.path.dirname(gen), "-o", lib, os.path.join(repo_root, "tests", "cpp", "quad_crba_sim.cpp")], check=True)
    sim = ctypes.CDLL(lib)
    g = np.load(f"{repo_root}/tests/golden/rbd_anymal_crba.npz")
    rows, cols = (a.astype(int) for a in ungar_amd.NodeModel("anymal_crba").jacobian_sparsity())
    nnz = cols.size
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(g["x"].shape[0]):
        x = np.ascontiguousarray(g["x"][b])
        y, Js = np.zeros(324), np.zeros(nnz)
        sim.anymal_crba_quad_sim(x.ctypes.data_as(dp), y.ctypes.data_as(dp), Js.ctypes.data_as(dp), ctypes.c_int(nnz))
        assert not np.isnan(y).any() and not np.isnan(Js).any(), "an entry of M or of the pattern was never written"
        J = g["J"][b].reshape(324, 19)
        assert np.abs(y - g["y"][b]).max() <= 1e-12 * np.abs(g["y"][b]).max()
        assert np.abs(Js - J[rows, cols]).max() <= 1e-12 * np.abs(J).max()


def test_centroidal_momentum_quad_program_matches_golden(repo_root, tmp_path_factory):
    """The lane-per-leg program of h_G and d h_G / d (q, v) (csrc/codegen/quad_centroidal_program.hpp) in the 4-lane simulator (the skeleton of the
    joint-torque program, 6 x 37): values, the dense block -- quaternion columns included, which are derivatives OFF the unit sphere and pin the
    cofactor form of the frame change -- and the CSR values through the per-leg indices, against the oracle's fixture."""
    import ungar_amd
    gen = os.path.join(repo_root, "ungar_amd", "csrc", "gen", "anymal_centroidal_quad_gen.hpp")
    if not os.path.exists(gen):
        pytest.skip("generated centroidal quad program missing: run __graft_entry__.build()")
    lib = str(tmp_path_factory.mktemp("quad_centroidal") / "libquad_centroidal_sim.so")
    subprocess.run(["g++", "-std=c++20", "-O0", "-shared", "-fPIC", "-DQUAD_SIM_CENTROIDAL", "-I", os.path.dirname(gen), "-o", lib,
                    os.path.join(repo_root, "tests", "cpp", "quad_rnea_sim.cpp")], check=True)
    sim = ctypes.CDLL(lib)
    g = np.load(f"{repo_root}/tests/golden/rbd_anymal_centroidal.npz")
    rows, cols = (a.astype(int) for a in ungar_amd.NodeModel("anymal_centroidal").jacobian_sparsity())
    nnz = cols.size
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(g["x"].shape[0]):
        x, u = np.ascontiguousarray(g["x"][b]), np.zeros(1)
        y, J, Js = np.zeros(6), np.zeros((6, 37)), np.zeros(nnz)
        sim.anymal_rnea_quad_sim(x.ctypes.data_as(dp), u.ctypes.data_as(dp), y.ctypes.data_as(dp), J.ctypes.data_as(dp), Js.ctypes.data_as(dp), ctypes.c_int(nnz))
        assert not np.isnan(y).any() and not np.isnan(J).any() and not np.isnan(Js).any()
        Jg = g["J"][b].reshape(6, 37)
        assert np.abs(y - g["y"][b]).max() <= 1e-12 * np.abs(g["y"][b]).max()
        assert np.abs(J - Jg).max() <= 1e-12 * np.abs(Jg).max()
        assert np.array_equal(Js, J[rows, cols])


def test_split_quad_program_matches_golden(repo_root, tmp_path_factory):
    """The lane-per-leg program SPLIT into a producer and a consumer half (quad_leg_program.hpp: QuadRole; the two wavefronts of a workgroup on
    the GPU, quad_split_kernel.hpp): here the two halves run as two threads that hand their 13 messages and the solved accelerations over
    through the same ring / counter protocol.  Every value and every entry of the dense block written by the consumer, equal to the golden
    vectors; an item read before it was sent or a ring slot overwritten too early would show up as a wrong entry, a missing post as a hang."""
    gen = os.path.join(repo_root, "ungar_amd", "csrc", "gen", "anymal_split_gen.hpp")
    if not os.path.exists(gen):
        pytest.skip("generated split program missing: run __graft_entry__.build()")
    lib = str(tmp_path_factory.mktemp("quad_split") / "libquad_split_sim.so")
    subprocess.run(["g++", "-std=c++20", "-O0", "-shared", "-fPIC", "-pthread", "-I", os.path.dirname(gen), "-o", lib,
                    os.path.join(repo_root, "tests", "cpp", "quad_split_sim.cpp")], check=True)
    sim = ctypes.CDLL(lib)
    g = np.load(f"{repo_root}/tests/golden/node_anymal.npz")
    dp = ctypes.POINTER(ctypes.c_double)
    for rep in range(3):  # the interleaving of the two threads differs from run to run
        for b in range(g["x"].shape[0]):
            x, u, p = (np.ascontiguousarray(g[k][b]) for k in ("x", "u", "p"))
            f, J = np.zeros(37), np.zeros((37, 49))
            sim.anymal_split_sim(x.ctypes.data_as(dp), u.ctypes.data_as(dp), p.ctypes.data_as(dp), f.ctypes.data_as(dp), J.ctypes.data_as(dp))
            assert not np.isnan(f).any() and not np.isnan(J).any(), "every entry of f and of the dense block must be written by the consumer"
            assert np.abs(f - g["f"][b]).max() <= 1e-11 * max(1.0, np.abs(g["f"][b]).max())
            assert np.abs(J - g["J"][b]).max() <= 1e-10 * np.abs(g["J"][b]).max()


def test_split_program_reads_every_message_inside_its_window(repo_root):
    """Static check of the generated consumer: every io.recv(m, i) sits between io.wait(m) and io.done(m) -- a read outside would see a ring
    slot that the producer may already have reused -- and the producer reads the returned accelerations only after io.wait_acc()."""
    import re
    gen = os.path.join(repo_root, "ungar_amd", "csrc", "gen", "anymal_split_gen.hpp")
    if not os.path.exists(gen):
        pytest.skip("generated split program missing: run __graft_entry__.build()")
    text = open(gen).read()
    producer, consumer = text.split("inline void ConsumerQuad", 1)
    current, reads = None, 0
    for line in consumer.splitlines():
        m = re.search(r"io\.wait\((\d+)\)", line)
        if m:
            current = int(m.group(1))
        if re.search(r"io\.done\((\d+)\)", line):
            current = None
        for m in re.finditer(r"io\.recv\((\d+), (\d+)\)", line):
            reads += 1
            assert int(m.group(1)) == current, line
    assert reads == 13 * 9
    waited = False
    for line in producer.splitlines():
        waited = waited or "io.wait_acc()" in line
        assert waited or "io.acc(" not in line, line
    assert waited


@pytest.mark.gpu
def test_split_kernel_on_the_device_equals_the_fused_kernel(repo_root):
    """The producer / consumer program at TWO wavefronts per SIMD (not the product's route -- it is slower than the fused kernel, whose result
    stores bound both) on the device: every value and every entry of the 81 920-node launch (BASELINE config 4: 148 520 960 Jacobian entries) against the fused kernel,
    for the split kernel with 8- and with paired 16-byte stores and for the fused kernel with paired stores; no entry left unwritten; differences at rounding level
    (the two programs contract some sums in a different order)."""
    import re
    exe = os.path.join(repo_root, "build", "variants", "quad_split_bench")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, "check", "81920"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:], r.stderr[-1000:])
    assert r.returncode == 0
    lines = re.findall(r"check (.+?)\s+vs reference: max \|dJ\| (\S+) \(max \|J\| (\S+)\), max \|df\| (\S+), unwritten (\d+), entries that differ (\d+) of (\d+)", r.stdout)
    assert len(lines) >= 3 and any(name.startswith("split") for name, *_ in lines), r.stdout[-2000:]
    for name, dj, scale, df, unwritten, _, entries in lines:
        assert int(unwritten) == 0 and int(entries) == 1813 * 81920, name
        assert float(dj) <= 1e-11 * float(scale) and float(df) <= 1e-11, (name, dj, scale, df)
