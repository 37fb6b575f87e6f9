"""Batched soft SQP from C++ on the reference's OCPs AS WRITTEN.

`build/batched_{quadrotor,rc_car,quadruped}_test` (tests/cpp/batched_*_test.cpp) are plain C++20 host programs over
the C ABI.  Each states one of the reference's THREE MPC problems twice through the facade -- whole-horizon, as
example/mpc/quadrotor.example.cpp:196-291 / rc_car.example.cpp:191-285 / quadruped.example.cpp:209-338 do (523 / 246 / 1123 decision
variables, input-rate coupling, terminal tracking, 480 foot-contact equality rows), and in stage form with the cross-knot quantity carried in the stage state -- and advances
  * >= 1024 perturbed instances (random references, measured states, rotor bounds active or not; random gaits for the quadruped) with
    Ungar::BatchedSoftSQPOptimizer: stage derivatives, QP assembly, Riccati recursion with stage equality rows, stacked backtracking
    search, all on the device;
  * a sample of them with the facade's Ungar::SoftSQPOptimizer: sparse L D L^T of the whole KKT system of the QP the reference hands
    to OSQP (soft_sqp.hpp:143-158) and the host line search (backtracking_line_search.hpp:116-151).
The search direction, the accepted step size and the iterate must agree to 1e-9 after the first AND after the second iteration."""
import os
import re
import subprocess

import numpy as np

import pytest
from conftest import measurement_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("problem,batch,compared", [("quadrotor", 1024, 8), ("rc_car", 1024, 8), ("quadruped", 1024, 8)])
def test_batched_sqp_equals_the_whole_horizon_facade(repo_root, shared_codegen, problem, batch, compared):
    exe = os.path.join(repo_root, "build", f"batched_{problem}_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(shared_codegen(f"batched_{problem}")), str(batch), str(compared)], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and f"PASS batched {problem} SQP (batch {batch}, {compared} compared)" in r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("iteration") and "instances accepted" in l]
    assert len(lines) == 2
    for line in lines:
        moved, total = map(int, re.search(r"(\d+) of (\d+) instances accepted", line).groups())
        assert total == batch and moved >= batch // 2  # the comparison is not vacuous: steps are taken


@pytest.mark.parametrize("problem", ["quadrotor", "rc_car"])
def test_workgroup_assembly_route_of_the_problems_without_equality_rows(repo_root, shared_codegen, problem):
    """Stage problems without equality rows whose row fits a wavefront take ShootingAssembleSmallKernel (one wavefront per node) by default -- that is what
    test_batched_sqp_equals_the_whole_horizon_facade runs; UNGAR_AMD_ASSEMBLE_VARIANT=workgroup keeps them on the workgroup kernel's generic sections (what larger rows
    take).  Same comparison with the facade's whole-horizon optimiser on that route."""
    exe = os.path.join(repo_root, "build", f"batched_{problem}_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(shared_codegen(f"batched_{problem}")), "512", "4"], capture_output=True, text=True, timeout=1500, env=measurement_env({"UNGAR_AMD_ASSEMBLE_VARIANT": "workgroup"}))
    print(r.stdout[-3000:], r.stderr[-1000:])
    assert r.returncode == 0 and f"PASS batched {problem} SQP (batch 512, 4 compared)" in r.stdout


@pytest.mark.parametrize("first_stage", ["0", "1", "5"])
def test_line_search_stages_give_the_same_iterates(repo_root, shared_codegen, first_stage):
    """The candidate steps are offered in stages -- the first few to every instance, the rest to the LISTED instances that accepted none (default: 2 first) -- and the
    stages must not change which step an instance takes: all candidates at once (0, no read-back), one first, five first; same comparison with the facade's
    optimiser (step sizes equal to the last bit) as the default schedule in test_batched_sqp_equals_the_whole_horizon_facade."""
    exe = os.path.join(repo_root, "build", "batched_quadrotor_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(shared_codegen("batched_quadrotor")), "512", "6"], capture_output=True, text=True, timeout=1500, env={**os.environ, "UNGAR_TEST_FIRST_STAGE": first_stage})
    print(r.stdout[-3000:], r.stderr[-1000:])
    assert r.returncode == 0 and "PASS batched quadrotor SQP (batch 512, 6 compared)" in r.stdout


def test_assembly_kernels_agree(repo_root, tmp_path, shared_codegen):
    """Three routes through the shooting assembly: the one-wavefront kernel (quadruped-shaped stage nodes: tiles of W and [A|B] in registers, the linear
    terms as the homogeneous column of the matrix-core products), the workgroup kernel with its wavefront-specialised sections
    (UNGAR_AMD_ASSEMBLE_VARIANT=workgroup), and the workgroup kernel's generic sections (UNGAR_AMD_ASSEMBLE_GENERIC=1).  Same pivot
    rule and the same arithmetic in the two workgroup routes: the assembled QP data (AB, b, W, w, reduced equality rows and residuals, printed with
    17 digits) and both steps of the compared instances agree to the last bit over two SQP iterations.  The one-wavefront kernel sums w' and b' on
    the matrix cores in another order: same pivots, data and steps within 1e-9 of each block's largest entry."""
    exe = os.path.join(repo_root, "build", "batched_quadruped_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    dumps = {}
    for mode in ("wavefront", "workgroup", "generic"):
        env = measurement_env()
        env.pop("UNGAR_AMD_ASSEMBLE_GENERIC", None)
        env.pop("UNGAR_AMD_ASSEMBLE_VARIANT", None)
        if mode == "generic":
            env["UNGAR_AMD_ASSEMBLE_GENERIC"] = "1"
        if mode == "workgroup":
            env["UNGAR_AMD_ASSEMBLE_VARIANT"] = "workgroup"
        folder = tmp_path / mode
        folder.mkdir()
        r = subprocess.run([exe, str(shared_codegen("batched_quadruped")), "256", "3", str(folder)], capture_output=True, text=True, timeout=1500, env=env)
        print(r.stdout[-2000:], r.stderr[-1000:])
        assert r.returncode == 0 and "PASS batched quadruped SQP (batch 256, 3 compared)" in r.stdout
        dumps[mode] = {f.name: f.read_bytes() for f in sorted(folder.iterdir())}
    assert len(dumps["workgroup"]) >= 6 and dumps["workgroup"].keys() == dumps["generic"].keys() == dumps["wavefront"].keys()  # 3 instances x 2 iterations
    for name, data in dumps["workgroup"].items():
        assert len(data) > 100_000 and data == dumps["generic"][name], f"{name}: the two workgroup code paths disagree"

    def blocks(data):
        lines = data.decode().split("\n")
        N, nz, nu, ne, nc = map(int, lines[0].split())
        values = np.array([float(x) for x in lines[1:] if x.strip()])
        nd = nz + nu
        sizes = [("AB", N * nz * nd), ("b", N * nz), ("W", (N + 1) * nd * nd), ("w", (N + 1) * nd), ("E", N * ne * nd), ("e", (N + 1) * ne), ("dz0", nz), ("dZ", (N + 1) * nz), ("dU", N * nu)]
        out, at = {}, 0
        for key, n in sizes:
            out[key] = values[at:at + n]
            at += n
        out["d_facade"] = values[at:]
        return out

    differs = 0
    for name, data in dumps["wavefront"].items():
        mine, theirs = blocks(data), blocks(dumps["workgroup"][name])
        for key, block in theirs.items():
            scale = max(float(np.max(np.abs(block))), 1e-300)
            worst = float(np.max(np.abs(mine[key] - block))) / scale
            assert worst <= 1e-9, f"{name} {key}: {worst:.3e}"
            differs += worst > 0.0
        # the pivot structure is the same: the reduced rows have their exact 0 / 1 entries in the same places
        assert np.array_equal(mine["E"] == 1.0, theirs["E"] == 1.0) and np.array_equal(mine["E"] == 0.0, theirs["E"] == 0.0), name
    assert differs > 0  # (the one-wavefront kernel did run: bitwise equality everywhere would mean the workgroup kernel was compared with itself)


def test_line_search_candidates_in_groups_give_the_same_iterates(repo_root, shared_codegen):
    """More candidate steps than one stacked evaluation holds (16; the reference accepts any BacktrackingLineSearch parameters,
    backtracking_line_search.hpp:56-78) are offered in groups, largest first, an instance taking the first acceptable candidate over all groups.  Pinned
    with the default 14 candidates in groups of 4 (UNGAR_AMD_STACKED_CANDIDATES): step sizes and iterates must still equal the facade's."""
    exe = os.path.join(repo_root, "build", "batched_rc_car_test")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(shared_codegen("batched_rc_car")), "512", "8"], capture_output=True, text=True, timeout=1500, env={**os.environ, "UNGAR_AMD_STACKED_CANDIDATES": "4"})
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0 and "PASS batched rc_car SQP (batch 512, 8 compared)" in r.stdout


@pytest.mark.parametrize("sizes", [(10, 3, 0, ""), (20, 9, 4, ""), (10, 3, 2, "_const")], ids=lambda s: "x".join(map(str, s)))
def test_user_problems_of_sizes_the_library_was_not_compiled_for(repo_root, shared_codegen, sizes):
    """The reference's optimiser takes ANY problem (optimization/concepts.hpp:153-262).  Two user OCPs whose stage sizes have no prebuilt solver kernels
    (10 + 3; 20 + 9 with 4 stage equality rows) through the batched driver: the program asserts that the register-resident Riccati recursion and the
    one-wavefront assembly INSTANTIATED FOR THE SIZES by the kernel factory ran (RiccatiRoute() == 2; AssembleRoute() == 2, or 3 without equality
    rows), and compares search direction, step size and iterate with the facade's sparse KKT solve of the whole-horizon statement over two iterations
    (<= 1e-9)."""
    nx, nu, ne, variant = sizes  # variant "_const": the same problem with constant data -- no knot / instance parameters, the rows are [x | u] alone
    exe = os.path.join(repo_root, "build", f"batched_user_ocp_test_{nx}_{nu}_{ne}{variant}")
    assert os.path.exists(exe), f"{exe} missing: run __graft_entry__.build()"
    r = subprocess.run([exe, str(shared_codegen(f"batched_user_{nx}_{nu}_{ne}{variant}")), "256", "4"], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0 and f"PASS batched user OCP {nx} + {nu}, {ne} equality rows" in r.stdout


# ---- redundant stage equality rows (ungar_shooting_assemble with eliminate_equalities) ------------------------------------------------------
import ctypes  # noqa: E402

import numpy as np  # noqa: E402


class _Pattern(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_void_p), ("cols", ctypes.c_void_p), ("nnz", ctypes.c_int64)]


class _Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in ("nx", "nu", "nc", "nw", "np", "horizon", "batch")] + [("carry_inputs", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class _BarrierC(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stiffness", ctypes.c_double), ("epsilon", ctypes.c_double)]


class _AssembleArgs(ctypes.Structure):
    _fields_ = ([("dims", _Dims)] + [(n, ctypes.c_void_p) for n in ("rows", "xm", "f", "f_jac", "carry_jac", "cost_grad", "cost_hes", "h", "h_jac", "eq_jac")] +
                [(n, _Pattern) for n in ("f_pattern", "carry_pattern", "cost_grad_pattern", "cost_hes_pattern", "h_pattern", "eq_pattern")] +
                [("nh", ctypes.c_int64), ("ne", ctypes.c_int64), ("barrier", _BarrierC), ("regularization", ctypes.c_double)] +
                [(n, ctypes.c_void_p) for n in ("AB", "b", "W", "w", "E", "dz0")] + [("eliminate_equalities", ctypes.c_int32), ("reserved", ctypes.c_int32)] +
                [(n, ctypes.c_void_p) for n in ("eq", "eq_reduced", "eq_pivots")])


@pytest.mark.parametrize("nx,nu", [(3, 5), (20, 14)])  # 64 lanes per node (generic sections) / 256 lanes (wavefront-specialised sections, and generic on request)
def test_redundant_equality_rows_take_no_pivot(nx, nu, monkeypatch, measurement_library):
    """Two identical stage equality rows (and a third that is a combination of the others): the duplicate reduces to rounding noise (~1e-16 of its entries, not
    exactly zero) and must be recognised against its ORIGINAL scale -- relative to its own reduced entries the noise would pass as a pivot and 1 / pivot would
    blow up W, [A|B] and w.  The reduced problem must equal the one assembled from the independent rows alone; a duplicate with a DIFFERENT residual cannot be
    met and is reported (-2).  Both code paths of the assembly kernel."""
    import torch
    import ungar_amd
    lib = ungar_amd.load_library()
    lib.ungar_shooting_assemble.argtypes = [ctypes.POINTER(_AssembleArgs), ctypes.c_void_p]
    N, B, nd = 2, 3, nx + nu
    rng = np.random.default_rng(7)
    dev = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")  # noqa: E731
    nodes = B * (N + 1)

    def dense_pattern(rows, cols):
        r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
        return dev(r.ravel(), torch.int32), dev(c.ravel(), torch.int32)

    def upper_pattern(n):
        r, c = np.triu_indices(n)
        return dev(r, torch.int32), dev(c, torch.int32)

    base = rng.standard_normal((nodes, 2, nd))  # two independent rows per node
    mix = rng.standard_normal((nodes, 2))
    resid = rng.standard_normal((nodes, 2))

    def run(rows_of, resid_of, generic):
        """rows_of(node) -> (ne, nd) equality Jacobian; returns (pivots, W, AB, w, b, E', e')."""
        ne = rows_of(0).shape[0]
        eJ = np.stack([rows_of(i) for i in range(nodes)])
        ev = np.stack([resid_of(i) for i in range(nodes)])
        rows = dev(rng.standard_normal((nodes, nd)))
        L = rng.standard_normal((nodes, nd, nd))
        H = L @ L.transpose(0, 2, 1) + nd * np.eye(nd)
        iu = np.triu_indices(nd)
        keep = {"pf": dense_pattern(nx, nd), "pg": dense_pattern(1, nd), "pH": upper_pattern(nd), "pe": dense_pattern(ne, nd)}
        t = {"rows": rows, "xm": dev(rng.standard_normal((B, nx))), "f": dev(rng.standard_normal((nodes, nx))), "fJ": dev(rng.standard_normal((nodes, nx * nd))),
             "lg": dev(rng.standard_normal((nodes, nd))), "lH": dev(H[:, iu[0], iu[1]]), "eJ": dev(eJ.reshape(nodes, -1)), "e": dev(ev),
             "AB": torch.zeros((B * N, nx * nd), dtype=torch.float64, device="cuda"), "b": torch.zeros((B * N, nx), dtype=torch.float64, device="cuda"),
             "W": torch.zeros((nodes, nd * nd), dtype=torch.float64, device="cuda"), "w": torch.zeros((nodes, nd), dtype=torch.float64, device="cuda"),
             "E": torch.zeros((B * N, ne * nd), dtype=torch.float64, device="cuda"), "dz0": torch.zeros((B, nx), dtype=torch.float64, device="cuda"),
             "er": torch.zeros((B * N, ne), dtype=torch.float64, device="cuda"), "piv": torch.full((B * N, ne), 99, dtype=torch.int32, device="cuda")}
        a = _AssembleArgs()
        a.dims = _Dims(nx, nu, 0, 0, 0, N, B, 0, 0)
        for name, key in (("rows", "rows"), ("xm", "xm"), ("f", "f"), ("f_jac", "fJ"), ("cost_grad", "lg"), ("cost_hes", "lH"), ("eq_jac", "eJ"), ("AB", "AB"), ("b", "b"), ("W", "W"),
                          ("w", "w"), ("E", "E"), ("dz0", "dz0"), ("eq", "e"), ("eq_reduced", "er"), ("eq_pivots", "piv")):
            setattr(a, name, t[key].data_ptr())
        for name, key in (("f_pattern", "pf"), ("cost_grad_pattern", "pg"), ("cost_hes_pattern", "pH"), ("eq_pattern", "pe")):
            r, c = keep[key]
            setattr(a, name, _Pattern(r.data_ptr(), c.data_ptr(), r.numel()))
        a.nh, a.ne, a.regularization, a.eliminate_equalities = 0, ne, 1e-6, 1
        a.barrier = _BarrierC(0, 0, 100.0, 2e-5)
        if generic:
            monkeypatch.setenv("UNGAR_AMD_ASSEMBLE_GENERIC", "1")
        else:
            monkeypatch.delenv("UNGAR_AMD_ASSEMBLE_GENERIC", raising=False)
        assert lib.ungar_shooting_assemble(ctypes.byref(a), None) == 0, lib.ungar_last_error()
        torch.cuda.synchronize()
        return {k: t[k].cpu().numpy() for k in ("piv", "W", "AB", "w", "b", "E", "er")}

    for generic in (False, True):
        rng = np.random.default_rng(11)  # the same random problem data for every call of run() below
        ref = run(lambda i: base[i], lambda i: resid[i], generic)
        assert (ref["piv"] >= 0).all()
        # rows: r0, r1, a copy of r0 (same residual) and a combination of r0 and r1 (combined residual): two redundant rows
        def with_copies(i):
            return np.stack([base[i][0], base[i][1], base[i][0], mix[i][0] * base[i][0] + mix[i][1] * base[i][1]])
        def residuals(i):
            return np.array([resid[i][0], resid[i][1], resid[i][0], mix[i][0] * resid[i][0] + mix[i][1] * resid[i][1]])
        rng = np.random.default_rng(11)
        out = run(with_copies, residuals, generic)
        assert (out["piv"][:, :2] == ref["piv"]).all() and (out["piv"][:, 2:] == -1).all(), out["piv"]
        for key in ("W", "AB", "w", "b"):
            assert np.isfinite(out[key]).all()
            scale = np.abs(ref[key]).max()
            assert np.abs(out[key] - ref[key]).max() <= 1e-10 * scale, key  # the redundant rows changed nothing
        # the duplicate with a DIFFERENT residual cannot be met by any input
        rng = np.random.default_rng(11)
        bad = run(with_copies, lambda i: residuals(i) + np.array([0.0, 0.0, 0.5, 0.0]), generic)
        assert (bad["piv"][:, 2] == -2).all() and (bad["piv"][:, 3] == -1).all() and (bad["piv"][:, :2] == ref["piv"]).all()


@pytest.mark.parametrize("nx,nu,nh,dense_row", [(13, 4, 8, False), (8, 2, 6, False), (6, 1, 3, False), (9, 3, 0, False), (20, 9, 5, True), (30, 12, 14, True), (10, 3, 0, False)])
def test_one_wavefront_assembly_of_problems_without_equality_rows_on_random_patterns(nx, nu, nh, dense_row, monkeypatch, measurement_library):
    """ShootingAssembleSmallKernel (stage problems without equality rows, nd + 1 <= 64) against the workgroup kernel (UNGAR_AMD_ASSEMBLE_VARIANT=workgroup) on random
    SPARSE patterns -- Hessian, gradient, dynamics Jacobian, inequality Jacobian with rows of one to several entries, and (dense_row) a row whose entry pairs do not
    fit a wavefront (the one-lane-per-entry fallback of the barrier terms) -- with carried inputs (nc = nu, identity carry).  Every block it writes (upper triangle of W,
    w, [A|B], b, dz0 for stage and terminal nodes) within 1e-13 of the other route's."""
    import torch
    import ungar_amd
    lib = ungar_amd.load_library()
    lib.ungar_shooting_assemble.argtypes = [ctypes.POINTER(_AssembleArgs), ctypes.c_void_p]
    N, B, nc = 2, 5, nu  # (15 nodes: the last wavefront of the several-nodes-per-wavefront kernel has a group without a node)
    nz, nd = nc + nx, nc + nx + nu
    rng = np.random.default_rng(5 + nx)
    dev = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")  # noqa: E731
    nodes = B * (N + 1)

    def random_pattern(rows, cols, density, upper=False, keep_diagonal=False):
        mask = rng.random((rows, cols)) < density
        if upper:
            mask = np.triu(mask)
        if keep_diagonal:
            mask |= np.eye(rows, cols, dtype=bool)
        r, c = np.nonzero(mask)  # (row-major order: canonical CSR)
        return r.astype(np.int32), c.astype(np.int32)

    pH = random_pattern(nd, nd, min(0.3, 180.0 / (nd * (nd + 1) / 2)), upper=True, keep_diagonal=True)
    pg = (np.zeros(nd, dtype=np.int32), np.arange(nd, dtype=np.int32))
    pf = random_pattern(nx, nx + nu, min(0.5, 200.0 / (nx * (nx + nu))), keep_diagonal=True)  # dynamics over [x | u]
    assert pH[0].size <= 256 and pf[0].size <= 256 and nd + 1 <= 64  # the bounds of the one-wavefront kernel (beyond them the launcher takes the workgroup kernel)
    hmask = rng.random((nh, nd)) < 0.08
    hmask[np.arange(nh), rng.integers(0, nd, nh)] = True  # no empty row
    if dense_row:
        hmask[nh // 2, :] = False
        hmask[nh // 2, rng.choice(nd, 12, replace=False)] = True  # 78 pairs in this row alone
    # the kernels take up to 64 inequality-Jacobian entries through their single-request path: keep the pattern below that bound
    while hmask.sum() > 64:
        r, c = np.nonzero(hmask)
        pick = rng.integers(0, r.size)
        if hmask[r[pick]].sum() > 1 and not (dense_row and r[pick] == nh // 2):
            hmask[r[pick], c[pick]] = False
    ph = tuple(a.astype(np.int32) for a in np.nonzero(hmask))
    pats = {k: (dev(v[0], torch.int32), dev(v[1], torch.int32)) for k, v in (("pH", pH), ("pg", pg), ("pf", pf), ("ph", ph))}
    t = {"rows": dev(rng.standard_normal((nodes, nd + 3))), "xm": dev(rng.standard_normal((B, nx))), "f": dev(rng.standard_normal((nodes, nx))),
         "fJ": dev(rng.standard_normal((nodes, pf[0].size))), "lg": dev(rng.standard_normal((nodes, nd))), "lH": dev(rng.standard_normal((nodes, pH[0].size))),
         "h": dev(-np.abs(rng.standard_normal((nodes, nh))) * 0.01), "hJ": dev(rng.standard_normal((nodes, ph[0].size)))}

    def run(workgroup, one_node_per_wavefront=False):
        out = {"AB": torch.full((B * N, nz * nd), 7.0, dtype=torch.float64, device="cuda"), "b": torch.full((B * N, nz), 7.0, dtype=torch.float64, device="cuda"),
               "W": torch.zeros((nodes, nd * nd), dtype=torch.float64, device="cuda"), "w": torch.full((nodes, nd), 7.0, dtype=torch.float64, device="cuda"),
               "dz0": torch.full((B, nz), 7.0, dtype=torch.float64, device="cuda")}
        a = _AssembleArgs()
        a.dims = _Dims(nx, nu, nc, 3, 0, N, B, 1, 0)  # (three knot parameters behind [c | x | u]: the row stride differs from nd)
        for name, key in (("rows", "rows"), ("xm", "xm"), ("f", "f"), ("f_jac", "fJ"), ("cost_grad", "lg"), ("cost_hes", "lH"), ("h", "h"), ("h_jac", "hJ")):
            setattr(a, name, t[key].data_ptr())
        for name in ("AB", "b", "W", "w", "dz0"):
            setattr(a, name, out[name].data_ptr())
        for name, key in (("f_pattern", "pf"), ("cost_grad_pattern", "pg"), ("cost_hes_pattern", "pH"), ("h_pattern", "ph")):
            r, c = pats[key]
            setattr(a, name, _Pattern(r.data_ptr(), c.data_ptr(), r.numel()))
        a.nh, a.ne, a.regularization, a.eliminate_equalities = nh, 0, 1e-6, 0
        a.barrier = _BarrierC(0, 0, 100.0, 2e-5)
        if workgroup:
            monkeypatch.setenv("UNGAR_AMD_ASSEMBLE_VARIANT", "workgroup")
        else:
            monkeypatch.delenv("UNGAR_AMD_ASSEMBLE_VARIANT", raising=False)
        if one_node_per_wavefront:
            monkeypatch.setenv("UNGAR_AMD_ASSEMBLE_ONE_NODE_PER_WAVEFRONT", "1")
        else:
            monkeypatch.delenv("UNGAR_AMD_ASSEMBLE_ONE_NODE_PER_WAVEFRONT", raising=False)
        assert lib.ungar_shooting_assemble(ctypes.byref(a), None) == 0, lib.ungar_last_error()
        torch.cuda.synchronize()
        res = {k: v.cpu().numpy() for k, v in out.items()}
        res["W"] = np.triu(res["W"].reshape(nodes, nd, nd))  # (only the upper triangle is written)
        return res

    mine, theirs = run(False), run(True)
    if nd + 1 <= 16:  # narrow rows: the default route packs four nodes into a wavefront -- same arithmetic in the same order as one node per wavefront
        single = run(False, one_node_per_wavefront=True)
        for key in ("W", "w", "AB", "b", "dz0"):
            assert np.array_equal(mine[key], single[key]), key
    assert np.abs(theirs["W"]).max() > 1.0 and np.abs(theirs["AB"]).max() > 0.1 and not (theirs["b"] == 7.0).any()
    for key in ("W", "w", "AB", "b", "dz0"):
        scale = np.abs(theirs[key]).max()
        assert np.abs(mine[key] - theirs[key]).max() <= 1e-13 * scale, key


# (stage sizes nx, nu, carried, equality rows, inequality rows; expected ungar_shooting_assemble_route): the reference's quadruped shape is compiled into the
# library (1); every other shape the kernel template fits is instantiated by the kernel factory on first use (2)
# the last shape: two states, 20 inputs, 16 rows -- the tableau [E | e] (16 x 23) is the largest of the images that share the kernel's one LDS region (packed W_e: 276)
WAVE_SHAPES = [(13, 24, 12, 16, 12, 1), (8, 9, 12, 4, 5, 2), (10, 3, 0, 2, 0, 2), (30, 12, 5, 7, 9, 2), (2, 20, 0, 16, 3, 2)]


@pytest.mark.parametrize("seed,shape", [(1, WAVE_SHAPES[0]), (2, WAVE_SHAPES[0]), (3, WAVE_SHAPES[0]), (4, WAVE_SHAPES[1]), (5, WAVE_SHAPES[2]), (6, WAVE_SHAPES[3]), (7, WAVE_SHAPES[4])])
def test_one_wavefront_assembly_with_equality_rows_on_random_patterns(seed, shape, monkeypatch, measurement_library):
    """ShootingAssembleWaveKernel<NZ, NU, NE> -- compiled in for quadruped-shaped stage nodes (12 carried + 13 states, 24 inputs, 16 equality rows eliminated per
    node), instantiated at run time by the kernel factory for the other shapes (20 + 9 with 4 rows, 10 + 3 with 2 rows and no inequality, 35 + 12 with 7 rows) -- against the
    workgroup kernel on random sparse patterns and values that the quadruped's own data never produce: equality rows with state and input entries in random places,
    empty rows, an empty row with a residual (cannot be met: -2), rows without input entries, a carry Jacobian, dense-ish inequality rows.  Same pivots; reduced
    rows, residuals, W and [A|B] to 1e-12 of their scale (the same matrix-core sequences); w and b to 1e-9 (summed in another order)."""
    import torch
    import ungar_amd
    lib = ungar_amd.load_library()
    lib.ungar_shooting_assemble.argtypes = [ctypes.POINTER(_AssembleArgs), ctypes.c_void_p]
    nx, nu, nc, ne, nh, route = shape
    nw, N, B = 2, 2, 6
    nz, nd = nc + nx, nc + nx + nu
    lib.ungar_shooting_assemble_route.argtypes = [ctypes.c_int64] * 4 + [ctypes.c_int32]
    assert lib.ungar_shooting_assemble_route(nz, nu, ne, nh, 1) == route, lib.ungar_last_error()
    rng = np.random.default_rng(100 + seed)
    dev = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")  # noqa: E731
    nodes = B * (N + 1)

    def pattern_of(mask):
        r, c = np.nonzero(mask)
        return r.astype(np.int32), c.astype(np.int32)

    pH = pattern_of(np.triu(rng.random((nd, nd)) < min(0.12, 150.0 / (nd * nd))) | np.eye(nd, dtype=bool))
    pg = (np.zeros(nd, dtype=np.int32), np.arange(nd, dtype=np.int32))
    pf = pattern_of((rng.random((nx, nx + nu)) < min(0.4, 180.0 / (nx * (nx + nu)))) | np.eye(nx, nx + nu, dtype=bool))  # (the one-wavefront kernels take patterns of up to 256 entries)
    pc = pattern_of(rng.random((nc, nx + nu)) < 0.2) if nc else (np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32))
    emask = np.zeros((ne, nd), dtype=bool)
    for r in range(ne):
        kind = rng.random()
        if kind < 0.25 and r > 0:
            continue  # empty row (an inactive contact)
        cols = rng.choice(nd, rng.integers(2, min(7, nd)), replace=False)
        emask[r, cols] = True
        if kind < 0.9:
            emask[r, nz + rng.integers(0, nu)] = True  # at least one input entry (else: a row the inputs cannot meet unless it is zero)
        else:
            emask[r, nz:] = False
    pe = pattern_of(emask)
    hmask = rng.random((nh, nd)) < 0.05
    if nh:
        hmask[np.arange(nh), rng.integers(0, nd, nh)] = True
    ph = pattern_of(hmask)
    assert pH[0].size <= 256 and pf[0].size <= 256 and pc[0].size <= 128 and pe[0].size <= 256 and ph[0].size <= 64
    pats = {k: (dev(v[0], torch.int32), dev(v[1], torch.int32)) for k, v in (("pH", pH), ("pg", pg), ("pf", pf), ("pc", pc), ("pe", pe), ("ph", ph))}
    L = rng.standard_normal((nodes, nd, nd)) * 0.3
    H = L @ L.transpose(0, 2, 1) + 3.0 * np.eye(nd)
    ev = rng.standard_normal((nodes, ne))
    empty = ~emask.any(axis=1)
    ev[:, empty] = 0.0
    if empty.any():
        ev[1, np.flatnonzero(empty)[0]] = 0.3  # node 1: an empty row with a residual
    t = {"rows": dev(rng.standard_normal((nodes, nd + nw))), "xm": dev(rng.standard_normal((B, nx))), "f": dev(rng.standard_normal((nodes, nx))),
         "fJ": dev(rng.standard_normal((nodes, pf[0].size))), "cJ": dev(rng.standard_normal((nodes, pc[0].size))), "lg": dev(rng.standard_normal((nodes, nd))),
         "lH": dev(H[:, pH[0], pH[1]]), "h": dev(-np.abs(rng.standard_normal((nodes, nh))) * 0.01), "hJ": dev(rng.standard_normal((nodes, ph[0].size))),
         "eJ": dev(rng.standard_normal((nodes, pe[0].size))), "e": dev(ev)}

    def run(workgroup):
        out = {"AB": torch.zeros((B * N, nz * nd), dtype=torch.float64, device="cuda"), "b": torch.zeros((B * N, nz), dtype=torch.float64, device="cuda"),
               "W": torch.zeros((nodes, nd * nd), dtype=torch.float64, device="cuda"), "w": torch.zeros((nodes, nd), dtype=torch.float64, device="cuda"),
               "E": torch.zeros((B * N, ne * nd), dtype=torch.float64, device="cuda"), "dz0": torch.zeros((B, nz), dtype=torch.float64, device="cuda"),
               "er": torch.zeros((B * N, ne), dtype=torch.float64, device="cuda"), "piv": torch.full((B * N, ne), 99, dtype=torch.int32, device="cuda")}
        a = _AssembleArgs()
        a.dims = _Dims(nx, nu, nc, nw, 0, N, B, 0, 0)
        for name, key in (("rows", "rows"), ("xm", "xm"), ("f", "f"), ("f_jac", "fJ"), ("carry_jac", "cJ"), ("cost_grad", "lg"), ("cost_hes", "lH"), ("h", "h"), ("h_jac", "hJ"),
                          ("eq_jac", "eJ"), ("eq", "e")):
            setattr(a, name, t[key].data_ptr())
        for name, key in (("AB", "AB"), ("b", "b"), ("W", "W"), ("w", "w"), ("E", "E"), ("dz0", "dz0"), ("eq_reduced", "er"), ("eq_pivots", "piv")):
            setattr(a, name, out[key].data_ptr())
        for name, key in (("f_pattern", "pf"), ("carry_pattern", "pc"), ("cost_grad_pattern", "pg"), ("cost_hes_pattern", "pH"), ("h_pattern", "ph"), ("eq_pattern", "pe")):
            r, c = pats[key]
            setattr(a, name, _Pattern(r.data_ptr(), c.data_ptr(), r.numel()))
        a.nh, a.ne, a.regularization, a.eliminate_equalities = nh, ne, 1e-6, 1
        a.barrier = _BarrierC(0, 0, 100.0, 2e-5)
        if workgroup:
            monkeypatch.setenv("UNGAR_AMD_ASSEMBLE_VARIANT", "workgroup")
        else:
            monkeypatch.delenv("UNGAR_AMD_ASSEMBLE_VARIANT", raising=False)
        assert lib.ungar_shooting_assemble(ctypes.byref(a), None) == 0, lib.ungar_last_error()
        torch.cuda.synchronize()
        res = {k: v.cpu().numpy() for k, v in out.items()}
        res["W"] = np.triu(res["W"].reshape(nodes, nd, nd))
        return res

    mine, theirs = run(False), run(True)
    assert (mine["piv"] == theirs["piv"]).all(), (mine["piv"], theirs["piv"])
    assert (theirs["piv"] >= 0).sum() >= B * N * min(4, ne // 2) and ((theirs["piv"] == -1).any() or not empty.any())  # (pivots were taken, empty rows took none)
    for key, tol in (("E", 1e-12), ("er", 1e-12), ("W", 1e-12), ("AB", 1e-12), ("w", 1e-9), ("b", 1e-9), ("dz0", 0.0)):
        assert np.isfinite(mine[key]).all(), key
        scale = np.abs(theirs[key]).max()
        assert np.abs(mine[key] - theirs[key]).max() <= tol * scale, (key, np.abs(mine[key] - theirs[key]).max(), scale)
