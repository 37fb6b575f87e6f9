"""Run-time function cache (reference function.hpp:420-451, 485-503 keeps an existence-only
`.so` cache).  A cache entry here is {meta: sparsity patterns + kernel table, one gfx950 code object per kernel}, keyed by a
hash of (optimised tape, enabled derivatives, arch, ROCm version, emitter build id, JIT flags).  hipcc cross-compiles without a
GPU, so the whole life cycle is checked on the CPU with UNGAR_AMD_COMPILE_ONLY=1: miss -> publish, hit (derive / emit / compile
skipped, same sparsity), an edited function never hits the stale entry, damaged entries are rebuilt, publishing is atomic
under concurrent builders, odd characters in the folder name survive the shell.  The GPU half (a function loaded from a hit
evaluates exactly like a freshly compiled one) is tests/cpp/function_test.cpp, run by tests/test_cpp_facade.py."""
import ctypes
import multiprocessing as mp
import os
import time

import pytest

import ungar_amd


class Node(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("a", ctypes.c_int32), ("b", ctypes.c_int32), ("c", ctypes.c_int32), ("d", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("value", ctypes.c_double)]


class Info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in ("n", "p", "m", "jac_nnz", "hes_nnz", "cache_hit")]


CONST, INPUT, ADD, MUL, SIN = 0, 1, 2, 4, 7


def _tape(scale):
    """y = scale * (x0^2 + x1^2 + x2^2) * p0 + sin(x0): n = 3, p = 1, m = 1."""
    nodes = [Node(INPUT, i, -1, -1, -1, 0, 0.0) for i in range(4)]               # 0..3: x0 x1 x2 p0
    nodes += [Node(MUL, i, i, -1, -1, 0, 0.0) for i in range(3)]                 # 4..6: squares
    nodes += [Node(ADD, 4, 5, -1, -1, 0, 0.0), Node(ADD, 7, 6, -1, -1, 0, 0.0)]  # 7, 8: sum
    nodes += [Node(CONST, -1, -1, -1, -1, 0, scale), Node(MUL, 8, 9, -1, -1, 0, 0.0), Node(MUL, 10, 3, -1, -1, 0, 0.0)]  # 9, 10, 11
    nodes += [Node(SIN, 0, -1, -1, -1, 0, 0.0), Node(ADD, 11, 12, -1, -1, 0, 0.0)]  # 12, 13
    return (Node * len(nodes))(*nodes), len(nodes), (ctypes.c_int32 * 1)(13)


def _make(folder, scale=1.0, enabled=6, recompile=0, name="cache_probe"):
    lib = ungar_amd.load_library()
    lib.ungar_function_make.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.ungar_function_free.argtypes = [ctypes.c_void_p]
    lib.ungar_function_free.restype = None
    lib.ungar_function_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(Info)]
    i32pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_int32))
    lib.ungar_function_jacobian_sparsity.argtypes = [ctypes.c_void_p, i32pp, i32pp, ctypes.POINTER(ctypes.c_int64)]
    lib.ungar_function_hessian_sparsity.argtypes = [ctypes.c_void_p, i32pp, i32pp, ctypes.POINTER(ctypes.c_int64)]
    nodes, count, outs = _tape(scale)
    fn = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = lib.ungar_function_make(nodes, count, outs, 1, 3, 1, name.encode(), enabled, str(folder).encode(), recompile, ctypes.byref(fn))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.ungar_last_error().decode()
    info = Info()
    assert lib.ungar_function_get_info(fn, ctypes.byref(info)) == 0

    def pattern(query):
        rows, cols, nnz = ctypes.POINTER(ctypes.c_int32)(), ctypes.POINTER(ctypes.c_int32)(), ctypes.c_int64()
        if query(fn, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(nnz)) != 0:
            return None
        return [(rows[k], cols[k]) for k in range(nnz.value)]

    out = {"hit": bool(info.cache_hit), "jac": pattern(lib.ungar_function_jacobian_sparsity), "hes": pattern(lib.ungar_function_hessian_sparsity),
           "jac_nnz": info.jac_nnz, "hes_nnz": info.hes_nnz, "seconds": dt}
    lib.ungar_function_free(fn)
    return out


@pytest.fixture(autouse=True)
def compile_only(monkeypatch):
    monkeypatch.setenv("UNGAR_AMD_COMPILE_ONLY", "1")


def _entry_files(folder, name="cache_probe"):
    d = os.path.join(str(folder), name, "ungar_amd")
    return sorted(os.listdir(d)) if os.path.isdir(d) else []


def test_miss_publish_hit_and_edit_invalidates(tmp_path):
    first = _make(tmp_path)
    assert not first["hit"] and first["jac"] == [(0, 0), (0, 1), (0, 2)] and first["hes"] == [(0, 0), (1, 1), (2, 2)]
    files = _entry_files(tmp_path)
    assert len([f for f in files if f.endswith(".meta")]) == 1 and len([f for f in files if f.endswith(".hsaco")]) == 3 and len([f for f in files if f.endswith(".lock")]) == 1
    assert not [f for f in files if f.endswith(".tmp") or f.endswith(".hip")], files  # nothing half-published, sources removed
    second = _make(tmp_path)
    assert second["hit"] and second["jac"] == first["jac"] and second["hes"] == first["hes"]
    assert second["seconds"] < 0.25 * first["seconds"]  # derive / emit / compile skipped
    # an EDITED function of the same name must not pick up the stale entry (the reference's existence-only cache does)
    edited = _make(tmp_path, scale=2.0)
    assert not edited["hit"]
    assert len([f for f in _entry_files(tmp_path) if f.endswith(".meta")]) == 2
    assert _make(tmp_path, scale=2.0)["hit"] and _make(tmp_path)["hit"]  # both entries stay valid side by side
    # a different set of enabled derivatives is a different entry; recompile = true never hits
    jac_only = _make(tmp_path, enabled=2)
    assert not jac_only["hit"] and jac_only["hes"] is None and jac_only["jac_nnz"] == 3
    assert not _make(tmp_path, recompile=1)["hit"]
    assert _make(tmp_path)["hit"]


def test_damaged_entries_are_rebuilt(tmp_path):
    assert not _make(tmp_path)["hit"]
    d = os.path.join(str(tmp_path), "cache_probe", "ungar_amd")
    meta = [f for f in os.listdir(d) if f.endswith(".meta")][0]
    text = open(os.path.join(d, meta)).read()
    with open(os.path.join(d, meta), "w") as fh:  # flip one sparsity index: the trailing checksum no longer matches
        fh.write(text.replace("jac 3 0 0 0 1", "jac 3 0 0 0 2"))
    again = _make(tmp_path)
    assert not again["hit"] and again["jac"] == [(0, 0), (0, 1), (0, 2)]
    assert _make(tmp_path)["hit"]
    with open(os.path.join(d, meta), "w") as fh:  # truncated meta
        fh.write(text[: len(text) // 2])
    assert not _make(tmp_path)["hit"]
    obj = [f for f in os.listdir(d) if f.endswith("_jacobian.hsaco")][0]
    os.remove(os.path.join(d, obj))  # missing code object
    assert not _make(tmp_path)["hit"]
    with open(os.path.join(d, obj), "ab") as fh:  # size differs from the recorded one
        fh.write(b"junk")
    assert not _make(tmp_path)["hit"]
    assert _make(tmp_path)["hit"]


def test_folder_names_survive_the_shell(tmp_path):
    folder = tmp_path / "it's a (dir) & more; #1"
    assert not _make(folder)["hit"]
    assert _make(folder)["hit"]
    lib = ungar_amd.load_library()
    nodes, count, outs = _tape(1.0)
    fn = ctypes.c_void_p()
    for bad_name, bad_folder in ((b"probe", str(tmp_path / "a $HOME dir").encode()), (b"probe", str(tmp_path / "a `dir`").encode()), (b"pro'be", str(tmp_path).encode()),
                                 (b"pro/be", str(tmp_path).encode())):
        assert lib.ungar_function_make(nodes, count, outs, 1, 3, 1, bad_name, 6, bad_folder, 0, ctypes.byref(fn)) == -1  # the hipcc driver re-expands these
        assert b"unsupported character" in lib.ungar_last_error()


def _concurrent_builder(folder, q):
    os.environ["UNGAR_AMD_COMPILE_ONLY"] = "1"
    try:
        q.put(_make(folder, scale=3.0, name="cache_race"))
    except Exception as exc:  # noqa: BLE001
        q.put(repr(exc))


def test_concurrent_builders_of_the_same_function(tmp_path):
    """One process per GPU: every rank builds the same functions at start-up (a shared in-place source file
    let one rank truncate what another rank's compiler was reading)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_concurrent_builder, args=(str(tmp_path), q)) for _ in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(r, dict) and r["jac_nnz"] == 3 for r in results), results
    assert sum(1 for r in results if not r["hit"]) == 1, results  # one builder compiles, the others wait on the entry lock and hit
    files = _entry_files(tmp_path, "cache_race")
    assert not [f for f in files if f.endswith(".tmp") or f.endswith(".hip")], files
    assert _make(tmp_path, scale=3.0, name="cache_race")["hit"]


def _wide_tape(m):
    """y_j = (x_j + x_{j+1}) * x_{j+1} + x_j * p0, j < m (n = m + 1, p = 1): 4 m operations -- above the chunk budget of the run-time factory (6000 statements) for m = 1700."""
    n = m + 1
    nodes = [Node(INPUT, i, -1, -1, -1, 0, 0.0) for i in range(n + 1)]  # x0..xm, p0 (index n)
    outs = []
    for j in range(m):
        base = len(nodes)
        nodes += [Node(ADD, j, j + 1, -1, -1, 0, 0.0), Node(MUL, base, j + 1, -1, -1, 0, 0.0), Node(MUL, j, n, -1, -1, 0, 0.0), Node(ADD, base + 1, base + 2, -1, -1, 0, 0.0)]
        outs.append(base + 3)
    return (Node * len(nodes))(*nodes), len(nodes), (ctypes.c_int32 * m)(*outs), n


def _make_wide(folder, m=1700):
    lib = ungar_amd.load_library()
    lib.ungar_function_make.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.ungar_function_free.argtypes = [ctypes.c_void_p]
    lib.ungar_function_free.restype = None
    lib.ungar_function_get_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(Info)]
    nodes, count, outs, n = _wide_tape(m)
    fn = ctypes.c_void_p()
    rc = lib.ungar_function_make(nodes, count, outs, m, n, 1, b"wide_probe", 1, str(folder).encode(), 0, ctypes.byref(fn))  # enabled = NONE (1): value only
    assert rc == 0, lib.ungar_last_error().decode()
    info = Info()
    assert lib.ungar_function_get_info(fn, ctypes.byref(info)) == 0
    return lib, fn, info, n


def test_a_large_body_is_compiled_in_chunks(tmp_path, monkeypatch):
    """A derivative whose body exceeds the statement budget is cut into chunks of consecutive outputs -- one kernel, one code object and one compiler process per
    chunk, all started together (the equality-constraint Jacobian of the reference's quadruped OCP: 85 s -> under 10 s of its cold start): the entry lists them
    (value, value.1, ...) and a second make finds them all."""
    monkeypatch.setenv("UNGAR_AMD_COMPILE_ONLY", "1")
    lib, fn, info, _ = _make_wide(tmp_path)
    assert info.cache_hit == 0
    lib.ungar_function_free(fn)
    objects = sorted(f for f in os.listdir(tmp_path / "wide_probe" / "ungar_amd") if f.endswith(".hsaco"))
    assert len(objects) >= 2 and any(f.endswith("_value.hsaco") for f in objects) and any(f.endswith("_value.1.hsaco") for f in objects), objects
    meta = next(f for f in os.listdir(tmp_path / "wide_probe" / "ungar_amd") if f.endswith(".meta"))
    text = open(tmp_path / "wide_probe" / "ungar_amd" / meta).read()
    assert f"units {len(objects)}" in text and "unit value.1 ungar_fn_forward_zero_c1" in text
    lib, fn, info, _ = _make_wide(tmp_path)
    assert info.cache_hit == 1
    lib.ungar_function_free(fn)
    os.remove(tmp_path / "wide_probe" / "ungar_amd" / objects[-1])  # a chunk missing: the entry is rebuilt, not half loaded
    lib, fn, info, _ = _make_wide(tmp_path)
    assert info.cache_hit == 0
    lib.ungar_function_free(fn)


@pytest.mark.gpu
def test_chunked_kernels_write_every_output(tmp_path, monkeypatch):
    """The chunks of a large body on the device: every output of the single-instance host call equals the closed form (each chunk writes its own range of the
    same operand; a chunk that is not launched leaves its range unwritten)."""
    import numpy as np
    monkeypatch.delenv("UNGAR_AMD_COMPILE_ONLY", raising=False)  # (this module's other tests stop after publishing the entry)
    lib, fn, info, n = _make_wide(tmp_path)
    lib.ungar_function_eval_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(3)
    xp = rng.uniform(-1.0, 1.0, n + 1)
    y = np.full(info.m, np.nan)
    assert lib.ungar_function_eval_host(fn, 0, xp.ctypes.data, y.ctypes.data) == 0, lib.ungar_last_error().decode()
    x, p0 = xp[:n], xp[n]
    expect = (x[:-1] + x[1:]) * x[1:] + x[:-1] * p0
    assert not np.isnan(y).any()
    assert np.abs(y - expect).max() <= 1e-14
    lib.ungar_function_free(fn)


def _make_loaded(folder, scale=1.0, name="resident_probe"):
    lib = ungar_amd.load_library()
    lib.ungar_function_make.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.ungar_function_free.argtypes = [ctypes.c_void_p]
    lib.ungar_function_free.restype = None
    lib.ungar_function_eval_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.ungar_function_host_call_resident.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.ungar_function_host_call_resident.restype = ctypes.c_int32
    nodes, count, outs = _tape(scale)
    fn = ctypes.c_void_p()
    rc = lib.ungar_function_make(nodes, count, outs, 1, 3, 1, name.encode(), 6, str(folder).encode(), 0, ctypes.byref(fn))
    assert rc == 0, lib.ungar_last_error().decode()
    return lib, fn


def _closed_forms(xp, scale):
    """y = scale (x0^2 + x1^2 + x2^2) p0 + sin x0: value, Jacobian over x (dense row), upper Hessian entries keyed by (row, col)."""
    import numpy as np
    x, p0 = xp[:3], xp[3]
    value = scale * float(x @ x) * p0 + np.sin(x[0])
    jac = 2.0 * scale * p0 * x
    jac[0] += np.cos(x[0])
    hes = {(i, i): 2.0 * scale * p0 for i in range(3)}
    hes[(0, 0)] -= np.sin(x[0])
    return value, jac, hes


def _resident_calls(folder, rounds=3000):
    """value, Jacobian and Hessian of two node-sized functions through the single-instance host call, fresh inputs every call, against the closed forms."""
    import numpy as np
    lib, fn = _make_loaded(folder, 1.0)
    lib2, fn2 = _make_loaded(folder, 2.5, name="resident_probe_b")
    assert [lib.ungar_function_host_call_resident(fn, w) for w in (0, 1, 2)] == [1, 1, 1]
    i32pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_int32))
    lib.ungar_function_jacobian_sparsity.argtypes = [ctypes.c_void_p, i32pp, i32pp, ctypes.POINTER(ctypes.c_int64)]
    lib.ungar_function_hessian_sparsity.argtypes = [ctypes.c_void_p, i32pp, i32pp, ctypes.POINTER(ctypes.c_int64)]

    def pattern(query, f):
        rows, cols, nnz = ctypes.POINTER(ctypes.c_int32)(), ctypes.POINTER(ctypes.c_int32)(), ctypes.c_int64()
        assert query(f, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(nnz)) == 0
        return [(rows[k], cols[k]) for k in range(nnz.value)]

    jac_pattern, hes_pattern = pattern(lib.ungar_function_jacobian_sparsity, fn), pattern(lib.ungar_function_hessian_sparsity, fn)
    rng = np.random.default_rng(11)

    def check(f, scale, what):
        xp = rng.uniform(-2.0, 2.0, 4)
        out = np.full(max(1, len(jac_pattern), len(hes_pattern)), np.nan)
        assert lib.ungar_function_eval_host(f, what, xp.ctypes.data, out.ctypes.data) == 0, lib.ungar_last_error().decode()
        value, jac, hes = _closed_forms(xp, scale)
        if what == 0:
            assert abs(out[0] - value) <= 1e-13 * (1.0 + abs(value))
        elif what == 1:
            for k, (_, c) in enumerate(jac_pattern):
                assert abs(out[k] - jac[c]) <= 1e-13 * (1.0 + abs(jac[c])), (k, out[k], jac[c])
        else:
            for k, rc in enumerate(hes_pattern):
                assert abs(out[k] - hes.get(rc, 0.0)) <= 1e-13 * (1.0 + abs(hes.get(rc, 0.0))), (rc, out[k])

    for i in range(rounds):  # back to back: the kernels stay
        check(fn, 1.0, i % 3)
    for i in range(12):  # further apart than the idle time: launched again each time
        time.sleep(0.002)
        check(fn, 1.0, i % 3)
    for i in range(rounds // 5):  # two functions in turn, six kernels alive
        check(fn if i % 2 else fn2, 1.0 if i % 2 else 2.5, (i // 2) % 3)
    t0 = time.perf_counter()
    xp, out = rng.uniform(-1.0, 1.0, 4), np.zeros(4)
    for _ in range(2000):
        lib.ungar_function_eval_host(fn, 1, xp.ctypes.data, out.ctypes.data)
    per_call_us = (time.perf_counter() - t0) / 2000 * 1e6
    print(f"resident single-instance Jacobian through ctypes: {per_call_us:.2f} us per call")
    lib.ungar_function_free(fn)
    lib2.ungar_function_free(fn2)
    import torch
    torch.cuda.synchronize()  # nothing is left running


@pytest.mark.gpu
def test_single_instance_host_calls_through_the_resident_kernel(tmp_path, monkeypatch):
    """Node-sized functions answer single-instance host calls from a kernel that stays on the device (ungar_amd.h: ungar_function_eval_host): value, Jacobian and
    Hessian interleaved over thousands of calls with fresh inputs each equal their closed forms -- every call reads ITS inputs, not an earlier call's -- and so do
    calls spaced further apart than the kernel's idle time (it has returned in between and is launched again) and calls on two functions in turn."""
    monkeypatch.delenv("UNGAR_AMD_COMPILE_ONLY", raising=False)
    monkeypatch.delenv("UNGAR_AMD_HOST_CALL_RESIDENT_US", raising=False)
    _resident_calls(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["doorbell in host memory", "no resident kernel"])
def test_single_instance_host_calls_on_the_other_routes(tmp_path, route):
    """The same calls where the device does not expose its memory to the host (doorbell and inputs in mapped host memory: forced through the measurement build),
    and with the resident kernel switched off (UNGAR_AMD_HOST_CALL_RESIDENT_US=0: one launch per call)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "UNGAR_AMD_COMPILE_ONLY"}
    program = "import sys; sys.path.insert(0, 'tests'); import test_function_cache as t; t._resident_calls(sys.argv[1], 600)"
    if route == "doorbell in host memory":
        env["UNGAR_AMD_LIBRARY"] = ungar_amd.measurement_library_path()
        env["UNGAR_AMD_HOST_CALL_NO_APERTURE"] = "1"
    else:
        env["UNGAR_AMD_HOST_CALL_RESIDENT_US"] = "0"
        program = program.replace("t._resident_calls(sys.argv[1], 600)", "t._resident_calls_without(sys.argv[1])")
    r = subprocess.run([sys.executable, "-c", program, str(tmp_path)], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _resident_calls_without(folder):
    """UNGAR_AMD_HOST_CALL_RESIDENT_US=0: the library reports no resident kernel, and the calls (one launch each) give the same results."""
    import numpy as np
    lib, fn = _make_loaded(folder, 1.0)
    assert [lib.ungar_function_host_call_resident(fn, w) for w in (0, 1, 2)] == [0, 0, 0]
    rng = np.random.default_rng(5)
    for _ in range(200):
        xp, out = rng.uniform(-2.0, 2.0, 4), np.full(1, np.nan)
        assert lib.ungar_function_eval_host(fn, 0, xp.ctypes.data, out.ctypes.data) == 0
        value, _, _ = _closed_forms(xp, 1.0)
        assert abs(out[0] - value) <= 1e-13 * (1.0 + abs(value))
    lib.ungar_function_free(fn)
