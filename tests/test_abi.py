"""CPU checks of the drop-in boundary: libungar_amd.so loads, exports every symbol that
include/ungar_amd.h declares, reports model metadata without a GPU, and FAILS LOUDLY (no fallback)
when asked to compute without a device or without the library."""
import ctypes
import os
import re

import numpy as np
import pytest

import ungar_amd


def _declared_symbols(repo_root):
    text = open(os.path.join(repo_root, "include", "ungar_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ungar_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(repo_root):
    lib = ungar_amd.load_library()
    names = _declared_symbols(repo_root)
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/ungar_amd.h but not exported: {missing}"


def test_header_cites_the_reference_interface(repo_root):
    text = open(os.path.join(repo_root, "include", "ungar_amd.h")).read()
    assert text.count("function.hpp:") >= 10 and "soft_sqp.hpp:257-264" in text


def test_model_metadata_without_gpu():
    dims = {"quadrotor": (13, 4, 0, 20, 118), "rc_car": (6, 2, 0, 15, 32), "srbd": (13, 24, 4, 6, 238), "anymal": (37, 12, 0, 1, 1691), "anymal_ad": (37, 12, 0, 1, 1691), "anymal_reg": (37, 12, 0, 1, 1691)}
    for name, (nx, nu, nw, npar, nnz) in dims.items():
        m = ungar_amd.NodeModel(name)
        assert (m.nx, m.nu, m.nw, m.np, m.ny) == (nx, nu, nw, npar, nx)
        assert m.independent_variable_size() == nx + nu and m.parameter_size() == nw + npar and m.dependent_variable_size() == nx
        assert m.implements_function() and m.implements_jacobian() and not m.implements_hessian()
        rows, cols = m.jacobian_sparsity()
        if nnz is not None:
            assert m.jac_nnz == nnz
        assert len(rows) == m.jac_nnz and rows.max() < nx and cols.max() < nx + nu
        key = rows.astype(np.int64) * (nx + nu) + cols
        assert (np.diff(key) > 0).all(), "CSR pattern must be canonical (row-major, ascending)"
        starts, outer = m.jacobian_csr()
        assert starts[-1] == m.jac_nnz and (np.diff(starts) >= 0).all()


def test_errors_are_reported_not_swallowed():
    lib = ungar_amd.load_library()
    with pytest.raises(ungar_amd.UngarError, match="unknown model"):
        ungar_amd.NodeModel("pendulum")
    h = ctypes.c_void_p()
    assert lib.ungar_model_open(None, ctypes.byref(h)) == -1
    assert b"null" in lib.ungar_last_error()
    m = ungar_amd.NodeModel("rc_car")
    rows = ctypes.POINTER(ctypes.c_int32)()
    nnz = ctypes.c_int64()
    assert lib.ungar_model_hessian_sparsity(m._h, ctypes.byref(rows), ctypes.byref(rows), ctypes.byref(nnz)) == -3  # UNSUPPORTED


def test_no_cpu_fallback_operands_must_be_device_tensors():
    import torch
    m = ungar_amd.NodeModel("rc_car")
    host = torch.zeros((4, 6), dtype=torch.float64)
    with pytest.raises(ungar_amd.UngarError, match="CUDA"):
        m.forward_zero(4, ungar_amd.Operand.aos(host, 6), ungar_amd.Operand.aos(host, 2), None, ungar_amd.Operand.aos(host, 15),
                       ungar_amd.Operand.aos(host, 6))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(ungar_amd, "_LIB", None)
    monkeypatch.setattr(ungar_amd, "library_path", lambda: str(tmp_path / "libungar_amd.so"))
    with pytest.raises(ungar_amd.UngarError, match="no CPU fallback"):
        ungar_amd.load_library()


def test_function_factory_rejects_bad_tapes():
    """Host logic of ungar_function_make (argument validation happens before any device work)."""
    lib = ungar_amd.load_library()

    class Node(ctypes.Structure):
        _fields_ = [("op", ctypes.c_int32), ("a", ctypes.c_int32), ("b", ctypes.c_int32), ("c", ctypes.c_int32), ("d", ctypes.c_int32),
                    ("reserved", ctypes.c_int32), ("value", ctypes.c_double)]

    fn = ctypes.c_void_p()
    out = (ctypes.c_int32 * 1)(1)
    # node 1 = add(node 0, node 5): forward reference -> invalid
    nodes = (Node * 2)(Node(1, 0, -1, -1, -1, 0, 0.0), Node(2, 0, 5, -1, -1, 0, 0.0))
    lib.ungar_function_make.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    rc = lib.ungar_function_make(nodes, 2, out, 1, 1, 0, b"bad", 2, None, 0, ctypes.byref(fn))
    assert rc == -1 and b"references" in lib.ungar_last_error()
    # Hessian of a vector-valued function: reference asserts "implemented only for scalar functions"
    nodes = (Node * 2)(Node(1, 0, -1, -1, -1, 0, 0.0), Node(7, 0, -1, -1, -1, 0, 0.0))
    out2 = (ctypes.c_int32 * 2)(0, 1)
    rc = lib.ungar_function_make(nodes, 2, out2, 2, 1, 0, b"bad", 6, None, 0, ctypes.byref(fn))
    assert rc == -3 and b"scalar functions" in lib.ungar_last_error()


def test_header_is_plain_c(repo_root, tmp_path):
    """include/ungar_amd.h compiled as C99 (-pedantic -Werror) and linked against the library: no C++ leaks into
    the boundary; the host-only entry points (open / info / version) work without a GPU."""
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(repo_root, "ungar_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(repo_root, "include"),
                    os.path.join(repo_root, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", lib, "-lungar_amd", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert "quadrotor_cost nx=13 ny=1 hes_nnz=17" in out and "argument checks: 0 unexpected" in out


def _amd_model_test(repo_root, mode, tmp_path, env=None):
    import subprocess
    exe = os.path.join(repo_root, "build", "amd_model_test")
    if not os.path.exists(exe):
        pytest.skip("build/amd_model_test missing: run __graft_entry__.build()")
    r = subprocess.run([exe, mode, str(tmp_path / "codegen")], capture_output=True, text=True, timeout=600, env={**os.environ, **(env or {})})
    assert r.returncode == 0 and "amd_model_test OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_reference_side_adapter_compiles_and_reports_sparsity(repo_root, tmp_path):
    """include/ungar_amd_model.hpp (AmdModel + TapeBuilder) with a hand-built tape; hipcc
    cross-compiles the kernels here, evaluation is left to the GPU test below."""
    _amd_model_test(repo_root, "sparsity", tmp_path, {"UNGAR_AMD_COMPILE_ONLY": "1"})


@pytest.mark.gpu
def test_reference_side_adapter_evaluates_closed_forms_on_gpu(repo_root, tmp_path):
    _amd_model_test(repo_root, "gpu", tmp_path)


def test_shipped_library_has_no_measurement_switches(repo_root):
    """A/B routes, per-phase clocks and experiment knobs are read from the environment only in the measurement build (csrc/runtime/measurement.hpp): the shipped
    library must not even contain their names -- a user environment that happens to carry UNGAR_AMD_ASSEMBLE_SKIP_SUBSTITUTION must not change what is computed.
    The names are collected from the sources (every UNGAR_MEASUREMENT_SWITCH("...")); the measurement build has them and says so."""
    import glob
    import re
    import ungar_amd
    names, library_names = set(), set()
    for path in glob.glob(os.path.join(repo_root, "ungar_amd", "csrc", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".hip", ".cpp", ".hpp")) and os.sep + "gen" + os.sep not in path:
            found = re.findall(r'UNGAR_MEASUREMENT_SWITCH\("([A-Z0-9_]+)"\)', open(path).read())
            names.update(found)
            if os.sep + "tape" + os.sep not in path:  # (the emitter's diagnostics are reached from the code generator, a build tool, not from the library)
                library_names.update(found)
    assert len(library_names) >= 15, library_names
    shipped = open(ungar_amd.library_path(), "rb").read()
    measurement_path = ungar_amd.measurement_library_path()
    assert os.path.exists(measurement_path), f"{measurement_path} missing: run __graft_entry__.build()"
    measurement = open(measurement_path, "rb").read()
    for name in sorted(names):
        assert name.encode() not in shipped, f"{name} is readable by the shipped library"
        assert name not in library_names or name.encode() in measurement, f"{name} missing from the measurement build"
    # no other getenv of an UNGAR_* name than the documented interface variables
    interface = {"UNGAR_HIPCC", "UNGAR_CODEGEN_FOLDER", "UNGAR_AMD_SCALAR_STORES", "UNGAR_AMD_JIT_FLAGS", "UNGAR_AMD_JACOBIAN_MODE", "UNGAR_AMD_VERBOSE", "UNGAR_AMD_KEEP_SOURCE",
                 "UNGAR_AMD_COMPILE_ONLY", "UNGAR_AMD_KERNEL_SOURCES", "UNGAR_AMD_JIT_JOBS", "UNGAR_AMD_HOST_CALL_RESIDENT_US"}
    found = set(m.decode() for m in re.findall(rb"UNGAR_[A-Z0-9_]{3,}", shipped))
    env_like = {n for n in found if n.startswith(("UNGAR_AMD_", "UNGAR_GN_", "UNGAR_HIPCC", "UNGAR_CODEGEN"))}
    undocumented = {n for n in env_like if n not in interface and not n.startswith(("UNGAR_AMD_EMITTER", "UNGAR_AMD_ABI", "UNGAR_AMD_H_", "UNGAR_AMD_DEFINE"))}
    integration = open(os.path.join(repo_root, "INTEGRATION.md")).read()
    for n in sorted(interface):
        assert n in integration, f"{n} is read by the shipped library but not documented in INTEGRATION.md"
    assert not undocumented, undocumented
    lib = ctypes.CDLL(measurement_path)
    lib.ungar_measurement_build.restype = ctypes.c_int32
    assert lib.ungar_measurement_build() == 1
    assert ungar_amd.load_library().ungar_measurement_build() == 0
