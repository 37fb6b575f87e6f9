"""ungar_amd -- MI355X-native batched derivative evaluation for Ungar's NMPC hot path.

Python face of the C ABI in include/ungar_amd.h (ctypes; torch only supplies device memory and
streams).  There is NO CPU fallback: importing works anywhere (so that host logic can be tested),
but every evaluation goes through libungar_amd.so and raises if the library is missing.

Reference interface mirrored here:
  NodeModel.forward_zero / sparse_jacobian / dense_jacobian  <->  Ungar::Autodiff::Function::
      operator() / Jacobian                       include/ungar/autodiff/function.hpp:206-230
  NodeModel.independent_variable_size / parameter_size / dependent_variable_size
                                                  include/ungar/autodiff/function.hpp:350-361
  NodeModel.jacobian_sparsity (CSR inner starts / outer indices)
                                                  include/ungar/autodiff/function.hpp:98-134
  gn_hessian                                      include/ungar/optimization/soft_sqp.hpp:257-264
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

__all__ = ["NodeModel", "Operand", "gn_hessian", "gn_hessian_lanes", "gn_hessian_tiles", "transpose_nodes", "gn_hessian_unit_fastest", "library_path", "load_library", "UngarError", "MODELS"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH_OVERRIDE: str | None = None  # use_library()
_LOADED: dict = {}                 # path -> declared CDLL
MODELS = ("quadrotor", "rc_car", "srbd", "anymal", "anymal_ad", "anymal_reg")
RBD_MODELS = ("anymal_rnea", "anymal_crba", "anymal_minv", "anymal_feet", "anymal_centroidal")  # rigid-body quantities as node models


class UngarError(RuntimeError):
    """An ungar_amd C-ABI call failed (message from ungar_last_error())."""


def library_path() -> str:
    """In-tree library; `use_library()` / UNGAR_AMD_LIBRARY override it (another build of the same ABI, e.g. `measurement_library_path()`)."""
    return _PATH_OVERRIDE or os.environ.get("UNGAR_AMD_LIBRARY") or os.path.join(_HERE, "lib", "libungar_amd.so")


def use_library(path: str | None) -> None:
    """Makes `path` (None: back to the default) the library every later `load_library()` returns; libraries already loaded stay loaded and are
    reused.  For the agreement tests between two kernel routes and for tools/, which need the measurement build for part of a session."""
    global _PATH_OVERRIDE, _LIB
    _PATH_OVERRIDE = path
    _LIB = None


def measurement_library_path() -> str:
    """The measurement build of the library (-DUNGAR_AMD_MEASUREMENT, csrc/runtime/measurement.hpp): the only build in which the A/B routes,
    per-phase clocks and experiment knobs of tools/ can be selected through the environment.  Same ABI; load it by pointing UNGAR_AMD_LIBRARY
    (Python) or LD_LIBRARY_PATH (the C++ programs) at it."""
    return os.path.join(_HERE, "lib", "measurement", "libungar_amd.so")


class _ModelInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in ("nx", "nu", "nw", "np", "ny", "jac_nnz", "hes_nnz")]


class _Operand(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("instance_stride", ctypes.c_int64), ("knot_stride", ctypes.c_int64),
                ("element_stride", ctypes.c_int64)]


class _TileLayout(ctypes.Structure):
    _fields_ = [("nodes_per_tile", ctypes.c_int32), ("band_tiles", ctypes.c_int32), ("images", ctypes.c_int32), ("unit_doubles", ctypes.c_int32),
                ("entries", ctypes.c_int32), ("reserved", ctypes.c_int32), ("entry_of_slot", ctypes.POINTER(ctypes.c_int16))]


class _NodeBatch(ctypes.Structure):
    _fields_ = [("count", ctypes.c_int64), ("knots", ctypes.c_int64), ("x", _Operand), ("u", _Operand), ("w", _Operand),
                ("p", _Operand), ("f", _Operand), ("jac", _Operand)]


_LIB = None


ABI_VERSION = 7  # UNGAR_AMD_ABI_VERSION of include/ungar_amd.h these bindings mirror


def _share_the_hip_runtime_of_torch() -> None:
    """PyTorch bundles its own libamdhip64.so (same SONAME as ROCm's) and loads it when it is imported.  A library dlopen'ed BEFORE that resolves its libamdhip64.so.7 to ROCm's copy
    through its RUNPATH, torch then brings its own: two HIP runtimes in one process, and whichever initialises second finds no device ("no ROCm-capable device is
    detected" from the first launch).  Where torch is installed its copy is therefore loaded first (whether or not it has been imported yet): the dynamic linker reuses it for ours by SONAME, and device
    memory, streams and events are shared with torch as the bindings assume."""
    import importlib.util
    import sys
    torch = sys.modules.get("torch")
    origin = getattr(torch, "__file__", None)
    if origin is None:  # not imported (yet): it may be later in this process -- locate it without importing it
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        origin = spec.origin if spec is not None else None
    if not origin:
        return
    bundled = os.path.join(os.path.dirname(origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> ctypes.CDLL:
    """dlopen libungar_amd.so and declare every symbol of include/ungar_amd.h.  Raises loudly if the
    HIP library has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if path in _LOADED:
        _LIB = _LOADED[path]
        return _LIB
    if not os.path.exists(path):
        raise UngarError(f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(ungar_amd has no CPU fallback).")
    _share_the_hip_runtime_of_torch()
    lib = ctypes.CDLL(path)
    try:
        lib.ungar_abi_version.restype = ctypes.c_int32
    except AttributeError:
        raise UngarError(f"{path} predates ABI versioning (no ungar_abi_version): rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`") from None
    if lib.ungar_abi_version() != ABI_VERSION:  # the structs below mirror include/ungar_amd.h at this version: a mismatch would shift arguments silently
        raise UngarError(f"{path} reports ABI version {lib.ungar_abi_version()}, these bindings were written for {ABI_VERSION} (include/ungar_amd.h: UNGAR_AMD_ABI_VERSION)")
    vp, i64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)
    i32pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_int32))
    lib.ungar_model_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    lib.ungar_model_close.argtypes = [vp]
    lib.ungar_model_close.restype = None
    lib.ungar_model_name.argtypes = [vp]
    lib.ungar_model_name.restype = ctypes.c_char_p
    lib.ungar_model_get_info.argtypes = [vp, ctypes.POINTER(_ModelInfo)]
    lib.ungar_model_jacobian_sparsity.argtypes = [vp, i32pp, i32pp, i64p]
    lib.ungar_model_hessian_sparsity.argtypes = [vp, i32pp, i32pp, i64p]
    for name in ("ungar_model_has_forward_zero", "ungar_model_has_sparse_jacobian", "ungar_model_has_sparse_hessian"):
        getattr(lib, name).argtypes = [vp]
    for name in ("ungar_model_forward_zero", "ungar_model_sparse_jacobian", "ungar_model_dense_jacobian"):
        getattr(lib, name).argtypes = [vp, ctypes.POINTER(_NodeBatch), vp]
    lib.ungar_model_sparse_hessian.argtypes = [vp, ctypes.POINTER(_NodeBatch), ctypes.POINTER(_Operand), vp]
    for name in ("ungar_gn_hessian", "ungar_gn_hessian_upper"):
        getattr(lib, name).argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, vp]
    lib.ungar_gn_hessian_upper_unit_fastest.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64,
                                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, vp]
    lib.ungar_transpose_nodes.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, vp]
    lib.ungar_gn_hessian_upper_tiles.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int64, vp]
    lib.ungar_gn_hessian_upper_lanes.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int64, vp]
    lib.ungar_model_tile_layout.argtypes = [vp, ctypes.POINTER(_TileLayout)]
    lib.ungar_model_tile_doubles.argtypes = [vp, ctypes.c_int64]
    lib.ungar_model_tile_doubles.restype = ctypes.c_int64
    lib.ungar_model_dense_jacobian_tiles.argtypes = [vp, ctypes.POINTER(_NodeBatch), vp, vp]
    lib.ungar_tiles_gather.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(_Operand), vp]
    lib.ungar_model_prepare.argtypes = [vp]
    lib.ungar_ocp_equality_sparsity.argtypes = [vp, ctypes.c_int64, vp, vp, i64p]
    lib.ungar_ocp_assemble_equality.argtypes = [vp, ctypes.c_int64, ctypes.c_int64] + [ctypes.POINTER(_Operand)] * 6 + [vp]
    lib.ungar_last_error.restype = ctypes.c_char_p
    lib.ungar_version.restype = ctypes.c_char_p
    _LIB = _LOADED[path] = lib
    return lib


def _check(code: int):
    if code != 0:
        raise UngarError(f"ungar_amd error {code}: {load_library().ungar_last_error().decode()}")


@dataclass
class Operand:
    """Strided device view of one operand (see `ungar_operand` in include/ungar_amd.h)."""
    tensor: object  # torch.Tensor (cuda, float64) or None
    instance_stride: int = 0
    knot_stride: int = 0
    element_stride: int = 1

    def _c(self) -> _Operand:
        if self.tensor is None:
            return _Operand(None, 0, 0, 0)
        t = self.tensor
        if not t.is_cuda or str(t.dtype) != "torch.float64":
            raise UngarError("operands must be float64 CUDA tensors (device memory; the engine has no host path)")
        return _Operand(t.data_ptr(), self.instance_stride, self.knot_stride, self.element_stride)

    # -- the two canonical layouts over a flat batch of `count` nodes --------------------------
    @staticmethod
    def soa(tensor, count: int, knots: int = 1) -> "Operand":
        """tensor shape (elements, count): node index is the fastest axis (coalesced)."""
        return Operand(tensor, instance_stride=knots, knot_stride=1, element_stride=count)

    @staticmethod
    def aos(tensor, elements: int, knots: int = 1, ld: int | None = None) -> "Operand":
        """tensor shape (count, ld>=elements): node-major."""
        ld = elements if ld is None else ld
        return Operand(tensor, instance_stride=knots * ld, knot_stride=ld, element_stride=1)

    @staticmethod
    def per_instance(tensor, elements: int, shared: bool = False) -> "Operand":
        """per-instance parameters, shape (instances, elements) or (elements,) when shared."""
        return Operand(tensor, instance_stride=0 if shared else elements, knot_stride=0, element_stride=1)


class NodeModel:
    """A compiled shooting-node model  x+ = f(x, u; w, p)  with its sparse Jacobian."""

    def __init__(self, name: str):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        _check(self._lib.ungar_model_open(name.encode(), ctypes.byref(self._h)))
        info = _ModelInfo()
        _check(self._lib.ungar_model_get_info(self._h, ctypes.byref(info)))
        self.name = name
        self.nx, self.nu, self.nw, self.np, self.ny = info.nx, info.nu, info.nw, info.np, info.ny
        self.jac_nnz, self.hes_nnz = info.jac_nnz, info.hes_nnz

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.ungar_model_close(self._h)
                self._h.value = None
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    # -- reference Function API names (function.hpp:340-361) ------------------------------------
    def independent_variable_size(self) -> int:
        return self.nx + self.nu

    def parameter_size(self) -> int:
        return self.nw + self.np

    def dependent_variable_size(self) -> int:
        return self.ny

    def implements_function(self) -> bool:
        return bool(self._lib.ungar_model_has_forward_zero(self._h))

    def implements_jacobian(self) -> bool:
        return bool(self._lib.ungar_model_has_sparse_jacobian(self._h))

    def implements_hessian(self) -> bool:
        return bool(self._lib.ungar_model_has_sparse_hessian(self._h))

    def jacobian_sparsity(self):
        """(rows, cols) int32 arrays, canonical row-major order."""
        rows = ctypes.POINTER(ctypes.c_int32)()
        cols = ctypes.POINTER(ctypes.c_int32)()
        nnz = ctypes.c_int64()
        _check(self._lib.ungar_model_jacobian_sparsity(self._h, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(nnz)))
        n = nnz.value
        return (np.ctypeslib.as_array(rows, shape=(n,)).copy(), np.ctypeslib.as_array(cols, shape=(n,)).copy())

    def jacobian_csr(self):
        """(inner_starts[ny+1], outer_indices[nnz]) -- the arrays Function's ctor builds
        (function.hpp:106-124)."""
        rows, cols = self.jacobian_sparsity()
        starts = np.zeros(self.ny + 1, dtype=np.int32)
        np.add.at(starts, rows + 1, 1)
        return np.cumsum(starts, dtype=np.int32), cols

    # -- batched evaluation -----------------------------------------------------------------------
    def _batch(self, count, knots, x, u, w, p, f, jac) -> _NodeBatch:
        none = Operand(None)
        return _NodeBatch(count, knots, x._c(), (u or none)._c(), (w or none)._c(), (p or none)._c(), (f or none)._c(), (jac or none)._c())

    @staticmethod
    def _stream(stream):
        if isinstance(stream, ctypes.c_void_p):
            return stream
        if stream is not None:
            return ctypes.c_void_p(stream)
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def forward_zero(self, count, x, u, w, p, f, knots=1, stream=None):
        b = self._batch(count, knots, x, u, w, p, f, None)
        _check(self._lib.ungar_model_forward_zero(self._h, ctypes.byref(b), self._stream(stream)))

    def sparse_jacobian(self, count, x, u, w, p, f, jac, knots=1, stream=None):
        b = self._batch(count, knots, x, u, w, p, f, jac)
        _check(self._lib.ungar_model_sparse_jacobian(self._h, ctypes.byref(b), self._stream(stream)))

    def dense_jacobian(self, count, x, u, w, p, f, jac, knots=1, stream=None):
        b = self._batch(count, knots, x, u, w, p, f, jac)
        _check(self._lib.ungar_model_dense_jacobian(self._h, ctypes.byref(b), self._stream(stream)))

    # -- wave tiles (include/ungar_amd.h: ungar_tile_layout) -----------------------------------------------------------
    def tile_layout(self):
        """dict(nodes_per_tile, band_tiles, images, unit_doubles, entries, entry_of_slot: int16 array of 4 * images) -- needs no device."""
        l = _TileLayout()
        _check(self._lib.ungar_model_tile_layout(self._h, ctypes.byref(l)))
        table = np.ctypeslib.as_array(l.entry_of_slot, shape=(4 * l.images,)).copy()
        return dict(nodes_per_tile=l.nodes_per_tile, band_tiles=l.band_tiles, images=l.images, unit_doubles=l.unit_doubles, entries=l.entries, entry_of_slot=table)

    def tile_doubles(self, count: int) -> int:
        n = self._lib.ungar_model_tile_doubles(self._h, count)
        if n < 0:
            _check(int(n))
        return int(n)

    def dense_jacobian_tiles(self, count, x, u, w, p, f, tiles, knots=1, stream=None):
        """Value into f (may be None) and the dense block of every node into the tile operand `tiles` (1-D float64 CUDA tensor of tile_doubles(count))."""
        if tiles.numel() < self.tile_doubles(count):
            raise UngarError(f"tile operand of {tiles.numel()} doubles for {count} nodes: {self.tile_doubles(count)} needed")
        b = self._batch(count, knots, x, u, w, p, f, None)
        _check(self._lib.ungar_model_dense_jacobian_tiles(self._h, ctypes.byref(b), ctypes.c_void_p(tiles.data_ptr()), self._stream(stream)))

    def tiles_gather(self, count, tiles, jac, knots=1, stream=None):
        """Tile operand -> strided dense operand `jac` (device)."""
        j = jac._c()
        _check(self._lib.ungar_tiles_gather(self._h, ctypes.c_void_p(tiles.data_ptr()), count, knots, ctypes.byref(j), self._stream(stream)))

    def untile_numpy(self, tiles: np.ndarray, count: int) -> np.ndarray:
        """Host-side reading of a tile operand through the layout table alone: (count, ny, nx + nu).  Independent of ungar_tiles_gather (tests)."""
        l = self.tile_layout()
        i = np.arange(count)
        t, n = i // l["nodes_per_tile"], i % l["nodes_per_tile"]
        g, r = t // l["band_tiles"], t % l["band_tiles"]
        out = np.full((count, l["entries"]), np.nan)
        for slot, e in enumerate(l["entry_of_slot"]):
            if e < 0:
                continue
            image, leg = slot // 4, slot % 4
            lane = 16 * (n // 4) + 4 * leg + n % 4
            out[:, e] = tiles[((g * (l["images"] // 2) + image // 2) * l["band_tiles"] + r) * l["unit_doubles"] + 2 * lane + image % 2]
        return out.reshape(count, self.ny, self.nx + self.nu)

    def sparse_hessian(self, count, x, u, w, p, f, grad, hes, knots=1, stream=None):
        """Scalar (cost) models: value into f, gradient w.r.t. (x, u) into grad (may be None), upper-triangular
        Hessian values (hessian_sparsity order) into hes."""
        b = self._batch(count, knots, x, u, w, p, f, grad)
        h = hes._c()
        _check(self._lib.ungar_model_sparse_hessian(self._h, ctypes.byref(b), ctypes.byref(h), self._stream(stream)))

    def hessian_sparsity(self):
        """(rows, cols) int32 arrays of the upper-triangular Hessian w.r.t. (x, u), canonical row-major order."""
        rows = ctypes.POINTER(ctypes.c_int32)()
        cols = ctypes.POINTER(ctypes.c_int32)()
        nnz = ctypes.c_int64()
        _check(self._lib.ungar_model_hessian_sparsity(self._h, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(nnz)))
        n = nnz.value
        return (np.ctypeslib.as_array(rows, shape=(n,)).copy(), np.ctypeslib.as_array(cols, shape=(n,)).copy())

    # -- whole-horizon assembly --------------------------------------------
    def ocp_equality_sparsity(self, horizon: int):
        """(row_starts, cols) of d g / d [X | U], g = [x0 - xm; x_{k+1} - f(x_k, u_k)], canonical CSR."""
        nnz = ctypes.c_int64()
        _check(self._lib.ungar_ocp_equality_sparsity(self._h, horizon, None, None, ctypes.byref(nnz)))
        starts = np.zeros((horizon + 1) * self.nx + 1, dtype=np.int32)
        cols = np.zeros(nnz.value, dtype=np.int32)
        _check(self._lib.ungar_ocp_equality_sparsity(self._h, horizon, starts.ctypes.data, cols.ctypes.data, ctypes.byref(nnz)))
        return starts, cols

    def prepare(self):
        """Uploads the node pattern to the current device (keeps the first assembly call allocation-free)."""
        _check(self._lib.ungar_model_prepare(self._h))

    def ocp_assemble_equality(self, horizon, batch, x, xm, f, jac, g, values, stream=None):
        ops = [o._c() for o in (x, xm, f, jac, g, values)]
        _check(self._lib.ungar_ocp_assemble_equality(self._h, horizon, batch, *[ctypes.byref(o) for o in ops], self._stream(stream)))

    # -- convenience: node-major numpy in, numpy out (tests, smoke) ---------------------------------
    def evaluate_numpy(self, x, u, w, p, mode="dense", layout="soa"):
        """Uploads (count, n) host arrays, evaluates on cuda:0, returns (f, J) as host arrays with
        J dense (count, ny, nx+nu).  `layout` selects the DEVICE layout being exercised."""
        import torch
        count = x.shape[0]
        dev = torch.device("cuda", 0)
        ncols = self.nx + self.nu
        nj = self.jac_nnz if mode == "sparse" else self.ny * ncols

        def up(a, n):
            if n == 0:
                return None
            t = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)
            return (t.t().contiguous() if layout == "soa" else t.contiguous()).to(dev)

        self._tiles_gather_requested = layout == "tiles_gather"
        tiled = layout in ("tiles", "tiles_gather")  # wave-tile output (dense mode): read on the host through the layout table / gathered on the device
        if tiled:
            if mode != "dense":
                raise UngarError("the wave-tile layout holds the dense block")
            layout = "soa"
        xt, ut, wt = up(x, self.nx), up(u, self.nu), up(w, self.nw)
        pt = torch.as_tensor(np.ascontiguousarray(p), dtype=torch.float64).to(dev) if self.np else None
        if layout == "soa":
            ft = torch.full((self.ny, count), float("nan"), dtype=torch.float64, device=dev)
            jt = torch.full((nj, count), float("nan"), dtype=torch.float64, device=dev)
            mk = lambda t, n: None if t is None else Operand.soa(t, count)  # noqa: E731
        else:
            ft = torch.full((count, self.ny), float("nan"), dtype=torch.float64, device=dev)
            jt = torch.full((count, nj), float("nan"), dtype=torch.float64, device=dev)
            mk = lambda t, n: None if t is None else Operand.aos(t, n)  # noqa: E731
        args = (count, mk(xt, self.nx), mk(ut, self.nu), mk(wt, self.nw), Operand.per_instance(pt, self.np) if self.np else None, mk(ft, self.ny))
        if mode == "value":
            self.forward_zero(*args)
            torch.cuda.synchronize()
            f = ft.cpu().numpy()
            return (f.T.copy() if layout == "soa" else f), None
        if tiled:
            tiles = torch.full((self.tile_doubles(count),), float("nan"), dtype=torch.float64, device=dev)
            self.dense_jacobian_tiles(*args, tiles)
            if gather_on_device := (tiled and nj and layout == "soa" and self._tiles_gather_requested):
                self.tiles_gather(count, tiles, mk(jt, nj))
            torch.cuda.synchronize()
            if not gather_on_device:
                return ft.cpu().numpy().T.copy(), self.untile_numpy(tiles.cpu().numpy(), count)
        else:
            (self.sparse_jacobian if mode == "sparse" else self.dense_jacobian)(*args, mk(jt, nj))
        torch.cuda.synchronize()
        f, j = ft.cpu().numpy(), jt.cpu().numpy()
        if layout == "soa":
            f, j = f.T.copy(), j.T.copy()
        if mode == "sparse":
            rows, cols = self.jacobian_sparsity()
            dense = np.zeros((count, self.ny, ncols))
            dense[:, rows, cols] = j
            return f, dense
        return f, j.reshape(count, self.ny, ncols)


def gn_hessian_unit_fastest(jac, d, g, rows: int, cols: int, count: int, ld_g=None, stream=None):
    """Upper triangle of G = J^T diag(d) J for a Jacobian in the unit-fastest layout of the node kernels:
    jac (rows * cols, count), d (rows, count) or None; g (count, cols, ld_g) node-major."""
    lib = load_library()
    ld_g = cols if ld_g is None else ld_g
    _check(lib.ungar_gn_hessian_upper_unit_fastest(jac.data_ptr(), jac.stride(0), d.data_ptr() if d is not None else None,
                                                   d.stride(0) if d is not None else 0, g.data_ptr(), cols * ld_g, ld_g, rows, cols, count,
                                                   NodeModel._stream(stream)))


def gn_hessian_lanes(jac, d, g, rows: int, cols: int, count: int, unit_fastest_out: bool = True, stream=None):
    """Upper triangle of G = J^T diag(d) J, one lane per node (FP64 vector ALU).  jac (rows * cols, count) unit-fastest, d (rows, count) or
    None; g (cols * cols, count) when unit_fastest_out, else (count, cols, cols) node-major.  Entries below the diagonal are left untouched."""
    lib = load_library()
    ges, gns = (g.stride(0), 1) if unit_fastest_out else (1, cols * cols)
    _check(lib.ungar_gn_hessian_upper_lanes(jac.data_ptr(), jac.stride(0), d.data_ptr() if d is not None else None, d.stride(0) if d is not None else 0, g.data_ptr(),
                                            ges, gns, cols, rows, cols, count, NodeModel._stream(stream)))


def transpose_nodes(src, dst, count: int, elements: int, src_strides, dst_strides, stream=None):
    """dst[n * dns + e * des] = src[n * sns + e * ses]; src_strides = (node stride, element stride) in doubles, likewise dst_strides.
    Unit-fastest tensors (elements, count) have strides (1, t.stride(0)); instance-major tensors (count, elements) have (t.stride(0), 1)."""
    lib = load_library()
    _check(lib.ungar_transpose_nodes(src.data_ptr(), src_strides[0], src_strides[1], dst.data_ptr(), dst_strides[0], dst_strides[1], count, elements,
                                     NodeModel._stream(stream)))


def gn_hessian_tiles(jac, d, g, rows: int, cols: int, count: int, unit_fastest_out: bool = True, stream=None):
    """Upper triangle of G = J^T diag(d) J, one lane per (node, 7 x 7 block), Jacobian rows streamed once through LDS.  Operands as gn_hessian_lanes."""
    lib = load_library()
    ges, gns = (g.stride(0), 1) if unit_fastest_out else (1, cols * cols)
    _check(lib.ungar_gn_hessian_upper_tiles(jac.data_ptr(), jac.stride(0), d.data_ptr() if d is not None else None, d.stride(0) if d is not None else 0, g.data_ptr(),
                                            ges, gns, cols, rows, cols, count, NodeModel._stream(stream)))


def gn_hessian(jac, d, g, rows: int, cols: int, count: int, ld_j=None, ld_g=None, stream=None, upper_only=False):
    """G = J^T diag(d) J per node on the FP64 matrix cores.  jac: (count, rows, ld_j) node-major,
    d: (count, rows) or None, g: (count, cols, ld_g); all float64 CUDA tensors.  upper_only: write only
    the entries with row <= col (the rest of g is left untouched)."""
    lib = load_library()
    ld_j = cols if ld_j is None else ld_j
    ld_g = cols if ld_g is None else ld_g
    _check((lib.ungar_gn_hessian_upper if upper_only else lib.ungar_gn_hessian)(jac.data_ptr(), rows * ld_j, ld_j, d.data_ptr() if d is not None else None, rows, g.data_ptr(),
                                cols * ld_g, ld_g, rows, cols, count, NodeModel._stream(stream)))
