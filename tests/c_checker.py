"""TEST INFRASTRUCTURE: every node of a device launch against the oracle's generated C (oracle/_gen/*_cg.c, compiled by
__graft_entry__.build() into the "portable" oracle library; pinned against the independent torch oracle's golden vectors by
tests/test_codegen_c.py).  Unit-fastest device tensors (elements, nodes) are brought to the host chunk by chunk."""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest


def compare_launch_with_c_checker(name, x, u, p, f, J, count, chunk=1 << 19):
    """x (nx, count), u (nu, count), f (nx, count), J (nx * (nx + nu), count): device tensors; p: shared parameter vector.  Values to
    1e-10 of their scale, Jacobian entries to 1e-9 of each block's scale, structural zeros exact."""
    from oracle import build_oracle
    path = build_oracle.lib_path("portable")
    if not os.path.exists(path):
        pytest.skip("oracle C library not built: run __graft_entry__.build()")
    clib = ctypes.CDLL(path)
    nx, nu = x.shape[0], u.shape[0]
    ncols = nx + nu
    nnz = ctypes.c_int.in_dll(clib, f"{name}_jac_nnz").value
    rows = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_row")).astype(np.int64)
    cols = np.ctypeslib.as_array((ctypes.c_int * nnz).in_dll(clib, f"{name}_jac_col")).astype(np.int64)
    flat = rows * ncols + cols
    off = np.ones(nx * ncols, dtype=bool)
    off[flat] = False
    dp = ctypes.POINTER(ctypes.c_double)
    loop = getattr(clib, f"{name}_sparse_jacobian_batch_shared")  # C loop over a range of nodes: one foreign call per thread and chunk
    loop.argtypes = [dp] * 6 + [ctypes.c_long, ctypes.c_long]
    loop.restype = None
    ph, w0 = np.ascontiguousarray(p.cpu().numpy()), np.zeros(1)
    workers = min(16, os.cpu_count() or 1)
    for lo in range(0, count, chunk):
        hi = min(count, lo + chunk)
        m = hi - lo
        # node-major on the DEVICE (a 1.2 GB block transposed by numpy on the host cost more than the C checker itself)
        xh, uh = x[:, lo:hi].t().contiguous().cpu().numpy(), u[:, lo:hi].t().contiguous().cpu().numpy()
        fh, Jh = f[:, lo:hi].t().contiguous().cpu().numpy(), J[:, lo:hi].t().contiguous().cpu().numpy()
        rf, rj = np.empty((m, nx)), np.empty((m, nnz))

        def part(a, b):  # ctypes releases the GIL inside the call
            loop(xh.ctypes.data_as(dp), uh.ctypes.data_as(dp), w0.ctypes.data_as(dp), ph.ctypes.data_as(dp), rf.ctypes.data_as(dp), rj.ctypes.data_as(dp), a, b)

        step = (m + workers - 1) // workers
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(lambda k: part(k * step, min(m, (k + 1) * step)), range(workers)))
        assert np.abs(fh - rf).max() <= 1e-10 * max(1.0, np.abs(rf).max()), (name, lo)
        assert (np.abs(Jh[:, flat] - rj) / np.abs(rj).max(axis=1, keepdims=True)).max() <= 1e-9, (name, lo)
        assert not Jh[:, off].any(), (name, lo)  # structural zeros are exact zeros in every block
