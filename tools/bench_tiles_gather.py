#!/usr/bin/env python3
"""Tile operand -> strided operand (ungar_tiles_gather), 81 920 ANYmal nodes: to unit-fastest and to node-major blocks, next to the route it competes with
for a caller that wants node-major blocks (unit-fastest kernel + ungar_transpose_nodes)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ungar_amd  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    N, batch = 20, 4096
    count = N * batch
    m = ungar_amd.NodeModel("anymal")
    lib = ungar_amd.load_library()
    x, u, _, p = W.synth_device_inputs("anymal", count, 0, torch)
    Op = ungar_amd.Operand
    f = torch.empty((37, count), dtype=torch.float64, device="cuda")
    tiles = torch.zeros((m.tile_doubles(count),), dtype=torch.float64, device="cuda")
    soa = torch.empty((1813, count), dtype=torch.float64, device="cuda")
    aos = torch.empty((count, 1813), dtype=torch.float64, device="cuda")
    P = Op.per_instance(p, m.np, shared=True)
    args = (count, Op.soa(x, count, N), Op.soa(u, count, N), None, P, Op.soa(f, count, N))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {"nodes": count}
    out["tile_kernel_ms"] = timed(lambda: m.dense_jacobian_tiles(*args, tiles, knots=N))
    out["unit_fastest_kernel_ms"] = timed(lambda: m.dense_jacobian(*args, Op.soa(soa, count, N), knots=N))
    out["gather_to_unit_fastest_ms"] = timed(lambda: m.tiles_gather(count, tiles, Op.soa(soa, count, N), knots=N))
    out["gather_to_node_major_ms"] = timed(lambda: m.tiles_gather(count, tiles, Op.aos(aos, 1813, N), knots=N))
    out["transpose_unit_fastest_to_node_major_ms"] = timed(lambda: lib.ungar_transpose_nodes(ctypes.c_void_p(soa.data_ptr()), 1, count, ctypes.c_void_p(aos.data_ptr()), 1813, 1, count, 1813, stream))
    out["node_major_blocks_via_tiles_ms"] = out["tile_kernel_ms"] + out["gather_to_node_major_ms"]
    out["node_major_blocks_via_unit_fastest_ms"] = out["unit_fastest_kernel_ms"] + out["transpose_unit_fastest_to_node_major_ms"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
