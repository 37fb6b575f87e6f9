#!/usr/bin/env python3
"""Tile-size sweep of the unit-fastest layout: the batch is stored as [tile][element][nodes of the tile] and evaluated with one
launch per tile (same stream).  Motivation: at 1.31 M nodes in ONE unit-fastest operand the element stride is 10.5 MB and the
1813 store streams of a wavefront fall on 1813 different pages; r02 measured 4.23 ns/node there against 3.29 ns/node at 81 920
nodes (element stride 655 KB).  usage: bench_tiles.py [total_instances]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd import workloads as W  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = 20
m = ungar_amd.NodeModel("anymal")
Op = ungar_amd.Operand
count = total * N
x, u, _, p = W.synth_device_inputs("anymal", count, 0, torch)
out = {}
for tile_inst in (256, 512, 1024, 2048, 4096, 8192, 16384, 65536):
    if tile_inst > total:
        continue
    tn = tile_inst * N
    tiles = total // tile_inst
    xs = [x[:, t * tn:(t + 1) * tn].contiguous() for t in range(tiles)]
    us = [u[:, t * tn:(t + 1) * tn].contiguous() for t in range(tiles)]
    f = torch.empty((tiles, 37, tn), dtype=torch.float64, device="cuda")
    J = torch.empty((tiles, 1813, tn), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        for t in range(tiles):
            m.dense_jacobian(tn, Op.soa(xs[t], tn, N), Op.soa(us[t], tn, N), None, Op.per_instance(p, 1, shared=True), Op.soa(f[t], tn, N), Op.soa(J[t], tn, N),
                             knots=N, stream=stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = max(5, int(2e7 / count))
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[tile_inst] = {"ms": dt * 1e3, "ns_per_node": dt / count * 1e9, "frac_hbm": count * 15192 / dt / 8e12, "checksum": float(J.sum() + f.sum())}
    print(tile_inst, out[tile_inst], flush=True)
    del f, J, xs, us
    torch.cuda.empty_cache()
print(json.dumps({"total_instances": total, "tiles": out}))
