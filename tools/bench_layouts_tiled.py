"""Instance-major caller of a wide node model through the unit-fastest kernel IN TILES: the unit-fastest scratch of a tile stays in the
last-level cache between the node kernel and the transposer, so the only HBM traffic of the conversion is the final instance-major write.
    python tools/bench_layouts_tiled.py [model] [count]
Prints one JSON line: ms for the whole batch per tile size, one stream and two streams (double-buffered scratch)."""
import json
import sys

import torch

sys.path.insert(0, ".")
import ungar_amd  # noqa: E402
from ungar_amd.sharding import unit_fastest  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "anymal"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 81920
m = ungar_amd.NodeModel(name)
nx, nu, npar = m.nx, m.nu, m.np
ncols = nx + nu
nj = nx * ncols
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
Op = ungar_amd.Operand
xa = torch.rand((count, nx), generator=gen, device="cuda", dtype=torch.float64)
ua = torch.rand((count, nu), generator=gen, device="cuda", dtype=torch.float64) + 1.0
p = torch.rand(npar, generator=gen, device="cuda", dtype=torch.float64) + 0.5
fa = torch.empty((count, nx), dtype=torch.float64, device="cuda")
Ja = torch.empty((count, nj), dtype=torch.float64, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


out = {"model": name, "count": count, "tiles": {}}
ref = None
for tile in (1024, 2048, 4096, 8192, 16384, 32768, count):
    if tile > count:
        continue
    bufs = []
    for _ in range(2):
        xs, us, fs, Js = (unit_fastest(r, tile, torch) for r in (nx, nu, nx, nj))
        bufs.append((xs, us, fs, Js))
    st = bufs[0][0].stride(0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def tile_pass(b, n0, n, stream=None):
        xs, us, fs, Js = bufs[b]
        ungar_amd.transpose_nodes(xa[n0:], xs, n, nx, (nx, 1), (1, st), stream=stream)
        ungar_amd.transpose_nodes(ua[n0:], us, n, nu, (nu, 1), (1, st), stream=stream)
        m.dense_jacobian(n, Op.soa(xs, st, 1), Op.soa(us, st, 1), None, Op.per_instance(p, npar, shared=True), Op.soa(fs, st, 1), Op.soa(Js, st, 1), stream=stream)
        ungar_amd.transpose_nodes(fs, fa[n0:], n, nx, (1, st), (nx, 1), stream=stream)
        ungar_amd.transpose_nodes(Js, Ja[n0:], n, nj, (1, st), (nj, 1), stream=stream)

    def one_stream():
        for n0 in range(0, count, tile):
            tile_pass(0, n0, min(tile, count - n0))

    def two_streams():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i, n0 in enumerate(range(0, count, tile)):
            tile_pass(i & 1, n0, min(tile, count - n0), stream=streams[i & 1].cuda_stream)
        for s in streams:
            cur.wait_stream(s)

    one_stream()
    torch.cuda.synchronize()
    if ref is None:
        ref = Ja.clone()
    else:
        assert torch.equal(ref, Ja)
    r = {"one_stream_ms": timed(one_stream)}
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        one_stream()
        cap.synchronize()
        with torch.cuda.graph(g, stream=cap):
            one_stream()
    torch.cuda.synchronize()
    Ja.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ref, Ja)
    r["graph_ms"] = timed(g.replay)
    if tile < count:
        Ja.zero_()
        two_streams()
        torch.cuda.synchronize()
        assert torch.equal(ref, Ja)
        r["two_streams_ms"] = timed(two_streams)
    out["tiles"][tile] = r
print(json.dumps(out))
