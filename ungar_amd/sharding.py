"""Batch sharding across GPUs: instances are independent, so rank r of G owns the
contiguous instance range [r*B/G, (r+1)*B/G) (remainder spread over the first ranks); nothing on the
data path crosses devices.  The only collective is the reduction of the measured time (MAX) and of
the evaluation counts / checksums (SUM) -- RCCL over xGMI on the GPU box, gloo in the CPU tests."""
from __future__ import annotations


def shard_range(total_instances: int, world: int, rank: int) -> tuple[int, int]:
    """[begin, end) of the instances owned by `rank`."""
    if not (0 <= rank < world) or total_instances < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total_instances, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


# Tile size of the unit-fastest device layout.  Within one operand the element stride is (nodes of the operand) x 8 bytes, and
# a wavefront of the ANYmal kernel keeps 1813 store streams open, one per Jacobian entry: at 65 536 instances x 20 knots in ONE
# operand the streams are 10.5 MB apart (1813 distinct pages per wavefront) and the kernel drops from 3.2 to 4.1 ns per node;
# tiles of 4096-8192 instances (element stride <= 1.3 MB) keep the full rate, smaller tiles lose it again to launch tails
# (tools/bench_tiles.py, profiles/archive/r02b_tile_sweep.json).  A rank therefore stores its shard as [tile][element][node of tile].
DEFAULT_TILE_INSTANCES = 8192


# Element stride of a unit-fastest operand.  With the natural stride (= nodes of the operand) the batch sizes of interest put
# consecutive elements a multiple of 2^17 bytes apart (4096 instances x 20 knots x 8 B = 5 x 2^17), so all 1813 store streams
# of a wavefront -- and all Jacobian rows a Gauss-Newton workgroup reads -- fall on the SAME memory channel.  Padding the
# stride by a few 128-byte segments rotates them over the channels: ANYmal node Jacobians 0.30-0.31 -> 0.25 ms per 81 920 nodes,
# the Gauss-Newton kernel 0.53 -> 0.45 ms (tools/bench_stride_pad.py, profiles/archive/r02e_stride_pad.json).  Any pad >= 32 nodes
# (256 B) recovers the node kernel; the contraction additionally prefers an odd number of segments.
def padded_stride(nodes: int) -> int:
    """Element stride (in doubles) for a unit-fastest operand of `nodes` nodes: a multiple of 16 nodes (128-byte segments stay
    aligned) whose segment count is 3 mod 4, at least 32 nodes larger than a power-of-two-ish `nodes`."""
    if nodes < 0:
        raise ValueError("bad node count")
    segments = -(-nodes // 16)
    return 16 * ((segments + 2) | 3)


def unit_fastest(elements: int, nodes: int, torch, device="cuda"):
    """Uninitialised (elements, nodes) float64 view with the padded element stride; pass `t.stride(0)` as the operand's element stride."""
    return torch.empty((elements, padded_stride(nodes)), dtype=torch.float64, device=device)[:, :nodes]


def tile_ranges(instances: int, tile: int = DEFAULT_TILE_INSTANCES) -> list[tuple[int, int]]:
    """[begin, end) instance ranges of the tiles of a shard: as many equal tiles of at most `tile` instances as needed
    (sizes differ by at most one), so that no launch is much smaller than the others."""
    if instances < 0 or tile < 1:
        raise ValueError("bad tiling request")
    if instances == 0:
        return []
    n = -(-instances // tile)
    return [shard_range(instances, n, t) for t in range(n)]


def reduce_timing(elapsed_s: float, evals: int, dist=None, device=None) -> tuple[float, int]:
    """(max elapsed over ranks, total evals over ranks).  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized():  # (an initialised group of ONE rank still runs the collectives: the single-GPU RCCL test)
        return elapsed_s, evals
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([evals], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def reduce_sums(values, dist=None, device=None) -> list[float]:
    """Element-wise SUM over ranks of a short list of floats (evaluation counts, output checksums: SURVEY.md §8(e))."""
    if dist is None or not dist.is_initialized():  # (an initialised group of ONE rank still runs the collectives: the single-GPU RCCL test)
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
