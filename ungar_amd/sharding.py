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


def reduce_timing(elapsed_s: float, evals: int, dist=None, device=None) -> tuple[float, int]:
    """(max elapsed over ranks, total evals over ranks).  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_s, evals
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([evals], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def reduce_sums(values, dist=None, device=None) -> list[float]:
    """Element-wise SUM over ranks of a short list of floats (evaluation counts, output checksums: SURVEY.md §8(e))."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
