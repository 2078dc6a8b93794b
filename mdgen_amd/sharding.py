"""Batch sharding across ranks for the sampler: independent peptides / rollouts are
split into contiguous shards, one process per GPU, with NO data-path collective -- the reference's own
pattern (`tps_inference.py:160-161` `--chunk_idx/--n_chunks`).  `torch.distributed` (RCCL on GPUs, gloo in
the CPU tests) is used only to synchronise and to reduce the timing."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `n_items` for `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_list(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_range(len(items), rank, world)
    return list(items[lo:hi])


def max_over_ranks(seconds: float, dist=None, device=None, force: bool = False) -> float:
    """Wall time of the slowest rank (what the whole job waited for).  force: run the collective in a group of one rank too
    (bench.py --dist-selftest)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(seconds)
    import torch
    if device is None and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None, force: bool = False) -> float:
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return float(value)
    import torch
    if device is None and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, dist=None, device=None, force: bool = False) -> List[float]:
    """`value` of every rank, in rank order (per-rank timings of a weak-scaling run: stragglers show up here)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return [float(value)]
    import torch
    if device is None and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
