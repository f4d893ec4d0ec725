"""Batch sharding of the path across GPUs: clips are independent units, so
rank r simply owns a contiguous slice of the global batch; nothing is exchanged on the data path.
The only cross-rank traffic is the metrics reduction at the end of a run (max of the elapsed time,
sums of the counters): one tiny all_reduce / all_gather over RCCL (xGMI) on GPUs, gloo in CPU tests."""
from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of rank `rank`; sizes differ by at most one clip."""
    assert 0 <= rank < world and global_batch >= 0
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def reduce_metrics(elapsed_s: float, counters: Dict[str, float], device="cpu") -> Dict[str, float]:
    """max over ranks of `elapsed_s`, sum over ranks of every counter (same keys on every rank)."""
    rank, world = world_info()
    keys = sorted(counters)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    c = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if len(keys):
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
    out = {k: float(v) for k, v in zip(keys, c.tolist())}
    out["elapsed_s"] = float(t.item())
    out["world"] = world
    return out
