"""Multi-GPU MSM: one process per GPU, points sharded per rank, partial results combined.

The MSM sum_i k_i P_i shards naturally over points: rank g holds bases[g*N/G ...) resident, runs the full
single-GPU Pippenger on its shard and produces ONE normalised point.  RCCL has no elliptic-curve reduce operator, so the
exchange step is an all-gather of the G partial points as opaque int64 words (96 B each for G1) followed by a G-term EC
sum (zkg_g1_sum).  The message is latency-bound; no bulk data crosses xGMI at prove time.
"""
import numpy as np

from . import api


def shard_bounds(n_total, world, rank):
    """contiguous, balanced point ranges: the first (n_total % world) ranks get one extra point"""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def combine_partials_g1(partial_jac, group=None, device=None):
    """all-gather the per-rank normalised G1 partials (12 x u64) and add them; every rank returns the full result."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return np.asarray(partial_jac, dtype=np.uint64).copy()
    world = dist.get_world_size(group)
    mine = torch.from_numpy(np.ascontiguousarray(partial_jac, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    pts = torch.stack(gathered).cpu().numpy().view(np.uint64)
    return api.g1_sum(pts)


def combine_partials_g2(partial_jac, group=None, device=None):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(partial_jac, dtype=np.uint64).copy()
    world = dist.get_world_size(group)
    mine = torch.from_numpy(np.ascontiguousarray(partial_jac, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    pts = torch.stack(gathered).cpu().numpy().view(np.uint64)
    return api.g2_sum(pts)
