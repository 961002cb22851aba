"""Multi-GPU MSM: one process per GPU, points sharded per rank, partial results combined.

The MSM sum_i k_i P_i shards naturally over points: rank g holds bases[g*N/G ...) resident, runs the full
single-GPU Pippenger on its shard and produces ONE normalised point.  RCCL has no elliptic-curve reduce operator, so the
exchange step is an all-gather of the G partial points as opaque int64 words (96 B each for G1) followed by a G-term EC
sum (zkg_g1_sum).  The message is latency-bound; no bulk data crosses xGMI at prove time.
"""
import os

import numpy as np

from . import api


def shard_bounds(n_total, world, rank):
    """contiguous, balanced point ranges: the first (n_total % world) ranks get one extra point"""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def window_shard(world, rank):
    """the window-sharded variant (every GPU holds every base): rank g owns the Pippenger windows g, g + G, g + 2G, ...
    -> (first_window, window_stride) for zkg_msm_g1_windows_dev"""
    return rank, world


class _Exchange:
    """Persistent buffers of the partial-point exchange (one per (group, device, point size)): a pinned host staging pair and
    a device pair for the RCCL path, plain CPU tensors for gloo.  One all_gather_into_tensor per call, no allocation."""

    def __init__(self, words, world, group, device):
        import torch
        self.words, self.world, self.group, self.device = words, world, group, device
        on_gpu = device is not None and str(device) != "cpu"
        self.h_send = torch.empty(words, dtype=torch.int64)
        self.h_recv = torch.empty(world * words, dtype=torch.int64)
        if on_gpu:
            self.h_send = self.h_send.pin_memory(); self.h_recv = self.h_recv.pin_memory()
            self.d_send = torch.empty(words, dtype=torch.int64, device=device)
            self.d_recv = torch.empty(world * words, dtype=torch.int64, device=device)
        self.on_gpu = on_gpu

    def gather(self, partial_jac):
        import torch
        import torch.distributed as dist
        self.h_send.numpy()[:] = np.ascontiguousarray(partial_jac, dtype=np.uint64).view(np.int64).reshape(-1)
        if self.on_gpu:
            self.d_send.copy_(self.h_send, non_blocking=True)
            dist.all_gather_into_tensor(self.d_recv, self.d_send, group=self.group)
            self.h_recv.copy_(self.d_recv, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        else:
            dist.all_gather_into_tensor(self.h_recv, self.h_send, group=self.group)
        return self.h_recv.numpy().view(np.uint64).reshape(self.world, self.words)


_exchanges = {}


def _combine(partial_jac, words, summer, group, device):
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not os.environ.get("ZKG_DIST_FORCE_EXCHANGE")):
        return np.asarray(partial_jac, dtype=np.uint64).copy()       # (the env switch lets a single rank rehearse the RCCL exchange)
    key = (id(group), str(device), words)
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = _Exchange(words, dist.get_world_size(group), group, device)
    return summer(ex.gather(partial_jac))


def combine_partials_g1(partial_jac, group=None, device=None):
    """all-gather the per-rank normalised G1 partials (12 x u64) and add them; every rank returns the full result."""
    return _combine(partial_jac, 12, api.g1_sum, group, device)


def combine_partials_g2(partial_jac, group=None, device=None):
    return _combine(partial_jac, 24, api.g2_sum, group, device)
