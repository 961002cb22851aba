"""CPU suite, world_size 2 over gloo: the multi-GPU MSM exchange step (all-gather of partial points + EC sum) and the shard
arithmetic.  The per-rank partial MSM — the GPU kernel on a real node — is stood in for by the oracle here; what is under
test is zklaim_amd.dist (the collective and the host-side zkg_g1_sum combine), which bench.py --gpus N uses unchanged."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _window_subset_scalars(sc_ints, c, first, stride, r):
    """scalars reduced to the signed c-bit digits of the Pippenger windows first, first + stride, ... (what rank `first` of a
    window-sharded run weighs its bases with): sum over owned w of d_w 2^(c w) mod r, digits in (-2^(c-1), 2^(c-1)] with carry"""
    out = []
    nwin = (255 + c - 1) // c
    for k in sc_ints:
        carry = 0; acc = 0
        for w in range(nwin):
            raw = ((k >> (c * w)) & ((1 << c) - 1)) + carry
            if raw > (1 << (c - 1)):
                d = raw - (1 << c); carry = 1
            else:
                d = raw; carry = 0
            if w >= first and (w - first) % stride == 0:
                acc += d << (c * w)
        out.append(acc % r)
    return out


def _worker_windows(rank, world, port, n_total, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import zkoracle
    from util import R, arr, ints, random_fr_canonical
    from zklaim_amd import dist as zdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ks = random_fr_canonical(n_total, 21); sc = random_fr_canonical(n_total, 22)
    bases = zkoracle.g1_fixed_base(zkoracle.g1_generator(), ks)
    first, stride = zdist.window_shard(world, rank)
    mine = arr(_window_subset_scalars(ints(sc), 12, first, stride, R))        # stand-in for zkg.msm_g1_windows_dev(first, stride) on this rank's GPU
    partial = zkoracle.msm_g1(bases, mine)
    full = zdist.combine_partials_g1(partial)
    q.put((rank, bool(np.array_equal(full, zkoracle.msm_g1(bases, sc))), (first, stride)))
    dist.destroy_process_group()


def _worker(rank, world, port, n_total, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import zkoracle
    from util import random_fr_canonical
    from zklaim_amd import dist as zdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ks = random_fr_canonical(n_total, 11); sc = random_fr_canonical(n_total, 12)
    bases = zkoracle.g1_fixed_base(zkoracle.g1_generator(), ks)
    lo, hi = zdist.shard_bounds(n_total, world, rank)
    partial = zkoracle.msm_g1(bases[lo:hi], sc[lo:hi])                  # stand-in for zkg.msm_g1_dev on this rank's GPU
    full = zdist.combine_partials_g1(partial)
    expect = zkoracle.msm_g1(bases, sc)
    q.put((rank, bool(np.array_equal(full, expect)), (lo, hi)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1001, 4096])
def test_sharded_msm_combine_world2(n_total):
    from zklaim_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n_total


def test_shard_bounds_cover():
    from zklaim_amd.dist import shard_bounds
    for n in (0, 1, 7, 1 << 20, (1 << 26) + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_window_sharded_combine_world2():
    """the window-sharded variant (every rank holds every base and owns the windows g, g + G, ...): the partials,
    each already weighted by its windows' 2^(c w), add up to the full multi-exponentiation through the same exchange"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_windows, args=(r, 2, port, 300, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and [x[2] for x in res] == [(0, 2), (1, 2)]
