"""GPU parity at BASELINE configs[4]'s own sizes — what ONE GPU can show of the 2^26-point MSM sharded over 8 GPUs:
  * the per-GPU share, 2^23 points: the size at which the digit sort switches back to its one-pass form (msm.hip sort_digits), compared
    with the oracle's BDLO12 restatement on ALL points;
  * the piece path above 2^23 points (msm.hip msm_shared cuts the job into 2^23-point pieces summed on the host): 2^23 + 2^20 points;
  * zkg_msm_g1_multi, the C-ABI multi-GPU entry point, with 8 shards of 2^20 points (device 0 listed eight times: same threads,
    streams and workspaces as eight devices) against the single call and the oracle.
Reference call site: the multi_exp inside r1cs_gg_ppzksnark_prover, zklaim/snark.cpp:126; partition: SURVEY.md §8(e).
The oracle runs with chunks = num_threads() (libff's MULTICORE chunking: same point as one chunk, asserted in tests/test_oracle_*)."""
import time

import numpy as np
import pytest

from gpu_util import dev_bases_g1, zkg  # noqa: F401
from util import random_fr_canonical

pytestmark = pytest.mark.gpu
SEED = 0x5A4B4C41494D0000
N_SHARE, N_EXTRA = 1 << 23, 1 << 20


@pytest.fixture(scope="module")
def share(zkg, oracle):
    """2^23 + 2^20 synthetic points (seed +6 as BASELINE.md's config 5) resident on the device, their scalars, and the oracle's MSM of
    the first 2^23 and of the remaining 2^20 (computed once for the module)"""
    import torch
    n = N_SHARE + N_EXTRA
    d_bases, bases, _ = dev_bases_g1(zkg, n, SEED + 6)
    sc = random_fr_canonical(n, SEED + 7)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    t0 = time.perf_counter()
    exp_share = oracle.msm_g1(bases[:N_SHARE], sc[:N_SHARE], oracle.BDLO12, oracle.num_threads())
    exp_extra = oracle.msm_g1(bases[N_SHARE:], sc[N_SHARE:], oracle.BDLO12, oracle.num_threads())
    print(f"oracle 2^23 + 2^20 MSMs on {oracle.num_threads()} threads: {time.perf_counter() - t0:.1f} s")
    return dict(d_bases=d_bases, bases=bases, sc=sc, d_sc=d_sc, exp_share=exp_share, exp_extra=exp_extra)


def test_share_2p23_all_points_vs_oracle(zkg, oracle, share):
    got = zkg.msm_g1_dev(share["d_bases"].data_ptr(), share["d_sc"].data_ptr(), N_SHARE)
    assert np.array_equal(got, share["exp_share"])
    # deterministic across calls (workspaces reused), and unchanged by the one-pass hint
    assert np.array_equal(zkg.msm_g1_dev(share["d_bases"].data_ptr(), share["d_sc"].data_ptr(), N_SHARE), got)


def test_piece_path_above_2p23(zkg, oracle, share):
    n = N_SHARE + N_EXTRA
    got = zkg.msm_g1_dev(share["d_bases"].data_ptr(), share["d_sc"].data_ptr(), n)
    exp = oracle.g1_sum(np.stack([share["exp_share"], share["exp_extra"]]))
    assert np.array_equal(got, exp)
    # the pieces themselves, through the same entry point
    tail = zkg.msm_g1_dev(share["d_bases"][N_SHARE:].data_ptr(), share["d_sc"][N_SHARE:].data_ptr(), N_EXTRA)
    assert np.array_equal(tail, share["exp_extra"])
    # a ragged size: one point more than a piece
    one_more = zkg.msm_g1_dev(share["d_bases"].data_ptr(), share["d_sc"].data_ptr(), N_SHARE + 1)
    last = oracle.msm_g1(share["bases"][N_SHARE:N_SHARE + 1], share["sc"][N_SHARE:N_SHARE + 1])
    assert np.array_equal(one_more, oracle.g1_sum(np.stack([share["exp_share"], last])))


def test_eight_shards_of_2p20_through_the_c_abi(zkg, oracle, share):
    """config 5's partition at one eighth of its size per shard count: 8 shards x 2^20 points = the 2^23-point job"""
    devices = [0] * 8
    zkg.init_multi(devices)
    sh = zkg.MsmShards(share["bases"][:N_SHARE], devices)
    got, parts = sh.msm(share["sc"][:N_SHARE], with_partials=True)
    assert np.array_equal(got, share["exp_share"])
    assert np.array_equal(zkg.g1_sum(parts), got) and np.array_equal(oracle.g1_sum(parts), got)
    # every shard's partial is the single-GPU MSM of its slice
    step = N_SHARE // 8
    for i in (0, 3, 7):
        lo = i * step
        single = zkg.msm_g1_dev(share["d_bases"][lo:].data_ptr(), share["d_sc"][lo:].data_ptr(), step)
        assert np.array_equal(parts[i], single)
    assert np.array_equal(sh.msm(share["sc"][:N_SHARE]), got)
    sh.free()
