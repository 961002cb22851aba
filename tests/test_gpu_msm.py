"""GPU parity: zkg_msm_g1 / zkg_msm_g2 (HIP Pippenger) vs golden vectors and the oracle's BDLO12 restatement.  Bit-exact
on the normalised result."""
import numpy as np
import pytest

from gpu_util import dev_bases_g1, zkg  # noqa: F401
from util import R, arr, g1_aff, g1_jac_expected, g2_aff, g2_jac_expected, golden, h, limbs, random_fr_canonical

pytestmark = pytest.mark.gpu


def test_msm_golden(zkg):
    g = golden("msm.json")
    for c in g["g1"]:
        n = len(c["bases"])
        bases = np.array([g1_aff(b) for b in c["bases"]], np.uint64).reshape(n, 8)
        sc = arr([h(s) for s in c["scalars"]]) if n else np.zeros((0, 4), np.uint64)
        assert np.array_equal(zkg.msm_g1(bases, sc), g1_jac_expected(c["result"])), c["tag"]
    for c in g["g2"]:
        n = len(c["bases"])
        bases = np.array([g2_aff(b) for b in c["bases"]], np.uint64).reshape(n, 16)
        sc = arr([h(s) for s in c["scalars"]])
        assert np.array_equal(zkg.msm_g2(bases, sc), g2_jac_expected(c["result"])), c["tag"]


def test_fixed_base_matches_oracle(zkg, oracle):
    _, bases, ks = dev_bases_g1(zkg, 300, 0x1234)
    exp = oracle.g1_fixed_base(oracle.g1_generator(), ks)
    assert np.array_equal(bases, exp)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4096, 20000])
def test_msm_g1_vs_oracle(zkg, oracle, n):
    _, bases, _ = dev_bases_g1(zkg, n, 0x5A4B4C41494D0001 + n)
    sc = random_fr_canonical(n, 0x5A4B4C41494D0002 + n)
    assert np.array_equal(zkg.msm_g1(bases, sc), oracle.msm_g1(bases, sc))


def test_msm_g1_bit_scalars_and_edges(zkg, oracle):
    """zklaim-shaped scalars: mostly 0/1 (multi_exp_with_mixed_addition's shortcut), plus r-1, duplicates, infinity bases."""
    n = 5000
    _, bases, _ = dev_bases_g1(zkg, n, 77)
    sc = random_fr_canonical(n, 78)
    rng = np.random.default_rng(5)
    kind = rng.integers(0, 100, n)
    sc[kind < 50] = 0
    sc[(kind >= 50) & (kind < 95), :] = np.array([1, 0, 0, 0], np.uint64)
    sc[4000] = limbs(R - 1); sc[4001] = limbs(R - 1)
    bases[4001] = bases[4000]                 # duplicate base, same scalar -> doubling path inside a bucket
    bases[4002] = 0                           # base at infinity
    bases[10] = bases[11]; sc[10] = sc[11] = np.array([1, 0, 0, 0], np.uint64)      # doubling inside the ones-sum
    assert np.array_equal(zkg.msm_g1(bases, sc), oracle.msm_g1(bases, sc, oracle.MIXED))


def test_msm_g2_vs_oracle(zkg, oracle):
    n = 600
    ks = random_fr_canonical(n, 99)
    bases = oracle.g2_fixed_base(oracle.g2_generator(), ks)
    sc = random_fr_canonical(n, 100)
    sc[:100] = 0; sc[100:300] = np.array([1, 0, 0, 0], np.uint64)
    assert np.array_equal(zkg.msm_g2(bases, sc), oracle.msm_g2(bases, sc, oracle.MIXED))


def test_msm_full_size_properties(zkg, oracle):
    """BASELINE config 2 size (2^20 points): linearity in the scalars and a split/sum identity, checked through the
    device-pointer entry points with bases resident; a 2^16 prefix is checked against the oracle directly."""
    import torch
    n = 1 << 20
    d_bases, bases, _ = dev_bases_g1(zkg, n, 0x5A4B4C41494D0001)
    sc = random_fr_canonical(n, 0x5A4B4C41494D0002)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    full = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n)
    # split: MSM(all) == MSM(first half) + MSM(second half)
    half = n // 2
    p0 = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), half)
    p1 = zkg.msm_g1_dev(d_bases[half:].data_ptr(), d_sc[half:].data_ptr(), n - half)
    assert np.array_equal(zkg.g1_sum(np.concatenate([p0, p1])), full)
    assert np.array_equal(oracle.g1_sum(np.concatenate([p0, p1])), full)
    # scalars all equal to s: sum s*P_i == s * (sum P_i)
    ones = np.zeros((n, 4), np.uint64); ones[:, 0] = 1
    d_ones = torch.from_numpy(ones.view(np.int64)).cuda()
    total = zkg.msm_g1_dev(d_bases.data_ptr(), d_ones.data_ptr(), n)
    k = 1 << 16
    assert np.array_equal(zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), k), oracle.msm_g1(bases[:k], sc[:k]))
    s = sc[0]; k2 = 2048
    same = np.tile(s, (k2, 1)); d_same = torch.from_numpy(np.ascontiguousarray(same).view(np.int64)).cuda()
    psum = zkg.msm_g1_dev(d_bases.data_ptr(), d_ones.data_ptr(), k2)
    lhs = zkg.msm_g1_dev(d_bases.data_ptr(), d_same.data_ptr(), k2)
    assert np.array_equal(lhs, oracle.g1_scalar_mul(psum[:8], s))
    assert total.any()


def test_msm_adversarial_scalar_distributions(zkg, oracle):
    """distributions that defeat a naive lane-per-bucket kernel: every scalar equal (one bucket per window holds everything),
    only two distinct scalars, tiny scalars (high windows empty), scalars of the form 2^k (single non-zero digit), all r-1
    (every signed digit negative or carrying)."""
    n = 3000
    _, bases, _ = dev_bases_g1(zkg, n, 4242)
    rng = np.random.default_rng(11)
    same = np.tile(random_fr_canonical(1, 5), (n, 1))
    two = random_fr_canonical(2, 6)[rng.integers(0, 2, n)]
    tiny = np.zeros((n, 4), np.uint64); tiny[:, 0] = rng.integers(0, 1000, n).astype(np.uint64)
    pow2 = np.zeros((n, 4), np.uint64)
    for i, k in enumerate(rng.integers(0, 253, n)):
        pow2[i, k // 64] = np.uint64(1) << np.uint64(k % 64)
    rm1 = np.tile(np.array(limbs(R - 1), np.uint64), (n, 1))
    for name, sc in (("same", same), ("two", two), ("tiny", tiny), ("pow2", pow2), ("r-1", rm1)):
        sc = np.ascontiguousarray(sc)
        assert np.array_equal(zkg.msm_g1(bases, sc), oracle.msm_g1(bases, sc, oracle.MIXED)), name


def test_msm_g2_larger_and_edges(zkg, oracle):
    n = 3000
    ks = random_fr_canonical(n, 199)
    bases = oracle.g2_fixed_base(oracle.g2_generator(), ks)
    sc = random_fr_canonical(n, 200)
    sc[5] = limbs(R - 1); bases[7] = 0; bases[9] = bases[8]; sc[9] = sc[8]
    assert np.array_equal(zkg.msm_g2(bases, sc), oracle.msm_g2(bases, sc))
    assert np.array_equal(zkg.msm_g2(bases[:0], sc[:0]), g2_jac_expected(None))          # empty
    assert np.array_equal(zkg.msm_g1(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64)), g1_jac_expected(None))


def test_size_limits_are_refused_not_truncated(zkg):
    """maximum sizes: the sorted index list is 32-bit (n * windows < 2^32) and Fr has 2-adicity 28; anything larger is an error
    before a single byte is touched (the device pointers below are never dereferenced)"""
    import torch
    d = torch.zeros(64, dtype=torch.int64, device="cuda")
    with pytest.raises(zkg.ZkgError):
        zkg.msm_g1_dev(d.data_ptr(), d.data_ptr(), 1 << 28)           # 2^28 points x 16 windows = 2^32 entries
    with pytest.raises(zkg.ZkgError):
        zkg.msm_g1_dev(d.data_ptr(), d.data_ptr(), 1 << 31)
    with pytest.raises(zkg.ZkgError):
        zkg.ntt_dev(d.data_ptr(), 29)
    # the largest windows-per-point product that is accepted is exercised by bench.py at 2^20..2^23; here: 2^24 entries per window
    # would need 1.6 GB of bases, so the positive side of the limit is covered by test_msm_full_size_properties


@pytest.mark.parametrize("world", [2, 3, 8, 20])
def test_window_sharded_partials_add_up(zkg, oracle, world):
    """the multi-GPU variant in which rank g owns the Pippenger windows g, g + G, ...: the G partials (each weighted by its windows'
    2^(c w)) sum to the plain MSM, for window counts that divide, do not divide, and are smaller than G"""
    n = 5000
    d_b, bases, _ = dev_bases_g1(zkg, n, 77)
    sc = random_fr_canonical(n, 78)
    import torch
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    full = zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n)
    parts = np.stack([zkg.msm_g1_windows_dev(d_b.data_ptr(), d_sc.data_ptr(), n, g, world) for g in range(world)])
    assert np.array_equal(zkg.g1_sum(parts), full)
    assert np.array_equal(full, oracle.msm_g1(bases, sc))
    # the default entry point is unaffected by a previous subset call
    assert np.array_equal(zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n), full)


def test_mostly_bits_flag_does_not_change_the_result(zkg, oracle):
    """ZKG_SCALARS_MOSTLY_BITS only selects the one-pass digit sort (libff's multi_exp_with_mixed_addition case); witness-like scalars
    (bits plus a few full-size values) and uniform ones give the same point either way, equal to the oracle"""
    import torch
    n = 70000                                              # c = 16, two-pass sort by default
    _mostly_bits_case(zkg, oracle, n)


def test_two_pass_sort_giant_bin(zkg, oracle):
    """200 000 witness-like scalars without the hint: ~97 000 ones share one bucket, so one coarse bin holds > 8 x 8192 entries and
    goes through k_rx_fine_big's wavefront-aggregated path; all-equal scalars put EVERY entry of a window in one bucket"""
    import torch
    _mostly_bits_case(zkg, oracle, 200000)
    n = 150000
    d_b, bases, _ = dev_bases_g1(zkg, n, 93)
    same = np.tile(random_fr_canonical(1, 94), (n, 1))
    d_sc = torch.from_numpy(same.view(np.int64)).cuda()
    got = zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n)
    assert np.array_equal(got, zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n, mostly_bits=True))
    assert np.array_equal(got, oracle.msm_g1(bases, same, oracle.BDLO12, oracle.num_threads()))


def _mostly_bits_case(zkg, oracle, n):
    import torch
    d_b, bases, _ = dev_bases_g1(zkg, n, 91)
    uni = random_fr_canonical(n, 92)
    wit = uni.copy()
    rng = np.random.default_rng(5)
    bits = rng.random(n) < 0.97
    wit[bits] = 0; wit[bits, 0] = rng.integers(0, 2, int(bits.sum())).astype(np.uint64)
    for sc in (uni, wit):
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        a = zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n)
        b = zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n, mostly_bits=True)
        assert np.array_equal(a, b) and np.array_equal(a, oracle.msm_g1(bases, sc, oracle.MIXED))


@pytest.mark.parametrize("n,devices", [(10007, [0, 0, 0]), (2, [0, 0, 0]), (40000, [0, 0]), (0, [0, 0])])
def test_multi_device_shards_in_one_process(zkg, oracle, n, devices):
    """zkg_msm_g1_multi (the C-ABI multi-GPU entry point: bases sharded by points, one host thread per shard, partials added on the
    host).  A single-GPU box rehearses it with one device listed several times: same threads, streams and workspaces, same result as
    the single-GPU call and the oracle; shards of unequal and of zero size included."""
    zkg.init_multi(devices)
    _, bases, _ = dev_bases_g1(zkg, max(n, 1), 4000 + n)
    bases = bases[:n]
    sc = random_fr_canonical(max(n, 1), 4001 + n)[:n]
    sh = zkg.MsmShards(bases, devices)
    got, parts = sh.msm(sc, with_partials=True)
    exp = oracle.msm_g1(bases, sc) if n else g1_jac_expected(None)
    assert np.array_equal(got, exp)
    assert np.array_equal(got, zkg.msm_g1(bases, sc))
    assert np.array_equal(zkg.g1_sum(parts), got)                       # the exchange step: ndev normalised partials, summed
    bounds = [n * i // len(devices) for i in range(len(devices) + 1)]
    for i in range(len(devices)):
        lo, hi = bounds[i], bounds[i + 1]
        assert np.array_equal(parts[i], oracle.msm_g1(bases[lo:hi], sc[lo:hi]) if hi > lo else g1_jac_expected(None))
    assert np.array_equal(sh.msm(sc), got)                              # workspaces are reused across calls
    sh.free()


@pytest.mark.parametrize("n", [1, 300, 5000, 70000])
def test_resident_bases_with_window_tables_equal_the_plain_msm(zkg, oracle, n):
    """zkg_msm_g1_bases_upload / zkg_msm_g1_resident (per-window tables of fixed bases: one bucket set, one reduction, no host doublings;
    from 49152 points on also windows sharing rows of buckets): the same point as zkg_msm_g1_dev and the oracle, bit for bit — uniform
    scalars, then mostly-bit scalars with r - 1, a duplicated base and a base at infinity; two calls on one handle; a wrong n is refused."""
    import torch
    d_bases, bases, _ = dev_bases_g1(zkg, n, 0xBA5E5 + n)
    if n >= 300:
        bases[7] = bases[8]; bases[9] = 0                               # a duplicated base, a base at infinity
        d_bases = torch.from_numpy(bases.view(np.int64)).cuda()
    h = zkg.ResidentBases(d_bases.data_ptr(), n)
    sc = random_fr_canonical(n, 0xBA5E6 + n)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    got = h.msm(d_sc.data_ptr())
    assert np.array_equal(got, zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n))
    if n <= 5000:
        assert np.array_equal(got, oracle.msm_g1(bases, sc))
    sc2 = sc.copy(); rng = np.random.default_rng(n); kind = rng.integers(0, 100, n)
    sc2[kind < 45] = 0; sc2[(kind >= 45) & (kind < 90), :] = np.array([1, 0, 0, 0], np.uint64); sc2[0] = limbs(R - 1)
    if n >= 300:
        sc2[7] = sc2[8] = limbs(R - 1)
    d_sc2 = torch.from_numpy(sc2.view(np.int64)).cuda()
    assert np.array_equal(h.msm(d_sc2.data_ptr()), zkg.msm_g1_dev(d_bases.data_ptr(), d_sc2.data_ptr(), n))
    assert np.array_equal(h.msm(d_sc.data_ptr()), got)                 # the handle's job and tables are reusable
    # ordering: scalars still being written on another stream when the call starts — the job waits for that stream, not for the host
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        junk = torch.empty(32 << 20, dtype=torch.int64, device="cuda")
        for _ in range(4):
            junk.add_(1)                                               # ~1 GB of traffic queued ahead of the copy
        d_late = torch.zeros_like(d_sc2)
        d_late.copy_(d_sc2, non_blocking=True)
    assert np.array_equal(h.msm(d_late.data_ptr(), stream=side.cuda_stream), zkg.msm_g1_dev(d_bases.data_ptr(), d_sc2.data_ptr(), n))
    side.synchronize(); del junk
    h.n = n + 1
    with pytest.raises(zkg.ZkgError):
        h.msm(d_sc.data_ptr())
    h.n = n
    h.free()


def _with_env(name, value, fn):
    import os
    old = os.environ.get(name)
    os.environ[name] = str(value)
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old


@pytest.mark.parametrize("levels", [1, 2, 3, 4])
def test_batched_affine_levels_match_the_plain_accumulation(zkg, oracle, levels):
    """ZKG_ACCUM_BA = levels (csrc/msm_ba.inc): the bucket lists summed as a pairwise tree of batched AFFINE additions (Montgomery's trick
    shared by a workgroup, safegcd inversion) before the XYZZ accumulation — the same point as libff's multi_exp_inner<BDLO12> loop
    (snark.cpp:126), bit for bit: uniform scalars against the oracle, and every exceptional pair inside the batch — equal points
    (tangent), P + (-P), infinity as an operand, infinity as a result feeding the next level."""
    from util import Q, from_limbs
    n = 20000
    _, bases, _ = dev_bases_g1(zkg, n, 0xBA0 + levels)
    sc = random_fr_canonical(n, 0xBA1 + levels)
    exp = oracle.msm_g1(bases, sc)
    assert np.array_equal(_with_env("ZKG_ACCUM_BA", levels, lambda: zkg.msm_g1(bases, sc)), exp)
    assert np.array_equal(_with_env("ZKG_BA_K", 6, lambda: _with_env("ZKG_ACCUM_BA", levels, lambda: zkg.msm_g1(bases, sc))), exp)
    # groups of four equal scalars land in one bucket; their bases are chosen so that neighbours in the bucket's list are equal, opposite
    # or at infinity (the order inside a bucket is the sort's, so several pairings of each group occur)
    n = 8192
    _, bases, _ = dev_bases_g1(zkg, n, 0xBA2)
    sc = np.repeat(random_fr_canonical(n // 4, 0xBA3 + levels), 4, axis=0)

    def negated(b):
        out = b.copy()
        out[4:] = limbs((Q - from_limbs(b[4:])) % Q)
        return out
    for g in range(n // 4):
        p, q_ = bases[4 * g].copy(), bases[4 * g + 1].copy()
        kind = g % 4
        if kind == 0:
            bases[4 * g: 4 * g + 4] = p
        elif kind == 1:
            bases[4 * g + 1] = negated(p); bases[4 * g + 2] = q_; bases[4 * g + 3] = 0
        elif kind == 2:
            bases[4 * g + 1] = q_; bases[4 * g + 2] = negated(q_); bases[4 * g + 3] = negated(p)
        else:
            bases[4 * g] = 0; bases[4 * g + 1] = 0; bases[4 * g + 2] = p; bases[4 * g + 3] = p
    exp = oracle.msm_g1(bases, sc)
    assert np.array_equal(zkg.msm_g1(bases, sc), exp)
    assert np.array_equal(_with_env("ZKG_ACCUM_BA", levels, lambda: zkg.msm_g1(bases, sc)), exp)


def test_batched_affine_levels_at_2p18(zkg):
    """the same switch at a size where the levels are most of the work (2^18 points, 16-bit windows): equal to the default path"""
    import torch
    n = 1 << 18
    d_bases, _, _ = dev_bases_g1(zkg, n, 0xBA9)
    d_sc = torch.from_numpy(random_fr_canonical(n, 0xBAA).view(np.int64)).cuda()
    exp = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n)
    for levels in (2, 3):
        assert np.array_equal(_with_env("ZKG_ACCUM_BA", levels, lambda: zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n)), exp)


@pytest.mark.parametrize("n", [3000, (1 << 19) + 777, 1 << 20])
def test_host_scalars_pieces_equal_the_resident_msm(zkg, oracle, n):
    """zkg_msm_g1_host_scalars (bases resident, scalars uploaded from pinned host memory in pieces under the work; every piece accumulates
    into the same 29-bit buckets, one reduction at the end): the same point as zkg_msm_g1_dev on the uploaded vector, for uniform scalars
    and for a vector whose digits pile up in a few buckets (heavy buckets continued from piece to piece), pageable memory included."""
    import torch
    d_bases, bases, _ = dev_bases_g1(zkg, n, 0x4051 + n)
    sc = random_fr_canonical(n, 0x4052 + n)
    h_sc = torch.from_numpy(sc.view(np.int64)).pin_memory()
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    exp = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n)
    if n <= 3000:
        assert np.array_equal(exp, oracle.msm_g1(bases, sc))
    assert np.array_equal(zkg.msm_g1_host_scalars(d_bases.data_ptr(), h_sc.data_ptr(), n), exp)
    assert np.array_equal(zkg.msm_g1_host_scalars(d_bases.data_ptr(), sc.ctypes.data, n), exp)          # pageable host memory
    if n == (1 << 19) + 777:                                                     # scalars handed over in Montgomery form (a witness vector as libff keeps it)
        sc_m = np.ascontiguousarray(zkg.field_op(1, 4, sc))
        h_m = torch.from_numpy(sc_m.view(np.int64)).pin_memory()
        assert np.array_equal(zkg.msm_g1_host_scalars(d_bases.data_ptr(), h_m.data_ptr(), n, scalars_mont=True), exp)
    # a third of the scalars equal (one giant bucket per window, in every piece), some zero, r - 1, a duplicated base and a base at infinity
    sc2 = sc.copy(); rng = np.random.default_rng(n); kind = rng.integers(0, 100, n)
    sc2[kind < 33] = sc2[0]; sc2[(kind >= 33) & (kind < 40)] = 0; sc2[n - 1] = limbs(R - 1); sc2[n // 2] = limbs(R - 1)
    bases[n // 2] = bases[n - 1]; bases[5] = 0
    d_bases2 = torch.from_numpy(bases.view(np.int64)).cuda()
    h_sc2 = torch.from_numpy(sc2.view(np.int64)).pin_memory()
    d_sc2 = torch.from_numpy(sc2.view(np.int64)).cuda()
    exp2 = zkg.msm_g1_dev(d_bases2.data_ptr(), d_sc2.data_ptr(), n)
    assert np.array_equal(zkg.msm_g1_host_scalars(d_bases2.data_ptr(), h_sc2.data_ptr(), n), exp2)
    assert np.array_equal(zkg.msm_g1_host_scalars(d_bases.data_ptr(), h_sc.data_ptr(), n), exp)       # and the job is clean afterwards


def test_multi_device_exchange_over_rccl(zkg, oracle):
    """ZKG_MULTI_RCCL=1: zkg_msm_g1_multi all-gathers the shards' 96-byte partials with RCCL (ncclCommInitAll in ONE process, the library
    opened at run time) before the sum — the collective form of the exchange BASELINE.json's north star names.  One shard per DISTINCT
    device: as many as the box has (one here), same point as the oracle, and the call counter moves; a handle that lists a device twice
    keeps the host exchange (RCCL refuses duplicate devices) and still returns the same point."""
    import torch
    n = 3000
    _, bases, _ = dev_bases_g1(zkg, n, 0x7CC1)
    sc = random_fr_canonical(n, 0x7CC2)
    exp = oracle.msm_g1(bases, sc)
    ndev = min(torch.cuda.device_count(), 8)
    before = zkg.lib().zkg_multi_rccl_calls()
    sh = zkg.MsmShards(bases, list(range(ndev)))
    got = _with_env("ZKG_MULTI_RCCL", 1, lambda: sh.msm(sc))
    got2 = _with_env("ZKG_MULTI_RCCL", 1, lambda: sh.msm(sc))
    sh.free()
    assert np.array_equal(got, exp) and np.array_equal(got2, exp)
    assert zkg.lib().zkg_multi_rccl_calls() == before + 2, "the exchange did not go through RCCL"
    dup = zkg.MsmShards(bases, [0, 0])
    got3 = _with_env("ZKG_MULTI_RCCL", 1, lambda: dup.msm(sc))
    dup.free()
    assert np.array_equal(got3, exp) and zkg.lib().zkg_multi_rccl_calls() == before + 2


def test_sort_workgroup_forms_and_wave_counts_agree(zkg):
    """The digit sort's two wide kernels exist as 1024- and 512-thread workgroups (the second runs beside an accumulation, msm.hip k_rx_scatter)
    and the 29-bit accumulation as two capped or three wavefronts per SIMD; the library picks by context.  Forced each way in a process of its
    own (the switches are read once), the resident multi-exponentiation returns the same point as this process's default path, for a size
    that is and one that is not a multiple of four (16-byte and 4-byte digit loads), with a third of the scalars equal (an oversized bin)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import zklaim_amd as zkg
from gpu_util import dev_bases_g1
from util import random_fr_canonical
zkg.init(0)
for n in (70000, 65537 + 4096):
    d_bases, bases, _ = dev_bases_g1(zkg, n, 0x51A0 + n)
    sc = random_fr_canonical(n, 0x51A1 + n); sc[np.random.default_rng(n).integers(0, 3, n) == 0] = sc[0]
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    print(zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n).tobytes().hex())
'''
    outs = []
    for env in ({}, {"ZKG_SORT_WG": "512"}, {"ZKG_SORT_WG": "1024", "ZKG_ACC29_WAVES": "2"}, {"ZKG_ACC29_WAVES": "3", "ZKG_SORT_WG": "512"}):
        r = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-2:])
    assert all(len(o) == 2 for o in outs)
    assert outs[1] == outs[0] and outs[2] == outs[0] and outs[3] == outs[0]
