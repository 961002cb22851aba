"""GPU known-answer test of the device field arithmetic: the generated v_mad_u64_u32 / v_addc streams
of zklaim_amd/csrc/mont_asm.inc, run on the GPU through zkg_field_op, against the definition-level vectors of tests/golden/field.json
(plain Python integers) and, on random operands including the edges of the lazy [0, 2p) range, against the CPU oracle."""
import numpy as np
import pytest

from gpu_util import zkg  # noqa: F401
from util import Q, R, arr, golden, h, ints, random_fr_canonical

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,field,p", [("fq", 0, Q), ("fr", 1, R)])
def test_field_ops_golden(zkg, name, field, p):
    cases = golden("field.json")["cases"][name]
    a = arr([h(c["a"]) for c in cases], p); b = arr([h(c["b"]) for c in cases], p)
    assert ints(zkg.field_op(field, 4, arr([h(c["a"]) for c in cases]))) == [h(c["mont"]) for c in cases]          # to_mont
    assert ints(zkg.field_op(field, 5, a)) == [h(c["a"]) for c in cases]                                           # from_mont
    for op, key in ((0, "mul"), (1, "add"), (2, "sub")):
        assert ints(zkg.field_op(field, op, a, b), p) == [h(c[key]) for c in cases], key
    assert ints(zkg.field_op(field, 6, a), p) == [h(c["neg"]) for c in cases]
    assert ints(zkg.field_op(field, 7, a), p) == [h(c["a"]) ** 2 % p for c in cases]
    inv = [c for c in cases if c["inv"] is not None]
    assert ints(zkg.field_op(field, 3, arr([h(c["a"]) for c in inv], p)), p) == [h(c["inv"]) for c in inv]


def test_fq2_ops_golden(zkg):
    cases = golden("field.json")["cases"]["fq2"]
    a = np.stack([arr([h(x) for x in c["a"]], Q).reshape(8) for c in cases]); b = np.stack([arr([h(x) for x in c["b"]], Q).reshape(8) for c in cases])
    flat = lambda key: [h(x) for c in cases for x in c[key]]
    assert ints(zkg.field_op(2, 0, a, b), Q) == flat("mul")
    assert ints(zkg.field_op(2, 3, a), Q) == flat("inv")
    assert ints(zkg.field_op(2, 7, a), Q) == flat("sqr")


@pytest.mark.parametrize("field,p", [(0, Q), (1, R)])
def test_field_ops_random_vs_oracle(zkg, oracle, field, p):
    """4096 random pairs plus the edge values 0, 1, p - 1, p - 2, 2^k; every result canonical and equal to the oracle's"""
    n = 4096
    a = random_fr_canonical(n, 11 + field); b = random_fr_canonical(n, 13 + field)            # < r < q: valid canonical values of both fields
    edge = arr([0, 1, p - 1, p - 2, 2, 1 << 128, (1 << 253) % p, (1 << 255) % p])
    a[: len(edge)] = edge; b[len(edge): 2 * len(edge)] = edge; b[:4] = arr([p - 1, p - 1, p - 1, 1])
    am = zkg.field_op(field, 4, a); bm = zkg.field_op(field, 4, b)
    for op in (0, 1, 2):
        got = zkg.field_op(field, op, am, bm)
        exp = np.stack([oracle.fp_op(field, op, am[i], bm[i]) for i in range(0, n, 37)])
        assert np.array_equal(got[::37], exp), op
    pa, pb = ints(a), ints(b)
    assert ints(zkg.field_op(field, 0, am, bm), p) == [x * y % p for x, y in zip(pa, pb)]
    assert ints(zkg.field_op(field, 2, am, bm), p) == [(x - y) % p for x, y in zip(pa, pb)]
    nz = [i for i, x in enumerate(pa) if x][:64]
    assert ints(zkg.field_op(field, 3, am[nz]), p) == [pow(pa[i], -1, p) for i in nz]


def test_fq_29bit_representation_vs_integers(zkg):
    """the 9 x 29-bit representation of the bucket-accumulation kernel (csrc/fq29.hip.hpp: Montgomery product without carry folds, limb-wise
    add / sub with spread multiples of q, conversions to and from libff's 4 x 64-bit Montgomery form) against plain Python integers:
    8192 random pairs plus the edges of the value range and pairs whose difference is 0 or a multiple of q away from it"""
    p = Q
    n = 8192
    a = random_fr_canonical(n, 31); b = random_fr_canonical(n, 33)
    edge = [0, 1, 2, p - 1, p - 2, (1 << 253) % p, (1 << 255) % p, (1 << 232) - 1, 1 << 232, (1 << 29) - 1, 1 << 29, (p - 1) // 2]
    ea = arr(edge); a[: len(edge)] = ea; b[len(edge): 2 * len(edge)] = ea; b[: len(edge)] = arr(list(reversed(edge)))
    b[100:140] = a[100:140]                                                  # equal operands: differences that are zero
    am = zkg.field_op(0, 4, a); bm = zkg.field_op(0, 4, b)
    pa, pb = ints(a), ints(b)
    assert ints(zkg.field_op(0, 10, am, bm), p) == [x * y % p for x, y in zip(pa, pb)]
    assert ints(zkg.field_op(0, 11, am, bm), p) == [(x + y) % p for x, y in zip(pa, pb)]
    assert ints(zkg.field_op(0, 12, am, bm), p) == [(x - y) % p for x, y in zip(pa, pb)]
    assert ints(zkg.field_op(0, 13, am, bm), p) == [x if x != y else 0 for x, y in zip(pa, pb)]
    assert ints(zkg.field_op(0, 14, am, bm), p) == [((y - x) * (x - y) - (y - x) ** 2 - 2 * x * y) % p for x, y in zip(pa, pb)]
    assert ints(zkg.field_op(0, 15, am, bm), p) == [pow(3 * x, -1, p) if x else 0 for x in pa]        # f29::inverse (safegcd divsteps), 0 -> 0
    # outputs are canonical limbs (below q), not merely congruent
    out = zkg.field_op(0, 10, am, bm)
    assert all(v < p for v in ints(out))


def test_g1_quad_addition_29bit_vs_oracle(zkg, oracle):
    """the 29-bit group law of the bucket-reduction kernels (csrc/fq29.hip.hpp xyzz29_add_quad, one DPP quad per addition) against the
    oracle's Jacobian arithmetic: random pairs, P + P (the doubling branch), P + (-P), infinity on either side, and chains of sums of
    sums (x <- 2x + b), which walk the value bounds the representation keeps between additions"""
    n = 256
    ks = random_fr_canonical(2 * n, 71)
    pts = oracle.g1_fixed_base(oracle.g1_generator(), ks)                    # affine (x, y), 8 limbs each
    one = arr([1], Q).reshape(4)
    jac = np.concatenate([pts, np.tile(one, (2 * n, 1))], axis=1)            # normalised Jacobian: Z = 1
    a, b = jac[:n].copy(), jac[n:].copy()
    inf = np.concatenate([arr([0, 1, 0], Q).reshape(12)])
    b[0] = a[0]                                                              # doubling
    b[1] = a[1]; b[1, 4:8] = arr([Q - int(ints(a[1, 4:8], Q)[0])], Q).reshape(4)   # P + (-P)
    a[2] = inf; b[3] = inf; a[4] = inf; b[4] = inf
    for chain in (0, 1, 5):
        got = zkg.g1_add_quad29(a, b, chain)
        for i in range(n):
            exp = oracle.g1_sum(np.stack([a[i], b[i]]))
            for _ in range(chain):
                exp = oracle.g1_sum(np.stack([exp, exp, b[i]]))
            assert np.array_equal(got[i], exp), (chain, i)
    lane = zkg.g1_add_quad29(a, b, -1)                                       # the one-lane form of the same addition
    assert all(np.array_equal(lane[i], oracle.g1_sum(np.stack([a[i], b[i]]))) for i in range(n))
    # the pair form (xyzz29_add_pair: two lanes per addition, the bucket reduction's since round 4): the same cases and chains, and long
    # chains (33 and 129 dependent additions per pair) on the first points
    for chain in (0, 1, 5):
        got = zkg.g1_add_pair29(a, b, chain)
        for i in range(n):
            exp = oracle.g1_sum(np.stack([a[i], b[i]]))
            for _ in range(chain):
                exp = oracle.g1_sum(np.stack([exp, exp, b[i]]))
            assert np.array_equal(got[i], exp), ("pair", chain, i)
    for chain in (16, 64):
        got = zkg.g1_add_pair29(a[:12], b[:12], chain)
        for i in range(12):
            exp = oracle.g1_sum(np.stack([a[i], b[i]]))
            for _ in range(chain):
                exp = oracle.g1_sum(np.stack([exp, exp, b[i]]))
            assert np.array_equal(got[i], exp), ("pair", chain, i)
