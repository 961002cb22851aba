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
