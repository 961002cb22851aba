"""CPU suite: the C++ oracle (oracle/zkoracle.cpp) against the big-integer golden vectors
(tests/golden/*.json, authored by tests/golden/gen_golden.py from oracle/pyref.py)."""
import numpy as np
import pytest

from util import (Q, R, arr, g1_aff, g1_jac_expected, g2_aff, g2_jac_expected, golden, h, ints, limbs)


def test_constants(oracle):
    import ctypes as C
    q = np.zeros(4, np.uint64); r = np.zeros(4, np.uint64); qi = C.c_uint64(); ri = C.c_uint64()
    q1 = np.zeros(4, np.uint64); r1 = np.zeros(4, np.uint64); q2 = np.zeros(4, np.uint64); r2 = np.zeros(4, np.uint64); root = np.zeros(4, np.uint64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.lib().zko_constants(P(q), P(r), C.byref(qi), C.byref(ri), P(q1), P(r1), P(q2), P(r2), P(root))
    g = golden("field.json")
    assert ints(q)[0] == Q == h(g["q"]) and ints(r)[0] == R == h(g["r"])
    # SURVEY.md §7 step 1 constants (verified there with independent big-int arithmetic)
    assert qi.value == 0x87d20782e4866389 and ri.value == 0xc2e1f593efffffff
    assert ints(r2)[0] == 0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7
    assert ints(q2)[0] == 0x06d89f71cab8351f47ab1eff0a417ff6b5e71911d44501fbf32cfc5b538afa89
    assert ints(root)[0] == h(g["fr_root_of_unity"]) == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert ints(q1)[0] == (1 << 256) % Q and ints(r1)[0] == (1 << 256) % R


@pytest.mark.parametrize("name,field,p", [("fq", 0, Q), ("fr", 1, R)])
def test_field_ops(oracle, name, field, p):
    for c in golden("field.json")["cases"][name]:
        a, b = arr([h(c["a"])], p)[0], arr([h(c["b"])], p)[0]
        assert ints(a)[0] == h(c["mont"])
        assert ints(oracle.fp_op(field, 4, arr([h(c["a"])])[0]))[0] == h(c["mont"])          # to_mont
        assert ints(oracle.fp_op(field, 5, a))[0] == h(c["a"])                               # from_mont
        for op, key in ((0, "mul"), (1, "add"), (2, "sub")):
            assert ints(oracle.fp_op(field, op, a, b), p)[0] == h(c[key]), key
        assert ints(oracle.fp_op(field, 6, a), p)[0] == h(c["neg"])
        assert ints(oracle.fp_op(field, 7, a), p)[0] == h(c["a"]) ** 2 % p
        if c["inv"] is not None:
            assert ints(oracle.fp_op(field, 3, a), p)[0] == h(c["inv"])


def test_fq2_ops(oracle):
    for c in golden("field.json")["cases"]["fq2"]:
        a = arr([h(x) for x in c["a"]], Q).reshape(8); b = arr([h(x) for x in c["b"]], Q).reshape(8)
        assert ints(oracle.fq2_op(0, a, b), Q) == [h(x) for x in c["mul"]]
        assert ints(oracle.fq2_op(3, a), Q) == [h(x) for x in c["inv"]]
        assert ints(oracle.fq2_op(7, a), Q) == [h(x) for x in c["sqr"]]


def test_curve(oracle):
    g = golden("curve.json")
    G1, G2 = oracle.g1_generator(), oracle.g2_generator()
    assert ints(G1, Q) == [1, 2] and oracle.g1_on_curve(G1) and oracle.g2_on_curve(G2)
    # SURVEY.md §8c KAT (3): 2*G1
    two = oracle.g1_scalar_mul(G1, limbs(2))
    assert ints(two, Q)[:2] == [1368015179489954701390400359078579693043519447331113978918064868415326638035,
                                9918110051302171585080402603319702774565515993150576347155970296011118125764]
    for c in g["g1_mul"]:
        assert np.array_equal(oracle.g1_scalar_mul(G1, limbs(h(c["k"]))), g1_jac_expected(c["p"]))
    for c in g["g2_mul"]:
        assert np.array_equal(oracle.g2_scalar_mul(G2, limbs(h(c["k"]))), g2_jac_expected(c["p"]))
    for c in g["g1_add"]:
        assert np.array_equal(oracle.g1_add(g1_aff(c["a"]), g1_aff(c["b"])), g1_jac_expected(c["sum"]))
    for c in g["g2_add"]:
        assert np.array_equal(oracle.g2_add(g2_aff(c["a"]), g2_aff(c["b"])), g2_jac_expected(c["sum"]))
    # r*G = O
    assert np.array_equal(oracle.g1_scalar_mul(G1, limbs(R)), g1_jac_expected(None))
    assert np.array_equal(oracle.g2_scalar_mul(G2, limbs(R)), g2_jac_expected(None))
    # fixed-base batch == scalar_mul
    ks = [0, 1, 2, R - 1, 0xdeadbeef << 200]
    fb = oracle.g1_fixed_base(G1, arr(ks))
    for k, row in zip(ks, fb):
        exp = oracle.g1_scalar_mul(G1, limbs(k))
        assert np.array_equal(row, exp[:8] if k else np.zeros(8, np.uint64))
    fb2 = oracle.g2_fixed_base(G2, arr(ks))
    for k, row in zip(ks, fb2):
        exp = oracle.g2_scalar_mul(G2, limbs(k))
        assert np.array_equal(row, exp[:16] if k else np.zeros(16, np.uint64))


def test_ntt(oracle):
    for c in golden("ntt.json"):
        a = arr([h(x) for x in c["a"]], R)
        for inv in (0, 1):
            for coset in (0, 1):
                out = oracle.fft(a, inverse=inv, coset=coset)
                assert ints(out, R) == [h(x) for x in c[f"out_inv{inv}_coset{coset}"]], (c["logn"], inv, coset)


def test_msm(oracle):
    g = golden("msm.json")
    for c in g["g1"]:
        n = len(c["bases"])
        bases = np.array([g1_aff(b) for b in c["bases"]], np.uint64).reshape(n, 8)
        sc = arr([h(s) for s in c["scalars"]]) if n else np.zeros((0, 4), np.uint64)
        for method in (oracle.NAIVE, oracle.BDLO12, oracle.MIXED):
            assert np.array_equal(oracle.msm_g1(bases, sc, method), g1_jac_expected(c["result"])), (c["tag"], method)
        if n >= 7:
            assert np.array_equal(oracle.msm_g1(bases, sc, oracle.BDLO12, chunks=3), g1_jac_expected(c["result"]))
    for c in g["g2"]:
        n = len(c["bases"])
        bases = np.array([g2_aff(b) for b in c["bases"]], np.uint64).reshape(n, 16)
        sc = arr([h(s) for s in c["scalars"]])
        for method in (oracle.NAIVE, oracle.BDLO12, oracle.MIXED):
            assert np.array_equal(oracle.msm_g2(bases, sc, method), g2_jac_expected(c["result"])), (c["tag"], method)


def test_domain_rule(oracle):
    # libfqfft get_evaluation_domain: basic_radix2 for powers of two and for big + rounded_small == 2 big, step_radix2 otherwise
    assert oracle.evaluation_domain_size(32) == 32
    assert oracle.evaluation_domain_size(31) == 32      # big=16, small=15 -> rounded 16
    assert oracle.evaluation_domain_size(24) == 24 and oracle.evaluation_domain_is_step(24)      # 16 + 8
    assert oracle.evaluation_domain_size(21) == 24 and oracle.evaluation_domain_is_step(21)      # 16 + 5 -> 16 + 8
    assert oracle.evaluation_domain_size((1 << 18) - 3) == 1 << 18
    for k, kind, m in golden("step_domain.json")["rule"]:
        assert oracle.evaluation_domain_size(k) == m and oracle.evaluation_domain_is_step(k) == (kind == "step"), k


def test_step_domain_transforms(oracle):
    """libfqfft's step_radix2_domain algorithms (restated in zkoracle.cpp) against the textbook definitions
    (evaluation at the domain points / Lagrange interpolation) in tests/golden/step_domain.json"""
    g = golden("step_domain.json")
    for c in g["fft"]:
        a = arr([h(x) for x in c["a"]], R)
        for inv in (0, 1):
            for coset in (0, 1):
                out = oracle.fft(a, inverse=inv, coset=coset)
                assert ints(out, R) == [h(x) for x in c[f"out_inv{inv}_coset{coset}"]], (c["m"], inv, coset)
    for c in g["lagrange"]:
        u, z = oracle.domain_lagrange(c["m"], limbs(h(c["t"])))
        assert ints(u, R) == [h(x) for x in c["u"]], c["m"]
        assert ints(z, R)[0] == h(c["Z"]), c["m"]
