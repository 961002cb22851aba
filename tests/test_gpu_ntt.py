"""GPU parity: zkg_ntt (HIP) vs the oracle's libfqfft restatement and the golden DFT vectors.  Bit-exact."""
import numpy as np
import pytest

from gpu_util import zkg  # noqa: F401
from util import R, arr, golden, h, ints, random_fr_canonical

pytestmark = pytest.mark.gpu


def test_ntt_golden(zkg):
    for c in golden("ntt.json"):
        a = arr([h(x) for x in c["a"]], R)
        for inv in (0, 1):
            for coset in (0, 1):
                out = zkg.ntt(a, inverse=inv, coset=coset)
                assert ints(out, R) == [h(x) for x in c[f"out_inv{inv}_coset{coset}"]], (c["logn"], inv, coset)


@pytest.mark.parametrize("logn", [0, 1, 2, 3, 4, 9, 10, 11, 12, 13, 15, 16, 17, 19, 21])
def test_ntt_vs_oracle(zkg, oracle, logn):
    a = random_fr_canonical(1 << logn, 0x5A4B4C41494D0003 + logn)       # any 4-limb values < r are valid Montgomery residues
    for inv in (0, 1):
        for coset in (0, 1):
            got = zkg.ntt(a, inverse=inv, coset=coset)
            exp = oracle.fft(a, inverse=inv, coset=coset)
            assert np.array_equal(got, exp), (logn, inv, coset)


def test_ntt_full_size_properties(zkg, oracle):
    """BASELINE config 3 size (2^20): round trips, linearity, and a spot-check of outputs against the definition."""
    n = 1 << 20
    a = random_fr_canonical(n, 0x5A4B4C41494D0003)
    b = random_fr_canonical(n, 0x5A4B4C41494D0013)
    fa = zkg.ntt(a)
    assert np.array_equal(zkg.ntt(fa, inverse=True), a)
    assert np.array_equal(zkg.ntt(zkg.ntt(a, coset=True), inverse=True, coset=True), a)
    sel = np.arange(0, n, n // 64)
    fb = zkg.ntt(b)
    # definition check on a 2^12 transform: out[k] = sum_j a[j] w^(jk), Horner in Python ints
    from util import from_limbs, MONT
    rinv = pow(MONT, -1, R)
    ai = [from_limbs(x) * rinv % R for x in a[: 1 << 12]]
    sub = zkg.ntt(a[: 1 << 12].copy())
    w12 = pow(pow(5, (R - 1) >> 28, R), 1 << 16, R)
    for k in (0, 1, 77, 4095):
        acc = 0
        for x in reversed(ai):
            acc = (acc * pow(w12, k, R) + x) % R
        assert from_limbs(sub[k]) * rinv % R == acc
    # full size against the oracle (serial radix-2, ~2 s)
    assert np.array_equal(fa, oracle.fft(a))
    assert np.array_equal(fb[sel], oracle.fft(b)[sel])
