"""CPU suite: the host pairing and Groth16 verifier of libzkg.so (zkg_groth16_verify, replaces
r1cs_gg_ppzksnark_verifier_strong_IC, snark.cpp:62).  The verification keys are assembled HERE from the golden known-trapdoor
cases (oracle arithmetic + Python integers), so the vk layout parser, the pairing and the verification equation are checked
against proofs whose validity pyref established independently (discrete-log identity, tests/golden/gen_golden.py)."""
import numpy as np
import pytest

import zklaim_amd as zkg
from r1cs_util import golden_case_arrays
from util import MONT, Q, R, arr, from_limbs, golden, h, ints, limbs

CASES = golden("groth16.json")


def test_pairing_bilinear_and_nondegenerate():
    e = lambda a, b: zkg.pairing_probe(limbs(a), limbs(b))
    one = e(0, 5)                                            # a point at infinity pairs to 1
    assert e(1, 1) != one and e(1, 1) != e(2, 1)
    assert e(2, 3) == e(6, 1) == e(1, 6) == e(3, 2)
    a, b = 0x1234567890abcdef1234567890abcdef, 0xfedcba0987654321fedcba0987654321
    assert e(a, b) == e(a * b % R, 1) == e(1, a * b % R)
    assert e(R - 1, 1) == e(1, R - 1) != e(1, 1)             # e(-P, Q) = e(P, -Q) = e(P, Q)^-1
    assert e(R - 1, R - 1) == e(1, 1)


def test_final_exponentiation_chain_and_frobenius():
    """libff's last chunk (Fuentes-Castaneda chain) raises to lambda_0 + lambda_1 q + lambda_2 q^2 + lambda_3 q^3 =
    2z(6z^2 + 3z + 1) (q^4 - q^2 + 1)/r; the coefficient-wise Frobenius maps equal powering by q, q^2, q^3"""
    z = 4965661367192848881
    assert 36 * z**4 + 36 * z**3 + 24 * z**2 + 6 * z + 1 == Q and 36 * z**4 + 36 * z**3 + 18 * z**2 + 6 * z + 1 == R
    lam = [12 * z**3 + 12 * z**2 + 6 * z + 1, 12 * z**3 + 6 * z**2 + 4 * z, 12 * z**3 + 6 * z**2 + 6 * z, 12 * z**3 + 6 * z**2 + 4 * z - 1]
    e = sum(l * Q**i for i, l in enumerate(lam))
    hard = (Q**4 - Q**2 + 1) // R
    assert (Q**4 - Q**2 + 1) % R == 0 and e == 2 * z * (6 * z * z + 3 * z + 1) * hard
    assert zkg.pairing_selfcheck(e) == 0
    assert zkg.pairing_selfcheck(hard) == 16                 # the exact hard exponent is a different (equally valid) pairing value


def ser_fq(x):
    return (x * MONT % Q).to_bytes(32, "little")


def ser_g1_aff(p8):
    """8 Montgomery limbs (affine, all-zero = infinity) -> libsnark compressed G1"""
    if not np.any(p8):
        return b"1" + bytes(32) + b"1"
    x, y = ints(p8, Q)
    return b"0" + ser_fq(x) + (b"1" if y & 1 else b"0")


def ser_g2_aff(p16):
    if not np.any(p16):
        return b"1" + bytes(64) + b"1"
    x0, x1, y0, y1 = ints(p16, Q)
    return b"0" + ser_fq(x0) + ser_fq(x1) + (b"1" if y0 & 1 else b"0")


def build_vk(oracle, case, keep):
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    ocs = oracle.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    td = {k: h(v) for k, v in case["trapdoor"].items()}
    setup = oracle.groth16_setup(ocs, arr([td[k] for k in ("t", "alpha", "beta", "gamma", "delta")]))
    At, Bt, Ct = (ints(setup[k], R) for k in ("At", "Bt", "Ct"))
    l = case["num_inputs"]
    ginv = pow(td["gamma"], -1, R)
    ic_scalars = [(td["beta"] * At[i] + td["alpha"] * Bt[i] + Ct[i]) * ginv % R for i in range(l + 1)]
    IC = oracle.g1_fixed_base(oracle.g1_generator(), arr(ic_scalars))
    gamma_g2 = oracle.g2_scalar_mul(oracle.g2_generator(), limbs(td["gamma"]))[:16]
    delta_g2 = oracle.g2_scalar_mul(oracle.g2_generator(), limbs(td["delta"]))[:16]
    blob = zkg.pairing_probe(limbs(td["alpha"]), limbs(td["beta"]))                 # alpha_g1_beta_g2 = e(alpha G1, beta G2)
    blob += ser_g2_aff(gamma_g2) + ser_g2_aff(delta_g2) + ser_g1_aff(IC[0])
    blob += b"%d\n%d\n" % (l, l) + b"".join(b"%d\n" % i for i in range(l)) + b"%d\n" % l + b"".join(ser_g1_aff(IC[i + 1]) for i in range(l))
    return blob, w[:l]


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_verify_golden_proofs(oracle, case):
    keep = []
    vk, x = build_vk(oracle, case, keep)
    proof = bytes.fromhex(case["proof_hex"])
    assert zkg.groth16_verify(vk, x, proof) == 0
    # any change to the proof, the public input or the key must be rejected
    for pos in (5, 40, 110):
        bad = bytearray(proof); bad[pos] ^= 1
        assert zkg.groth16_verify(vk, x, bytes(bad)) != 0
    xb = x.copy(); xb[0, 0] ^= np.uint64(1)
    assert zkg.groth16_verify(vk, xb, proof) == 1
    assert zkg.groth16_verify(vk, x[:-1], proof) == 1                            # strong input consistency: wrong input length
    vkb = bytearray(vk); vkb[10] ^= 1
    assert zkg.groth16_verify(bytes(vkb), x, proof) != 0
    assert zkg.groth16_verify(vk[:200], x, proof) == 2
    # swapping A and C (both G1) is a well-formed but invalid proof
    swapped = proof[100:134] + proof[34:100] + proof[0:34]
    assert zkg.groth16_verify(vk, x, swapped) == 1


def test_hostile_counts_in_a_vk_blob_are_refused(oracle):
    """counts inside a key blob are bounded by the bytes that follow them before anything is sized by them: a vk whose gamma_ABC
    header claims 2^61 indices / values, or more values than the blob holds, is 'malformed' (2), never an allocation or a crash"""
    keep = []
    vk, x = build_vk(oracle, CASES[1], keep)
    proof = bytes.fromhex(CASES[1]["proof_hex"])
    head = 384 + 66 + 66 + 34
    assert zkg.groth16_verify(vk, x, proof) == 0
    l = CASES[1]["num_inputs"]
    tail_idx = b"".join(b"%d\n" % i for i in range(l))
    pts = vk[len(vk) - 34 * l:]
    for dom, nidx, nval in ((1 << 61, 1 << 61, l), (l, l, 1 << 61), (l, l, l + 5), ((1 << 64) - 1, l, l), (10 ** 19, l, l)):
        forged = vk[:head] + b"%d\n%d\n" % (dom, nidx) + tail_idx + b"%d\n" % nval + pts
        assert zkg.groth16_verify(forged, x, proof) in (1, 2), (dom, nidx, nval)
    forged = vk[:head] + b"%d\n%d\n" % (l, l) + tail_idx + b"%d\n" % l + pts[:-10]                 # truncated values
    assert zkg.groth16_verify(forged, x, proof) == 2
