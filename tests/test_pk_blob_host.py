"""CPU suite: the host half of the pk blob loader (zkg_pk_blob_inspect — the section walk, the sparse B index list and the constraint
system parser that zkg_crs_upload_blob runs beside the GPU decompression), on blobs written by the oracle's restatement of
operator<<(r1cs_gg_ppzksnark_proving_key) and on damaged copies of them.  No GPU: this layer is host code
(tools/asan_host_tests.sh runs it under AddressSanitizer as well)."""
import numpy as np
import pytest

import zklaim_amd as zkg
from r1cs_util import golden_case_arrays
from util import golden

CASES = golden("groth16.json") + golden("groth16_step.json")


def _blob(oracle, case, keep):
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    ocs = oracle.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    if "domain" in case and case["domain"] == "step":
        pts = dict(pts); pts["m"] = case["m"]
    return oracle.pk_write_blob(oracle.make_pk(ocs, pts)), (A, B, C)


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_inspect_reports_the_systems_sizes(oracle, case):
    keep = []
    blob, (A, B, C) = _blob(oracle, case, keep)
    info = zkg.pk_blob_inspect(blob)
    n, l = case["num_variables"], case["num_inputs"]
    assert info["A_query"] == n + 1 and info["L_query"] == n - l and info["H_query"] == case["m"] - 1
    assert info["num_inputs"] == l and info["num_constraints"] == len(A[0]) - 1 and info["domain_size"] == case["m"]
    assert info["terms"] == len(A[1]) + len(B[1]) + len(C[1])
    assert info["B_values"] <= n + 1


def test_damaged_blobs_are_errors_not_faults(oracle):
    """every prefix of a small blob, and 400 random single-byte edits / digit insertions of a larger one: an error or (when the edit
    stays well-formed, e.g. inside a coefficient) a consistent report — never a crash or a read outside the buffer"""
    keep = []
    small, _ = _blob(oracle, CASES[0], keep)
    ok = zkg.pk_blob_inspect(small)
    for cut in range(len(small)):
        with pytest.raises(zkg.ZkgError):
            zkg.pk_blob_inspect(small[:cut])
    big, _ = _blob(oracle, CASES[2], keep)
    ref = zkg.pk_blob_inspect(big)
    rng = np.random.default_rng(41)
    refused = 0
    for _ in range(400):
        b = bytearray(big)
        kind = int(rng.integers(0, 3))
        pos = int(rng.integers(0, len(b)))
        if kind == 0:
            b[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b[pos:pos] = b"%d" % int(rng.integers(0, 10 ** 12))          # digits where none belong / a count made huge
        else:
            nl = bytes(b).find(b"\n", pos)                                 # corrupt a decimal field just before a newline
            if nl > 0:
                b[nl - 1:nl] = b"99999999999"
        try:
            info = zkg.pk_blob_inspect(bytes(b))
            assert info["A_query"] == ref["A_query"] or info["num_constraints"] != 0
        except zkg.ZkgError:
            refused += 1
    assert refused > 50
    assert zkg.pk_blob_inspect(small) == ok
    with pytest.raises(zkg.ZkgError):
        zkg.pk_blob_inspect(b"")
