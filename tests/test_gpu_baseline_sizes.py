"""GPU parity at the sizes BASELINE.json quotes (the -m gpu suite elsewhere stays small so that it runs in minutes):
  configs[1]  the full 2^20-point G1 MSM, compared with the oracle's BDLO12 restatement on ALL points;
  configs[3]  the 8-payload zklaim credential, evaluation domain m = 2^18: coefficients_for_H and the 134 proof bytes vs the oracle;
  north star  the 2^20-constraint case (37 payloads, C + l + 1 = 1 023 318 -> m = 2^20): the same two comparisons.
Reference call sites: r1cs_gg_ppzksnark_prover at zklaim/snark.cpp:126; the payload sweep of src/main_benchmark.c:175-182.
The oracle runs with `chunks = num_threads()` (libff's MULTICORE chunking: same result as one chunk, asserted in tests/test_oracle_*)."""
import time

import numpy as np
import pytest

from gpu_util import credential_payloads, dev_bases_g1, oracle_pk_from_keypair, zkg  # noqa: F401
from util import random_fr_canonical

pytestmark = pytest.mark.gpu
SEED = 0x5A4B4C41494D0000


def test_msm_2p20_all_points_vs_oracle(zkg, oracle):
    import torch
    n = 1 << 20
    d_bases, bases, _ = dev_bases_g1(zkg, n, SEED + 1)
    sc = random_fr_canonical(n, SEED + 2)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    got = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n)
    t0 = time.perf_counter()
    exp = oracle.msm_g1(bases, sc, oracle.BDLO12, oracle.num_threads())
    print(f"oracle 2^20 MSM on {oracle.num_threads()} threads: {time.perf_counter() - t0:.1f} s")
    assert np.array_equal(got, exp)
    assert np.array_equal(zkg.msm_g1(bases, sc), exp)                      # host-pointer entry point (stages the 96 MiB itself)


@pytest.mark.parametrize("k,log_m", [(8, 18), (37, 20)], ids=["cfg4_8payloads_m2p18", "northstar_37payloads_m2p20"])
def test_credential_prove_vs_oracle(zkg, oracle, k, log_m):
    keep = []
    ck = zkg.ZklaimCircuit(zkg.make_ctx(credential_payloads(k), keep))
    assert ck.is_satisfied()
    nv, l, ncons = ck.r1cs.num_variables, ck.r1cs.num_inputs, ck.r1cs.num_constraints
    w = ck.witness()
    kp = zkg.Keypair(ck.r1cs, random_fr_canonical(5, SEED + 4))
    ocs, opk, m = oracle_pk_from_keypair(oracle, kp, ck.csr(), nv, l, keep)
    assert m == 1 << log_m == zkg.evaluation_domain_size(ncons + l + 1)[0]
    crs = zkg.Crs(kp.pk)
    rs = random_fr_canonical(2, SEED + 5)
    threads = oracle.num_threads()
    t0 = time.perf_counter()
    rc_o, proof_o = oracle.groth16_prove(opk, w, rs[0], rs[1], True, threads)
    print(f"oracle prove, {k} payloads, m = 2^{log_m}, {threads} threads: {time.perf_counter() - t0:.1f} s")
    rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc_o == 0 and rc == 0 and len(proof) == 134 and proof == proof_o
    # r1cs_to_qap_witness_map: all m + 1 coefficients.  The oracle evaluates the system as the key stores it (A and B possibly
    # exchanged by the generator's swap_AB_if_beneficial; H is symmetric in them).
    h_gpu = crs.qap_witness_h(w)
    assert np.array_equal(h_gpu, oracle.qap_witness_h(ocs, w, m))
    # the seam's sparse witness form: same bytes; and the proof verifies against the vk of the same key
    tags, idx, vals = ck.sparse_witness()
    rc_s, proof_s = crs.prove_sparse(tags, idx, vals, rs[0], rs[1])
    assert rc_s == 0 and proof_s == proof
    assert zkg.groth16_verify(kp.vk_blob(), w[:l], proof) == 0
    bad = w.copy(); bad[nv - 1, 0] ^= np.uint64(1)
    assert crs.prove(bad, rs[0], rs[1])[0] == zkg.UNSATISFIED
    crs.free(); kp.free(); ck.free()
