"""CPU suite: oracle prover / generator against the known-trapdoor golden Groth16 cases."""
import numpy as np
import pytest

from r1cs_util import golden_case_arrays
from util import R, arr, golden, h, ints

CASES = golden("groth16.json") + golden("groth16_step.json")      # basic_radix2 and step_radix2 domains
SETUP_CASES = CASES[:3] + CASES[4:7]


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_prove_matches_definition(oracle, case):
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    keep = []
    cs = oracle.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    assert oracle.r1cs_is_satisfied(cs, w)
    assert oracle.evaluation_domain_size(cs.num_constraints + cs.num_inputs + 1) == case["m"]
    assert oracle.evaluation_domain_is_step(cs.num_constraints + cs.num_inputs + 1) == (case["domain"] == "step")
    # coefficients_for_H == polynomial long division in pyref
    hh = oracle.qap_witness_h(cs, w, case["m"])
    assert ints(hh, R) == [h(x) for x in case["h"]]
    pk = oracle.make_pk(cs, pts)
    rc, proof = oracle.groth16_prove(pk, w, r, s)
    assert rc == 0 and len(proof) == 134
    assert proof.hex() == case["proof_hex"]
    rc2, proof2 = oracle.groth16_prove(pk, w, r, s, chunks=2)
    assert rc2 == 0 and proof2 == proof
    # unsatisfied witness -> rc 1 and no proof (snark.cpp:121-124)
    bad = w.copy(); bad[-1, 0] ^= np.uint64(1)
    rc3, _ = oracle.groth16_prove(pk, bad, r, s)
    assert rc3 == 1


@pytest.mark.parametrize("case", SETUP_CASES, ids=[c["tag"] for c in SETUP_CASES])
def test_setup_matches_definition(oracle, case):
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    keep = []
    cs = oracle.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    td = arr([h(case["trapdoor"][k]) for k in ("t", "alpha", "beta", "gamma", "delta")])
    crs = oracle.groth16_setup(cs, td)
    for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "A_query", "B_g1", "B_g2", "H_query", "L_query"):
        assert np.array_equal(crs[k].reshape(-1), pts[k].reshape(-1)), k
