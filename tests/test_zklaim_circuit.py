"""CPU suite: zklaim's credential circuit rebuilt on the host (zklaim_amd/csrc/zklaim_circuit.hip) — the R1CS + witness
that the reference builds with zklaim_gadget (zklaim_gadget.cpp:153-784).  Checked against SHA-256 from hashlib, the oracle's
R1CS evaluator and the truth table of the comparison operators.  No GPU needed: this layer is host code."""
import hashlib
import struct

import numpy as np
import pytest

import zklaim_amd as zkg
from util import R, ints

ATTRS = [1994, 7, 42, 0, 2 ** 64 - 1]


def payload(ops, refs, attrs=ATTRS, salt=0x1122334455667788, **kw):
    return dict(attrs=attrs, refs=refs, ops=ops, salt=salt, **kw)


def test_single_payload_satisfied_and_shapes(oracle):
    keep = []
    ctx = zkg.make_ctx([payload(["less", "eq", "greater", "noop", "greater_or_eq"], [2000, 7, 41, 5, 2 ** 64 - 1])], keep)
    ck = zkg.ZklaimCircuit(ctx)
    assert ck.is_satisfied(), ck.first_unsatisfied()
    n, l, C = ck.r1cs.num_variables, ck.r1cs.num_inputs, ck.r1cs.num_constraints
    assert l == 6                                           # ceil(1280 / 253), zklaim_gadget.cpp:357-362
    assert 24576 < C + l + 1 <= 32768, C                     # k = 1 lands on the 2^15 radix-2 domain like the reference's circuit
    # the oracle's evaluator agrees that the exported CSR + witness is a satisfied system
    A, B, Cm = ck.csr(); w = ck.witness()
    ocs = oracle.make_r1cs(n, l, A, B, Cm, keep)
    assert oracle.r1cs_is_satisfied(ocs, w)
    # public input == zklaim_input_map(ctx) (what the verifier recomputes, zklaim_gadget.cpp:115-150)
    assert np.array_equal(zkg.zklaim_input_map(ctx), w[:l])
    # structure does not depend on the witness: the setup-time circuit has the same shape
    ck0 = zkg.ZklaimCircuit(ctx, with_witness=False)
    assert (ck0.r1cs.num_variables, ck0.r1cs.num_inputs, ck0.r1cs.num_constraints) == (n, l, C)
    for (r0, c0, v0), (r1, c1, v1) in zip(ck0.csr(), ck.csr()):
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)
    # the witness-only pass (what libsnark_prove uses with a resident key) numbers and assigns the variables identically
    wo = zkg.ZklaimCircuit(ctx, witness_only=True)
    assert wo.r1cs.num_variables == n and np.array_equal(wo.witness(), w)
    print("zklaim k=1: variables", n, "constraints", C)


def test_sha256_gadget_matches_hashlib():
    """the hash bits are public inputs; a wrong digest (any single bit) must violate the system"""
    keep = []
    rng = np.random.default_rng(7)
    for _ in range(3):
        attrs = [int(x) for x in rng.integers(0, 2 ** 63, 5)]
        salt = int(rng.integers(0, 2 ** 63))
        good = zkg.make_ctx([payload(["noop"] * 5, [0] * 5, attrs=attrs, salt=salt)], keep)
        assert zkg.ZklaimCircuit(good).is_satisfied()
        pre = struct.pack("<5QQ", *attrs, salt)
        h = bytearray(hashlib.sha256(pre).digest())
        h[int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
        bad = zkg.make_ctx([payload(["noop"] * 5, [0] * 5, attrs=attrs, salt=salt, hash=bytes(h))], keep)
        assert not zkg.ZklaimCircuit(bad).is_satisfied()


@pytest.mark.parametrize("op,fn", [("less", lambda a, b: a < b), ("less_or_eq", lambda a, b: a <= b), ("eq", lambda a, b: a == b),
                                   ("greater_or_eq", lambda a, b: a >= b), ("greater", lambda a, b: a > b), ("not_eq", lambda a, b: a != b),
                                   ("noop", lambda a, b: True)])
def test_comparison_truth_table(op, fn):
    keep = []
    for a, b in [(5, 9), (9, 5), (7, 7), (0, 0), (0, 2 ** 64 - 1), (2 ** 64 - 1, 0), (2 ** 64 - 1, 2 ** 64 - 1), (2 ** 63, 2 ** 63 - 1)]:
        ctx = zkg.make_ctx([payload([op, "noop", "noop", "noop", "noop"], [b, 0, 0, 0, 0], attrs=[a, 1, 2, 3, 4])], keep)
        assert zkg.ZklaimCircuit(ctx).is_satisfied() == fn(a, b), (op, a, b)


def test_payload_counts(oracle):
    """k = 0 (tests/zklaim.cpp:341-353 can_handle_no_payload), 2 and 3 payloads (can_handle_two/three_payloads)"""
    keep = []
    ck = zkg.ZklaimCircuit(zkg.make_ctx([], keep))
    assert ck.is_satisfied() and ck.r1cs.num_inputs == 0 and ck.r1cs.num_constraints == 1
    pls = [payload(["less", "noop", "noop", "noop", "noop"], [3000, 0, 0, 0, 0], salt=s) for s in (1, 2, 3)]
    for k in (2, 3):
        ctx = zkg.make_ctx(pls[:k], keep)
        ck = zkg.ZklaimCircuit(ctx)
        assert ck.is_satisfied()
        assert ck.r1cs.num_inputs == -(-1280 * k // 253)
        assert np.array_equal(zkg.zklaim_input_map(ctx), ck.witness()[: ck.r1cs.num_inputs])


def test_parallel_sub_circuits_equal_the_serial_pass():
    """payload sub-circuits built on the host pool through views (variables, witness, constraints appended in payload order)
    give byte-identical CSR matrices and witness to the single-threaded pass"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import zklaim_amd as zkg
keep = []
pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i) for i in range(3)]
ck = zkg.ZklaimCircuit(zkg.make_ctx(pls, keep))
h = hashlib.sha256()
for m in ck.csr():
    for a in m:
        h.update(np.ascontiguousarray(a).tobytes())
h.update(np.ascontiguousarray(ck.witness()).tobytes())
wo = zkg.ZklaimCircuit(zkg.make_ctx(pls, keep), witness_only=True)
h.update(np.ascontiguousarray(wo.witness()).tobytes())
print(h.hexdigest(), ck.r1cs.num_constraints, ck.r1cs.num_variables)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for serial in (False, True):
        env = dict(os.environ)
        env.pop("ZKG_SERIAL_CIRCUIT", None)
        if serial:
            env["ZKG_SERIAL_CIRCUIT"] = "1"
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and outs[0].split()[1] == str(3 * 27651 + 2), outs


def _var_layout(k):
    """0-based witness positions of the value variables, in the allocation order of build_zklaim (zklaim_circuit.hip; the reference's
    constructor order, zklaim_gadget.cpp:348-540): inputs, zero, (data, less, less_or_eq) x 5 per payload, plvars, refvals, opsvals"""
    n_inputs = -(-1280 * k // 253)
    base = n_inputs + 1
    data = lambda i, j: base + 3 * (5 * i + j)
    pl0 = base + 15 * k
    return dict(data=data, plvars=lambda c: pl0 + c, refvals=lambda c: pl0 + 6 * k + c, opsvals=lambda c: pl0 + 14 * k + c)


def test_malicious_prover_cannot_rebind_values(oracle):
    """The reference never generates the pack_PL / pack_REF / pack_OPS constraints (zklaim_gadget.cpp:583-699), which leaves the op
    one-hot, the reference values and the compared attributes free: a prover who knows the pre-image satisfies ANY public predicate.
    The default circuit here enforces the packings; ZKG_CIRCUIT_REFERENCE_QUIRK reproduces the reference's shape.  An honest
    witness for the false claim 1994 > 2000 is patched the way a cheating prover would, and checked on both systems."""
    from util import R, arr
    keep = []
    honest = payload(["less", "noop", "noop", "noop", "noop"], [2000, 0, 0, 0, 0])
    claim = payload(["greater", "noop", "noop", "noop", "noop"], [2000, 0, 0, 0, 0])                  # 1994 > 2000: false
    lay = _var_layout(1)
    one, zero = arr([1], R)[0], arr([0], R)[0]
    systems = {}
    for quirk in (False, True):
        ck = zkg.ZklaimCircuit(zkg.make_ctx([claim], keep), reference_quirk=quirk)
        assert not ck.is_satisfied()                                                                  # the honest witness generator refuses
        A, B, Cm = ck.csr()
        systems[quirk] = (oracle.make_r1cs(ck.r1cs.num_variables, ck.r1cs.num_inputs, A, B, Cm, keep), ck.witness(), ck.r1cs.num_constraints)
    assert systems[False][2] == systems[True][2] + 78 and np.array_equal(systems[False][1], systems[True][1])     # same witness, 6 + 8 + 64 more rows
    w = systems[False][1]
    assert np.array_equal(w[lay["data"](0, 0)], arr([1994], R)[0]) and np.array_equal(w[lay["refvals"](0)], arr([2000], R)[0])
    assert np.array_equal(w[lay["opsvals"](4)], one) and np.array_equal(w[lay["plvars"](0)], arr([1994], R)[0])
    # attack 1: flip the private copy of the op one-hot to noop (public ops bits still say "greater")
    forged = w.copy(); forged[lay["opsvals"](4)] = zero; forged[lay["opsvals"](6)] = one
    assert oracle.r1cs_is_satisfied(systems[True][0], forged), "the reference's shape accepts the forged one-hot (that is the quirk)"
    assert not oracle.r1cs_is_satisfied(systems[False][0], forged)
    # attack 2: compare against a different reference value than the public one.  Rebuild the comparison's witness by asking the
    # generator for the same attributes against ref 1000 (1994 > 1000 holds) and keep the PUBLIC input of the false claim.
    other = zkg.ZklaimCircuit(zkg.make_ctx([payload(["greater", "noop", "noop", "noop", "noop"], [1000, 0, 0, 0, 0])], keep), reference_quirk=True)
    assert other.is_satisfied()
    forged2 = other.witness().copy()
    l = other.r1cs.num_inputs
    first_bit = 6 + 1 + 15 + 6 + 8 + 64                                                              # h_bits | ref_bits | ops_bits follow the value variables
    forged2[:l] = w[:l]; forged2[first_bit:first_bit + 1280] = w[first_bit:first_bit + 1280]          # public values and their bits: the false claim's
    assert oracle.r1cs_is_satisfied(systems[True][0], forged2), "the reference's shape lets refvals differ from the public reference bits"
    assert not oracle.r1cs_is_satisfied(systems[False][0], forged2)
    # the honest statement is accepted by both
    for quirk in (False, True):
        ck = zkg.ZklaimCircuit(zkg.make_ctx([honest], keep), reference_quirk=quirk)
        assert ck.is_satisfied()
