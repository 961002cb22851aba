"""GPU parity: zkg_groth16_prove (HIP pipeline on a resident CRS) vs the definition-level golden proofs and the oracle's
restatement of r1cs_gg_ppzksnark_prover.  Proof bytes must be identical."""
import numpy as np
import pytest

from gpu_util import zkg  # noqa: F401
from r1cs_util import golden_case_arrays
from util import R, arr, golden, h, ints, random_fr_canonical

pytestmark = pytest.mark.gpu
CASES = golden("groth16.json")


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_prove_golden(zkg, case):
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    keep = []
    cs = zkg.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    pk = zkg.make_pk(cs, pts, case["m"].bit_length() - 1, keep)
    crs = zkg.Crs(pk)
    assert ints(crs.qap_witness_h(w), R) == [h(x) for x in case["h"]]
    rc, proof = crs.prove(w, r, s)
    assert rc == 0 and proof.hex() == case["proof_hex"]
    rc, proof2 = crs.prove(w, r, s, check_satisfied=False)
    assert rc == 0 and proof2 == proof
    bad = w.copy(); bad[-1, 0] ^= np.uint64(1)
    rc, _ = crs.prove(bad, r, s)
    assert rc == zkg.UNSATISFIED                     # libsnark_prove maps this to its return value 1
    crs.free()


@pytest.mark.parametrize("log_m", [10, 13, 16])
def test_prove_zklaim_shaped_vs_oracle(zkg, oracle, log_m):
    """Synthetic zklaim-shaped system (bits, AND/XOR, packing rows), CRS from the oracle's known-trapdoor generator.  2^16: the H query's
    65535 points take the two-pass digit sort and windows that share rows of buckets in pairs (prover.hip H_ROW_MERGE_MIN)."""
    from zklaim_amd import synth
    n, l, A, B, C, w = synth.zklaim_shaped(log_m, num_inputs=5, seed=log_m)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    assert oracle.r1cs_is_satisfied(ocs, w)
    td = random_fr_canonical(5, 0x5A4B4C41494D0004)
    crs_arrays = oracle.groth16_setup(ocs, td)
    assert crs_arrays["m"] == 1 << log_m
    opk = oracle.make_pk(ocs, crs_arrays)
    rs = random_fr_canonical(2, 0x5A4B4C41494D0005)
    rc_o, proof_o = oracle.groth16_prove(opk, w, rs[0], rs[1])
    assert rc_o == 0
    cs = zkg.make_r1cs(n, l, A, B, C, keep)
    pk = zkg.make_pk(cs, crs_arrays, log_m, keep)
    crs = zkg.Crs(pk)
    assert np.array_equal(crs.qap_witness_h(w), oracle.qap_witness_h(ocs, w, 1 << log_m))
    rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc == 0 and proof == proof_o
    print("stage ms", crs.stage_ms())
    crs.free()


@pytest.mark.parametrize("case", CASES[1:3], ids=[c["tag"] for c in CASES[1:3]])
def test_prove_from_pk_blob(zkg, oracle, case):
    """ctx->pk style byte blob (libsnark operator<< format, compressed points) -> GPU decompression -> same proof bytes."""
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    keep = []
    ocs = oracle.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    blob = oracle.pk_write_blob(oracle.make_pk(ocs, pts))
    crs = zkg.Crs(blob=blob, m=case["m"])
    rc, proof = crs.prove(w, r, s)
    assert rc == 0 and proof.hex() == case["proof_hex"]
    crs.free()
    # corrupt one x coordinate: the point falls off the curve (or the proof changes); truncated blobs are refused
    bad = bytearray(blob); bad[40] ^= 0x55
    try:
        crs2 = zkg.Crs(blob=bytes(bad), m=case["m"])
        rc2, proof2 = crs2.prove(w, r, s)
        assert proof2 != proof
        crs2.free()
    except zkg.ZkgError:
        pass
    with pytest.raises(zkg.ZkgError):
        zkg.Crs(blob=blob[: len(blob) // 2], m=case["m"])
    # hostile counts (the blob comes from the issuer): every count is bounded by the bytes that follow it before anything is sized by
    # it, so 2^61 A_query entries, a count just past the end, or 20 digits are refusals, not wrapped products or allocations
    head = 34 + 34 + 66 + 34 + 66
    nl = blob.index(b"\n", head)
    assert int(blob[head:nl]) == case["num_variables"] + 1
    for bogus in (b"2305843009213693952", b"%d" % (case["num_variables"] + 2), b"99999999999999999999", b"4294967296"):
        with pytest.raises(zkg.ZkgError):
            zkg.Crs(blob=blob[:head] + bogus + blob[nl:], m=case["m"])


def test_truncated_and_damaged_pk_blobs_are_refused(zkg, oracle):
    """The blob loader walks the point sections on the calling thread and the constraint-system text on a thread of its own while
    the GPU decompresses: a blob cut at any place — inside the head, a query, the sparse B index list, a term count, a coefficient —
    or with a damaged constraint system must come back as an error (never a crash, a hang or a key), and the loader must still work
    afterwards."""
    case = CASES[1]
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    keep = []
    ocs = oracle.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep)
    blob = oracle.pk_write_blob(oracle.make_pk(ocs, pts))
    n = len(blob)
    cuts = sorted(set([1, 33, 100, 269, 270, n - 1, n - 2, n - 33, n - 40] + [int(n * f / 37) for f in range(1, 37)]))
    for cut in cuts:
        with pytest.raises(zkg.ZkgError):
            zkg.Crs(blob=blob[:cut], m=case["m"])
    # the constraint-system section: a term index beyond the variable count, a term count that overruns the blob, a non-digit
    tail = blob.rindex(b"\n", 0, n - 40)                     # a newline inside the last constraints
    for bad in (blob[:tail - 1] + b"9" * 9 + blob[tail:], blob[:tail - 1] + b"x" + blob[tail:]):
        try:
            crs = zkg.Crs(blob=bad, m=case["m"])
            rc, proof = crs.prove(w, r, s)                     # (an edit that happens to stay well-formed gives another system: no golden proof)
            assert rc != 0 or proof.hex() != case["proof_hex"]
            crs.free()
        except zkg.ZkgError:
            pass
    crs = zkg.Crs(blob=blob, m=case["m"])
    rc, proof = crs.prove(w, r, s)
    assert rc == 0 and proof.hex() == case["proof_hex"]
    crs.free()


def test_pk_blob_zklaim_shaped(zkg, oracle):
    from zklaim_amd import synth
    log_m = 12
    n, l, A, B, C, w = synth.zklaim_shaped(log_m, num_inputs=5, seed=3)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x44))
    opk = oracle.make_pk(ocs, crs_arrays)
    rs = random_fr_canonical(2, 0x45)
    rc_o, proof_o = oracle.groth16_prove(opk, w, rs[0], rs[1])
    blob = oracle.pk_write_blob(opk)
    crs = zkg.Crs(blob=blob, m=1 << log_m)
    rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc_o == 0 and rc == 0 and proof == proof_o
    crs.free()


def test_prove_real_zklaim_circuit(zkg, oracle):
    """One-payload zklaim credential (the reference's main.c / can_proof shape): circuit + witness from the host layer,
    CRS from the oracle's known-trapdoor generator, proof bytes GPU == oracle; an unsatisfied credential returns 1."""
    keep = []
    pl = dict(attrs=[1994, 7, 42, 0, 5], refs=[2000, 7, 41, 0, 0], ops=["less", "eq", "greater", "noop", "noop"], salt=0xABCDEF)
    ctx = zkg.make_ctx([pl], keep)
    ck = zkg.ZklaimCircuit(ctx)
    assert ck.is_satisfied()
    n, l = ck.r1cs.num_variables, ck.r1cs.num_inputs
    A, B, C = ck.csr(); w = ck.witness()
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x77))
    log_m = crs_arrays["m"].bit_length() - 1
    assert log_m == 15
    opk = oracle.make_pk(ocs, crs_arrays)
    rs = random_fr_canonical(2, 0x78)
    rc_o, proof_o = oracle.groth16_prove(opk, w, rs[0], rs[1])
    cs = zkg.make_r1cs(n, l, A, B, C, keep)
    crs = zkg.Crs(zkg.make_pk(cs, crs_arrays, log_m, keep))
    rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc_o == 0 and rc == 0 and proof == proof_o
    print("zklaim k=1 prove stage ms", crs.stage_ms())
    bad = dict(pl); bad["ops"] = ["greater", "eq", "greater", "noop", "noop"]           # 1994 > 2000 is false
    wbad = zkg.ZklaimCircuit(zkg.make_ctx([bad], keep)).witness()
    rc, _ = crs.prove(wbad, rs[0], rs[1])
    assert rc == zkg.UNSATISFIED
    crs.free()


def test_prove_sparse_witness_matches_dense(zkg):
    """zkg_groth16_prove_sparse (tags + listed values, what the seam uploads) gives the same proof bytes as the dense witness; a
    tampered tag is an unsatisfied system (rc 1); tags agree with the dense vector"""
    keep = []
    pls = [dict(attrs=[1994 + i, 7, 42, 0, 5], refs=[2100, 7, 41, 0, 0], ops=["less", "eq", "greater", "noop", "noop"], salt=0xABC + i) for i in range(2)]
    ck = zkg.ZklaimCircuit(zkg.make_ctx(pls, keep))
    assert ck.is_satisfied()
    w = ck.witness()
    tags, idx, vals = ck.sparse_witness()
    one = arr([1], R)[0]
    assert np.array_equal(tags == 0, ~w.any(axis=1)) and np.array_equal(tags == 1, (w == one).all(axis=1))
    assert np.array_equal(w[idx], vals) and int((tags == 2).sum()) == idx.size and 0 < idx.size < 0.1 * tags.size
    kp = zkg.Keypair(ck.r1cs, random_fr_canonical(5, 0x5BA))
    crs = zkg.Crs(kp.pk)
    rs = random_fr_canonical(2, 0x5BB)
    rc, dense = crs.prove(w, rs[0], rs[1])
    rc2, sparse = crs.prove_sparse(tags, idx, vals, rs[0], rs[1])
    assert rc == 0 and rc2 == 0 and dense == sparse
    assert zkg.groth16_verify(kp.vk_blob(), w[:ck.r1cs.num_inputs], sparse) == 0
    bad = tags.copy(); k = int(np.flatnonzero(tags == 1)[-1]); bad[k] = 0
    rc3, _ = crs.prove_sparse(bad, idx, vals, rs[0], rs[1])
    assert rc3 == zkg.UNSATISFIED
    rc4, again = crs.prove(w, rs[0], rs[1])
    assert rc4 == 0 and again == dense
    crs.free(); kp.free()


def test_sparse_witness_upload_is_the_witness_split(zkg, oracle):
    """The sparse upload builds z = [1 | w] AND its multi_exp_with_mixed_addition split in one go (k_expand_tags, k_scatter_full): a listed
    value is tagged by what it IS (0 and 1 included), a tag-2 variable nobody lists is zero, a listed index that is out of range or not
    tagged 2 is refused.  Proof bytes: the dense call's and the oracle's."""
    vals_int = [0, 1, 5, R - 1, 1, 0, 7, 2, 1, 0, 123456789, 3]
    n, l, A, B, C, w = _trivial_system(vals_int)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x5CA))
    rs = random_fr_canonical(2, 0x5CB)
    rc_o, proof_o = oracle.groth16_prove(oracle.make_pk(ocs, crs_arrays), w, rs[0], rs[1])
    log_m = crs_arrays["m"].bit_length() - 1
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, log_m, keep))
    rc, dense = crs.prove(w, rs[0], rs[1])
    assert rc_o == 0 and rc == 0 and dense == proof_o
    # every variable listed (tag 2), whatever its value: the zeros and ones among them must land in the right part of the split
    tags = np.full(n, 2, np.uint8); idx = np.arange(n, dtype=np.uint32)
    assert crs.prove_sparse(tags, idx, w, rs[0], rs[1]) == (0, dense)
    # bits by tag, the rest listed in reverse order
    tags = np.array([0 if v == 0 else 1 if v == 1 else 2 for v in vals_int], np.uint8)
    idx = np.flatnonzero(tags == 2)[::-1].astype(np.uint32)
    assert crs.prove_sparse(tags, idx, w[idx], rs[0], rs[1]) == (0, dense)
    # a tag-2 variable that is not listed counts as zero: the proof of the witness with that variable zeroed
    w0 = w.copy(); w0[int(idx[0])] = 0
    rc_z, proof_z = oracle.groth16_prove(oracle.make_pk(ocs, crs_arrays), w0, rs[0], rs[1])
    assert rc_z == 0 and crs.prove_sparse(tags, idx[1:], w[idx[1:]], rs[0], rs[1]) == (0, proof_z)
    # refused: an index out of range; an index whose tag is not 2
    with pytest.raises(zkg.ZkgError):
        crs.prove_sparse(tags, np.append(idx, np.uint32(n)), np.vstack([w[idx], w[:1]]), rs[0], rs[1])
    bit = np.uint32(np.flatnonzero(tags == 1)[0])
    with pytest.raises(zkg.ZkgError):
        crs.prove_sparse(tags, np.append(idx, bit), np.vstack([w[idx], w[bit:bit + 1]]), rs[0], rs[1])
    # refused: an index listed twice — with the same value or another one (it used to count z[i] twice in A / B / L and once in H: ZKG_OK
    # with a proof that does not verify); adjacent in one wavefront and far apart
    for dup_at in (0, len(idx) - 1):
        d = idx[dup_at:dup_at + 1]
        with pytest.raises(zkg.ZkgError):
            crs.prove_sparse(tags, np.append(idx, d), np.vstack([w[idx], w[d]]), rs[0], rs[1])
        with pytest.raises(zkg.ZkgError):
            crs.prove_sparse(tags, np.append(d, idx), np.vstack([w[:1], w[idx]]), rs[0], rs[1])
    assert crs.prove_sparse(tags, idx, w[idx], rs[0], rs[1]) == (0, dense)            # the slot is clean afterwards
    crs.free()


def _trivial_system(values):
    """x_i * 1 = x_i for every variable: satisfied by ANY assignment, so the witness can be shaped at will (one public input)"""
    n = len(values)
    rp = np.arange(n + 1, dtype=np.uint32); cols = np.arange(1, n + 1, dtype=np.uint32)
    one = np.tile(arr([1], R), (n, 1))
    A = (rp, cols, one); B = (rp, np.zeros(n, np.uint32), one); C = (rp, cols, one)
    return n, 1, A, B, C, arr(values, R)


@pytest.mark.parametrize("shape", ["all_bits", "no_bits", "repeated_values", "all_zero_but_one"])
def test_prove_witness_split_edge_cases(zkg, oracle, shape):
    """the prover's multi_exp_with_mixed_addition split at its edges: a witness of bits only (nothing reaches the bucket method), one
    with no bit at all (everything does), one whose non-bit values repeat 700 times (one heavy bucket per window, through the gathered
    table path) and an almost empty one — proof bytes against the oracle each time"""
    rng = np.random.default_rng(17)
    n = 3000
    if shape == "all_bits":
        vals = [int(x) for x in rng.integers(0, 2, n)]
    elif shape == "no_bits":
        vals = [int.from_bytes(rng.bytes(31), "little") % (R - 2) + 2 for _ in range(n)]
    elif shape == "repeated_values":
        rep = int.from_bytes(rng.bytes(31), "little") % R
        vals = [rep if i % 4 == 0 else (int(rng.integers(0, 2)) if i % 4 < 3 else int.from_bytes(rng.bytes(31), "little") % R) for i in range(n)]
    else:
        vals = [0] * n; vals[1234] = 5
    n, l, A, B, C, w = _trivial_system(vals)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    assert oracle.r1cs_is_satisfied(ocs, w)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x61))
    rs = random_fr_canonical(2, 0x62)
    rc_o, proof_o = oracle.groth16_prove(oracle.make_pk(ocs, crs_arrays), w, rs[0], rs[1])
    m = crs_arrays["m"]
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, (m - 1).bit_length(), keep, domain_size=m))
    rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc_o == 0 and rc == 0 and proof == proof_o, shape
    one = arr([1], R)[0]
    tags = np.where(~w.any(axis=1), 0, np.where((w == one).all(axis=1), 1, 2)).astype(np.uint8)
    idx = np.flatnonzero(tags == 2).astype(np.uint32)
    rc_s, proof_s = crs.prove_sparse(tags, idx, w[idx], rs[0], rs[1])
    assert rc_s == 0 and proof_s == proof, shape
    crs.free()


def test_witness_tables_grow_with_the_witnesses_seen(zkg, oracle):
    """The witness queries' window tables cover only the elements some proof has sent through the bucket method (subset tables,
    prover.hip subset_extend).  One resident key, a sequence of witnesses whose non-bit positions differ: bits only (no table yet),
    a first set of non-bit positions, a disjoint second set, their union, a witness with the public input and the first / last
    variables non-bit (the L table holds infinity for the constant and the public inputs), and the first one again.  Every proof's
    bytes against the oracle; blob-loaded key as well (queries decompressed on the device)."""
    rng = np.random.default_rng(29)
    n = 2500

    def witness(nonbit_positions):
        vals = [int(x) for x in rng.integers(0, 2, n)]
        for p_ in nonbit_positions:
            vals[p_] = int.from_bytes(rng.bytes(31), "little") % (R - 2) + 2
        return vals

    first = list(range(100, 400, 3)); second = list(range(1000, 1900, 7))
    shapes = [[], first, second, first + second, [0, 1, n - 1], first, list(range(0, 2400, 2)), second]     # (up to 1024 elements the levels are built on the host, above on the GPU)
    n_, l, A, B, C, _ = _trivial_system([0] * n)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x71))
    opk = oracle.make_pk(ocs, crs_arrays)
    m = crs_arrays["m"]
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, (m - 1).bit_length(), keep, domain_size=m))
    crs_blob = zkg.Crs(blob=oracle.pk_write_blob(opk), m=m)
    for j, shape in enumerate(shapes):
        w = arr(witness(shape), R)
        rs = random_fr_canonical(2, 0x72 + j)
        rc_o, proof_o = oracle.groth16_prove(opk, w, rs[0], rs[1])
        assert rc_o == 0
        for c in (crs, crs_blob):
            rc, proof = c.prove(w, rs[0], rs[1])
            assert rc == 0 and proof == proof_o, (j, c is crs_blob)
    crs.free(); crs_blob.free()


def test_concurrent_callers_extend_the_witness_tables(zkg, oracle):
    """Three callers on ONE fresh key (its prover slots), each walking witnesses whose non-bit positions differ from the others': the
    shared witness tables get extended while other proofs are in flight (a proof that must extend them waits for the others and holds
    new ones back, prover.hip zkg_crs).  Every proof's bytes against the oracle's."""
    import threading
    rng = np.random.default_rng(31)
    n = 1200
    n_, l, A, B, C, _ = _trivial_system([0] * n)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x81))
    opk = oracle.make_pk(ocs, crs_arrays)
    m = crs_arrays["m"]
    shapes = [list(range(a, b, st)) for a, b, st in ((0, 90, 3), (100, 400, 5), (400, 1200, 11), (3, 1100, 13), (50, 60, 1), (600, 1199, 2))]
    cases = []
    for j, shape in enumerate(shapes):
        vals = [int(x) for x in rng.integers(0, 2, n)]
        for p_ in shape:
            vals[p_] = int.from_bytes(rng.bytes(31), "little") % (R - 2) + 2
        w = arr(vals, R); rs = random_fr_canonical(2, 0x82 + j)
        rc_o, proof_o = oracle.groth16_prove(opk, w, rs[0], rs[1])
        assert rc_o == 0
        cases.append((w, rs, proof_o))
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, (m - 1).bit_length(), keep, domain_size=m))
    errors = []

    def caller(order):
        try:
            for rep in range(3):
                for j in order:
                    w, rs, expect = cases[j]
                    rc, proof = crs.prove(w, rs[0], rs[1])
                    if rc != 0 or proof != expect:
                        errors.append((j, rc))
        except Exception as e:                                   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=caller, args=(o,)) for o in ([0, 1, 2, 3, 4, 5], [5, 3, 1, 4, 2, 0], [2, 4, 0, 5, 1, 3])]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a caller is stuck"
    assert not errors, errors[:5]
    crs.free()


def test_concurrent_callers_get_the_serial_proofs(zkg, oracle):
    """Re-entrancy of the boundary (SURVEY §8(b) threading: the reference seam is not re-entrant, this one is).  Two resident keys of
    different domain sizes, four host threads proving on them at once (two per key: callers of one key queue on its slot, the two
    keys' pipelines really overlap on the GPU), MSM and NTT calls (host-pointer entry points, same sizes) from two more threads in between.  Every proof must be the byte
    string the same call gives alone; the oracle pins one of them."""
    import threading
    from zklaim_amd import synth
    keep = []
    jobs = []
    for log_m, seed in ((10, 3), (13, 4)):
        n, l, A, B, C, w = synth.zklaim_shaped(log_m, num_inputs=5, seed=seed)
        ocs = oracle.make_r1cs(n, l, A, B, C, keep)
        crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0xC0 + seed))
        crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, log_m, keep))
        rs = [random_fr_canonical(2, 0xD0 + 16 * seed + i) for i in range(6)]
        alone = [crs.prove(w, r[0], r[1]) for r in rs]
        assert all(rc == 0 for rc, _ in alone)
        if log_m == 10:
            rc_o, proof_o = oracle.groth16_prove(oracle.make_pk(ocs, crs_arrays), w, rs[0][0], rs[0][1])
            assert rc_o == 0 and proof_o == alone[0][1]
        jobs.append((crs, w, rs, [p for _, p in alone]))
    from gpu_util import dev_bases_g1
    _, bases, _ = dev_bases_g1(zkg, 3000, 77)
    scalars = random_fr_canonical(3000, 78)
    msm_alone = zkg.msm_g1(bases, scalars)
    vec = random_fr_canonical(1 << 12, 79)
    ntt_alone = zkg.ntt(vec)
    errors = []

    def prover(job, order):
        crs, w, rs, expect = job
        try:
            for rep in range(4):
                for i in order:
                    rc, proof = crs.prove(w, rs[i][0], rs[i][1])
                    if rc != 0 or proof != expect[i]:
                        errors.append(("prove", i, rc))
        except Exception as e:                                   # noqa: BLE001 — a failure on a worker thread must fail the test
            errors.append(("exception", repr(e)))

    def other():
        try:
            for _ in range(8):
                if not np.array_equal(zkg.msm_g1(bases, scalars), msm_alone):
                    errors.append(("msm",))
                if not np.array_equal(zkg.ntt(vec), ntt_alone):
                    errors.append(("ntt",))
        except Exception as e:                                   # noqa: BLE001
            errors.append(("exception", repr(e)))

    threads = [threading.Thread(target=prover, args=(jobs[0], range(6))), threading.Thread(target=prover, args=(jobs[0], range(5, -1, -1))),
               threading.Thread(target=prover, args=(jobs[1], range(6))), threading.Thread(target=prover, args=(jobs[1], range(5, -1, -1))),
               threading.Thread(target=other), threading.Thread(target=other)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a caller is stuck"
    assert not errors, errors[:5]
    for crs, *_ in jobs:
        crs.free()


def test_h_shards_large_enough_to_merge_window_rows(zkg, oracle):
    """2^17 constraints over two shards: 65535 points each, so every shard's launch takes the two-pass sort and pairs its windows' rows of
    buckets (prover.hip H_ROW_MERGE_MIN) — sharding and row merging together, against the unsharded key and the oracle"""
    from zklaim_amd import synth
    log_m = 17
    n, l, A, B, C, w = synth.zklaim_shaped(log_m, num_inputs=5, seed=171)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x5A4B4C41494D0171))
    rs = random_fr_canonical(2, 0x5A4B4C41494D0172)
    rc_o, proof_o = oracle.groth16_prove(oracle.make_pk(ocs, crs_arrays), w, rs[0], rs[1])
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, log_m, keep))
    assert rc_o == 0 and crs.prove(w, rs[0], rs[1]) == (0, proof_o)
    crs.shard_h([0, 0])
    assert crs.prove(w, rs[0], rs[1]) == (0, proof_o)
    crs.shard_h([0, 0, 0])                                           # 43690 points each: below the merge threshold again
    assert crs.prove(w, rs[0], rs[1]) == (0, proof_o)
    crs.free()


@pytest.mark.parametrize("shards", [[0, 0], [0, 0, 0], [0] * 8], ids=["2", "3", "8"])
def test_one_proof_with_the_h_query_sharded(zkg, oracle, shards):
    """SURVEY §8(e), one proof over several devices from C: zkg_crs_shard_h splits the H query by points (a device listed several times
    rehearses the path on a one-GPU box: own table, own stream, coefficients copied per shard, partial points summed on the host).
    Proof bytes must stay those of the unsharded key and of the oracle; sparse witnesses, an unsatisfied witness and callers on
    two threads included."""
    import threading
    from zklaim_amd import synth
    log_m = 13
    n, l, A, B, C, w = synth.zklaim_shaped(log_m, num_inputs=5, seed=91)
    keep = []
    ocs = oracle.make_r1cs(n, l, A, B, C, keep)
    crs_arrays = oracle.groth16_setup(ocs, random_fr_canonical(5, 0x5A4B4C41494D0091))
    rs = [random_fr_canonical(2, 0x5A4B4C41494D0092 + i) for i in range(4)]
    rc_o, proof_o = oracle.groth16_prove(oracle.make_pk(ocs, crs_arrays), w, rs[0][0], rs[0][1])
    assert rc_o == 0
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(n, l, A, B, C, keep), crs_arrays, log_m, keep))
    plain = [crs.prove(w, r[0], r[1]) for r in rs]
    assert plain[0] == (0, proof_o)
    crs.shard_h(shards)
    for r, expect in zip(rs, plain):
        assert crs.prove(w, r[0], r[1]) == expect
        assert crs.prove(w, r[0], r[1], check_satisfied=False) == expect
    bad = w.copy(); bad[-1, 0] ^= np.uint64(1)
    assert crs.prove(bad, rs[0][0], rs[0][1])[0] == zkg.UNSATISFIED
    assert crs.prove(w, rs[1][0], rs[1][1]) == plain[1]          # the slot is clean after the refused witness
    errors = []

    def caller(order):
        for i in order:
            if crs.prove(w, rs[i][0], rs[i][1]) != plain[i]:
                errors.append(i)

    threads = [threading.Thread(target=caller, args=(range(4),)), threading.Thread(target=caller, args=(range(3, -1, -1),))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads) and not errors
    crs.shard_h(shards[:1])                                      # re-sharding replaces the tables; one shard = the whole query on one device
    assert crs.prove(w, rs[2][0], rs[2][1]) == plain[2]
    crs.free()


def test_h_sharding_rejects_bad_arguments(zkg):
    case = CASES[0]
    A, B, C, pts, w, r, s = golden_case_arrays(case)
    keep = []
    crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(case["num_variables"], case["num_inputs"], A, B, C, keep), pts, case["m"].bit_length() - 1, keep))
    with pytest.raises(zkg.ZkgError):
        crs.shard_h([99])                                        # no such device
    with pytest.raises(zkg.ZkgError):
        crs.shard_h([])
    crs.shard_h([0] * 64)                                        # more shards than H points (m - 1 is tiny here): empty shards are skipped
    rc, proof = crs.prove(w, r, s)
    assert rc == 0 and proof.hex() == case["proof_hex"]
    crs.free()


def test_unsatisfied_only_in_a_long_row_is_refused(zkg):
    """The satisfiability gate (snark.cpp:121-124) is evaluated in two places on the GPU: k_r1cs_eval tests a row whose three sides it holds in
    registers, k_r1cs_check_rows the rows with a long side (packing / addition rows, filled in by k_r1cs_long).  A witness bit is flipped that
    keeps every short row it touches satisfied (its booleanity constraint) and breaks a long one — found here with Python integers over the
    CSR matrices — and the prover must refuse it; the untouched witness still proves."""
    keep = []
    pl = dict(attrs=[1994, 7, 42, 0, 5], refs=[2000, 7, 41, 0, 0], ops=["less", "eq", "greater", "noop", "noop"], salt=0x10A6)
    ck = zkg.ZklaimCircuit(zkg.make_ctx([pl], keep))
    assert ck.is_satisfied()
    w = ck.witness()
    mats = ck.csr()
    rinv = pow(1 << 256, -1, R)
    to_int = lambda a: (sum(int(a[i]) << (64 * i) for i in range(4)) * rinv) % R
    one = arr([1], R)[0]
    z = [1] + [to_int(x) for x in w]
    LONG_ROW = 24                                                            # prover.hip
    ncons = ck.r1cs.num_constraints
    is_long = np.zeros(ncons, bool)
    for rp, _, _ in mats:
        is_long |= np.diff(rp.astype(np.int64)) > LONG_ROW
    rows_of = {}
    for rp, col, _ in mats:
        row_index = np.repeat(np.arange(ncons), np.diff(rp.astype(np.int64)))
        for r_, c_ in zip(row_index[is_long[row_index]], col[is_long[row_index]]):
            rows_of.setdefault(int(c_), set()).add(int(r_))                   # variables that occur in a long row
    all_rows_of = {}
    for rp, col, _ in mats:
        row_index = np.repeat(np.arange(ncons), np.diff(rp.astype(np.int64)))
        sel = np.isin(col, np.fromiter(rows_of.keys(), dtype=np.int64))
        for r_, c_ in zip(row_index[sel], col[sel]):
            all_rows_of.setdefault(int(c_), set()).add(int(r_))

    def side(m, row, zz):
        rp, col, val = mats[m]
        return sum(to_int(val[k]) * zz[int(col[k])] for k in range(int(rp[row]), int(rp[row + 1]))) % R

    chosen = None
    for c_ in sorted(rows_of):
        if c_ == 0 or z[c_] not in (0, 1):
            continue
        zz = list(z); zz[c_] = 1 - z[c_]
        ok_short, broke_long = True, False
        for row in all_rows_of[c_]:
            sat = side(0, row, zz) * side(1, row, zz) % R == side(2, row, zz)
            if is_long[row]:
                broke_long |= not sat
            elif not sat:
                ok_short = False
        if ok_short and broke_long:
            chosen = c_
            break
    assert chosen is not None, "no bit whose flip breaks long rows only"
    kp = zkg.Keypair(ck.r1cs, random_fr_canonical(5, 0x10A7))
    crs = zkg.Crs(kp.pk)
    rs = random_fr_canonical(2, 0x10A8)
    rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc == 0
    wbad = w.copy(); wbad[chosen - 1] = 0 if z[chosen] == 1 else one
    assert crs.prove(wbad, rs[0], rs[1])[0] == zkg.UNSATISFIED
    tags, idx, vals = ck.sparse_witness()
    tbad = tags.copy(); tbad[chosen - 1] = 0 if z[chosen] == 1 else 1
    assert crs.prove_sparse(tbad, idx, vals, rs[0], rs[1])[0] == zkg.UNSATISFIED
    rc2, again = crs.prove(w, rs[0], rs[1])
    assert rc2 == 0 and again == proof
    crs.free(); kp.free()
