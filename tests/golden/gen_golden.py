#!/usr/bin/env python3
"""Generates tests/golden/*.json from oracle/pyref.py (pure big-integer definitions).

The reference holds no golden vectors for the prove path (its tests assert
return codes only, with fresh randomness), so these known-answer vectors are authored here
from first-principles arithmetic.  Run:  python tests/golden/gen_golden.py
All integers are canonical (non-Montgomery) values written as hex strings.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as P  # noqa: E402

rnd = random.Random(0x5A4B4C41494D)
H = lambda x: hex(x)


def pt1(p): return None if p is None else [H(p[0]), H(p[1])]
def pt2(p): return None if p is None else [[H(p[0][0]), H(p[0][1])], [H(p[1][0]), H(p[1][1])]]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
    print("wrote", name)


def gen_field():
    out = {"q": H(P.Q), "r": H(P.R), "fr_root_of_unity": H(P.FR_ROOT), "cases": {"fq": [], "fr": [], "fq2": []}}
    for name, p in (("fq", P.Q), ("fr", P.R)):
        edge = [0, 1, 2, p - 1, p - 2, (1 << 253), (1 << 254) % p, P.MONT_R % p]
        vals = edge + [rnd.randrange(p) for _ in range(24)]
        for i in range(len(vals)):
            a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
            out["cases"][name].append(dict(a=H(a), b=H(b), mul=H(a * b % p), add=H((a + b) % p), sub=H((a - b) % p),
                                           neg=H((-a) % p), inv=H(P.inv(a, p)) if a else None, mont=H(P.to_mont(a, p))))
    for _ in range(16):
        a = (rnd.randrange(P.Q), rnd.randrange(P.Q)); b = (rnd.randrange(P.Q), rnd.randrange(P.Q))
        out["cases"]["fq2"].append(dict(a=[H(a[0]), H(a[1])], b=[H(b[0]), H(b[1])], mul=[H(x) for x in P.f2_mul(a, b)],
                                        inv=[H(x) for x in P.f2_inv(a)], sqr=[H(x) for x in P.f2_mul(a, a)]))
    dump("field.json", out)


def gen_curve():
    ks = [1, 2, 3, 5, P.R - 1, P.R - 2, (1 << 128) + 12345, 0] + [rnd.randrange(P.R) for _ in range(8)]
    out = {"g1_mul": [dict(k=H(k), p=pt1(P.g1_mul(k))) for k in ks],
           "g2_mul": [dict(k=H(k), p=pt2(P.g2_mul(k))) for k in ks[:12]], "g1_add": [], "g2_add": []}
    pairs = [(3, 5), (7, 7), (9, P.R - 9), (0, 4), (4, 0), (0, 0)] + [(rnd.randrange(P.R), rnd.randrange(P.R)) for _ in range(4)]
    for a, b in pairs:
        out["g1_add"].append(dict(a=pt1(P.g1_mul(a)), b=pt1(P.g1_mul(b)), sum=pt1(P.g1_mul(a + b))))
        out["g2_add"].append(dict(a=pt2(P.g2_mul(a)), b=pt2(P.g2_mul(b)), sum=pt2(P.g2_mul(a + b))))
    dump("curve.json", out)


def gen_ntt():
    out = []
    for logn in (1, 2, 3, 6, 8):
        n = 1 << logn
        a = [rnd.randrange(P.R) for _ in range(n)]
        if logn == 3:
            a[0], a[1], a[2] = 0, 1, P.R - 1
        naive = logn <= 6
        case = dict(logn=logn, a=[H(x) for x in a], how="naive O(n^2) DFT" if naive else "recursive radix-2")
        for inv_ in (0, 1):
            for coset in (0, 1):
                case[f"out_inv{inv_}_coset{coset}"] = [H(x) for x in P.domain_fft(a, inverse=bool(inv_), coset=bool(coset), naive=naive)]
        out.append(case)
    dump("ntt.json", out)


def gen_msm():
    out = {"g1": [], "g2": []}
    for n, tag in ((1, "single"), (7, "ragged"), (33, "random"), (64, "edge")):
        ks = [rnd.randrange(1, P.R) for _ in range(n)]
        sc = [rnd.randrange(P.R) for _ in range(n)]
        if tag == "edge":
            sc[:8] = [0, 1, P.R - 1, 2, 1, 0, (1 << 253), 1]
            ks[10] = ks[11]                       # duplicate bases (forces the doubling branch when digits match)
            sc[10] = sc[11]
            ks[12] = P.R - ks[13]                 # base and its negation with equal scalars -> cancels
            sc[12] = sc[13]
            ks[20] = 0                            # base at infinity
        b1 = [P.g1_mul(k) for k in ks]
        out["g1"].append(dict(tag=tag, bases=[pt1(b) for b in b1], scalars=[H(s) for s in sc], result=pt1(P.msm_naive(P.Field1, b1, sc))))
        if n <= 33:
            b2 = [P.g2_mul(k) for k in ks]
            out["g2"].append(dict(tag=tag, bases=[pt2(b) for b in b2], scalars=[H(s) for s in sc], result=pt2(P.msm_naive(P.Field2, b2, sc))))
    # everything cancels -> infinity
    b1 = [P.g1_mul(5), P.g1_mul(P.R - 5)]
    out["g1"].append(dict(tag="cancel", bases=[pt1(b) for b in b1], scalars=[H(77), H(77)], result=None))
    out["g1"].append(dict(tag="empty", bases=[], scalars=[], result=None))
    dump("msm.json", out)


def random_r1cs(n_in, n_free, n_mul, bits_frac=0.5):
    """Satisfiable system: variables = inputs | free | one product variable per constraint."""
    l = n_in
    vals = [rnd.randrange(P.R) if rnd.random() > bits_frac else rnd.randrange(2) for _ in range(n_in + n_free)]
    rows = []
    for _ in range(n_mul):
        def lc():
            d = {}
            for _ in range(rnd.randrange(1, 4)):
                d[rnd.randrange(0, len(vals) + 1)] = rnd.choice([1, 1, 2, P.R - 1, rnd.randrange(P.R)])
            return d
        a, b = lc(), lc()
        z = [1] + vals
        ev = lambda d: sum(c * z[i] for i, c in d.items()) % P.R
        vals.append(ev(a) * ev(b) % P.R)
        rows.append((a, b, {len(vals): 1}))
    return P.R1CS(len(vals), l, rows), vals


def groth16_cases(shapes):
    out = []
    for (n_in, n_free, n_mul, tag) in shapes:
        cs, w = random_r1cs(n_in, n_free, n_mul)
        assert cs.is_satisfied(w)
        td = {k: rnd.randrange(1, P.R) for k in ("t", "alpha", "beta", "gamma", "delta")}
        cs2, crs = P.groth16_setup(cs, **td)
        r, s = rnd.randrange(P.R), rnd.randrange(P.R)
        proof = P.groth16_prove(cs2, crs, w, r, s)
        assert P.groth16_check_dlog(cs2, crs, w, r, s, proof=proof, **td)
        h = P.qap_witness_h(cs2, w)
        enc = lambda d: [[str(i), H(c)] for i, c in sorted(d.items())]
        out.append(dict(tag=tag, num_variables=cs2.n, num_inputs=cs2.l, m=crs["m"], domain=cs2.domain().kind,
                        rows=[[enc(a), enc(b), enc(c)] for a, b, c in cs2.rows],
                        trapdoor={k: H(v) for k, v in td.items()}, witness=[H(x) for x in w], r=H(r), s=H(s), h=[H(x) for x in h],
                        crs=dict(alpha_g1=pt1(crs["alpha_g1"]), beta_g1=pt1(crs["beta_g1"]), delta_g1=pt1(crs["delta_g1"]),
                                 beta_g2=pt2(crs["beta_g2"]), delta_g2=pt2(crs["delta_g2"]),
                                 A=[pt1(p) for p in crs["A"]], B1=[pt1(p) for p in crs["B1"]], B2=[pt2(p) for p in crs["B2"]],
                                 H=[pt1(p) for p in crs["H"]], L=[pt1(p) for p in crs["L"]]),
                        proof_points=dict(A=pt1(proof[0]), B=pt2(proof[1]), C=pt1(proof[2])),
                        proof_hex=P.ser_proof(proof).hex()))
        print(tag, "n", cs2.n, "C", len(cs2.rows), "m", crs["m"], cs2.domain().kind)
    return out


def gen_groth16():
    dump("groth16.json", groth16_cases(((1, 2, 5, "tiny"), (2, 5, 13, "m16"), (3, 6, 27, "m32_exact_fill"), (41 % 7, 9, 50, "m64"))))


def gen_step_domain():
    """step_radix2_domain (m = 2^a + 2^b): what libfqfft's get_evaluation_domain picks for 10 of zklaim's 20 payload
    counts.  Everything from oracle/pyref.py's Domain class (points, naive evaluation, Lagrange interpolation)."""
    out = {"rule": [[k, *P.evaluation_domain(k)] for k in list(range(2, 70)) + [82738, 137894, 165472, 275784, 303362, 330940, 551566, (1 << 18) - 3]],
           "fft": [], "lagrange": []}
    for m in (3, 5, 6, 10, 12, 20, 24, 48, 40, 36):
        d = P.Domain.for_size(m)
        assert d.kind == "step" and d.m == m
        a = [rnd.randrange(P.R) for _ in range(m)]
        if m == 12:
            a[0], a[1], a[2], a[11] = 0, 1, P.R - 1, 0
        case = dict(m=m, big=d.big, small=d.small, a=[H(x) for x in a], points=[H(x) for x in d.points])
        for inv_ in (0, 1):
            for coset in (0, 1):
                case[f"out_inv{inv_}_coset{coset}"] = [H(x) for x in d.fft(a, inverse=bool(inv_), coset=bool(coset))]
        out["fft"].append(case)
        t = rnd.randrange(P.R)
        out["lagrange"].append(dict(m=m, t=H(t), u=[H(x) for x in d.lagrange_at(t)], Z=H(d.Z(t))))
    dump("step_domain.json", out)
    # need = C + l + 1: 5+1+1 = 7 -> basic 8 (control); 8+1+1 = 10 -> step 10; 7+2+1=10; 14+2+1 = 17 -> step 17 (16+1);
    # 17+3+1 = 21 -> step 24; 40+2+1 = 43 -> step 48
    dump("groth16_step.json", groth16_cases(((1, 2, 8, "step10"), (2, 3, 14, "step17"), (3, 4, 17, "step24_from21"), (2, 6, 40, "step48_from43"))))


if __name__ == "__main__":
    gen_field(); gen_curve(); gen_ntt(); gen_msm(); gen_groth16(); gen_step_domain()
