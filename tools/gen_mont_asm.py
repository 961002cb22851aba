#!/usr/bin/env python3
"""Generates zklaim_amd/csrc/mont_asm.inc: the gfx950 instruction stream of one 254-bit Montgomery multiplication.

Finely integrated product scanning (FIPS) over 8 x 32-bit limbs.  Column k keeps a 96-bit accumulator (lo, hi, ex); every
partial product is ONE v_mad_u64_u32 accumulating into the 64-bit-aligned pair (lo:hi) with its carry-out in an SGPR pair,
folded into ex by ONE v_addc_co_u32 issued at least two instructions later (gfx950 needs two wait states between a VALU
writing an SGPR and a VALU reading it; the scheduler below guarantees that and pads with s_nop only when nothing useful can
fill the slot).  gfx950 wants 64-bit VGPR operands even-aligned, so the columns alternate between two pairs X, Y: column k's
ex IS the odd half of the other pair, and moving on costs one v_mov (next lo <- this hi).
128 multiply-adds + 128 carry folds + 8 v_mul_lo_u32 + 23 moves — against 128 + ~470 other instructions from the compiler's
C++ version.

Fixed physical registers (all caller-saved in the AMDGPU calling convention): X = v[2:3], Y = v[4:5], M = v6..v13 (the
Montgomery quotients m_k), modulus limbs + inv in s16..s24.  Operands: %0-%7 result limbs (value in [0, 2p)), %8-%10 carry SGPR pairs,
%11-%18 a, %19-%26 b.
"""
import os

PAIRS, M0, S0 = (2, 4), 6, 16   # even registers of the two accumulator pairs, first VGPR of M, first SGPR of the constants
FIELDS = {
    "FQ": ([0xd87cfd47, 0x3c208c16, 0x6871ca8d, 0x97816a91, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72], 0xe4866389),
    "FR": ([0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72], 0xefffffff),
}


class Ins:
    def __init__(self, text, wr=(), rd=(), salu=False):
        self.text, self.wr, self.rd, self.salu = text, set(wr), set(rd), salu


def gen(P, INV):
    LO = lambda k: f"v{PAIRS[k % 2]}"
    HI = lambda k: f"v{PAIRS[k % 2] + 1}"
    EX = lambda k: f"v{PAIRS[(k + 1) % 2] + 1}"
    TP = lambda k: f"v[{PAIRS[k % 2]}:{PAIRS[k % 2] + 1}]"
    M = lambda k: f"v{M0 + k}"
    SP = lambda j: f"s{S0 + j}"
    SINV = f"s{S0 + 8}"
    A = lambda i: f"%{11 + i}"
    B = lambda i: f"%{19 + i}"
    CARRY = ["%8", "%9", "%10"]
    out = []
    # constants: inv and p0 first, the rest just in time (they double as hazard fillers)
    out.append(Ins(f"s_mov_b32 {SINV}, 0x{INV:08x}", salu=True))
    out.append(Ins(f"s_mov_b32 {SP(0)}, 0x{P[0]:08x}", salu=True))
    pending = []          # (carry operand index, column ex register, first-touch flag holder)
    rot = [0]
    ex_started = {}

    def mad(k, x, y, first_in_mul=False):
        c = rot[0] % 3; rot[0] += 1
        src2 = "0" if first_in_mul else TP(k)
        out.append(Ins(f"v_mad_u64_u32 {TP(k)}, {CARRY[c]}, {x}, {y}, {src2}", wr=[CARRY[c]]))
        pending.append((c, k))

    def fold(n_keep):
        """emit carry folds, leaving the newest n_keep pending"""
        while len(pending) > n_keep:
            c, k = pending.pop(0)
            ex = EX(k)
            if not ex_started.get(k):
                out.append(Ins(f"v_addc_co_u32_e64 {ex}, vcc, 0, 0, {CARRY[c]}", rd=[CARRY[c]], wr=["vcc"]))
                ex_started[k] = True
            else:
                out.append(Ins(f"v_addc_co_u32_e64 {ex}, vcc, 0, {ex}, {CARRY[c]}", rd=[CARRY[c]], wr=["vcc"]))

    for k in range(15):
        terms = []
        if k < 8:
            for i in range(k + 1):
                terms.append((A(i), B(k - i)))
                if i < k:
                    terms.append((M(i), SP(k - i)))
        else:
            for i in range(k - 7, 8):
                terms.append((A(i), B(k - i)))
                terms.append((M(i), SP(k - i)))
        for n, (x, y) in enumerate(terms):
            mad(k, x, y, first_in_mul=(k == 0 and n == 0))
            fold(2)
        if k < 8:
            out.append(Ins(f"v_mul_lo_u32 {M(k)}, {LO(k)}, {SINV}"))
            mad(k, M(k), SP(0))
            fold(2)
            if k < 7:
                out.append(Ins(f"s_mov_b32 {SP(k + 1)}, 0x{P[k + 1]:08x}", salu=True))
        fold(0)           # the next column's first mad reads ex as its high addend: every fold of this column must precede it
        if k >= 8:
            out.append(Ins(f"v_mov_b32 %{k - 8}, {LO(k)}"))
        if k < 14:
            out.append(Ins(f"v_mov_b32 {LO(k + 1)}, {HI(k)}"))          # next lo <- this hi (its hi is this column's ex already)
        else:
            out.append(Ins(f"v_mov_b32 %7, {HI(k)}"))                  # column 15 holds no products: t7 = hi of column 14
    # ---- hazard pass: >= 2 wait states between a VALU that writes an SGPR pair / vcc and a VALU that reads it
    final = []
    for ins in out:
        need = 0
        dist = 0
        for prev in reversed(final):
            if dist >= 2:
                break
            if prev.wr & ins.rd:
                need = max(need, 2 - dist)
            dist += prev.nops if hasattr(prev, "nops") else 1
        if need:
            nop = Ins(f"s_nop {need - 1}"); nop.nops = need
            final.append(nop)
        final.append(ins)
    return final


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "zklaim_amd", "csrc", "mont_asm.inc")
    lines = ["// GENERATED by tools/gen_mont_asm.py — do not edit.  See that script for the register plan and hazard rules.", ""]
    for name, (P, INV) in FIELDS.items():
        ins = gen(P, INV)
        nops = sum(getattr(i, "nops", 0) for i in ins)
        lines.append(f"// {name}: {len(ins)} instructions, {sum(1 for i in ins if i.text.startswith('v_mad'))} v_mad_u64_u32, {nops} wait states of s_nop")
        lines.append(f"#define ZK_MONT_MUL_ASM_{name} \\")
        for i in ins:
            lines.append(f'    "{i.text}\\n\\t" \\')
        lines.append('    ""')
        lines.append("")
    for name, (P, INV) in FIELDS.items():
        for kind, sub in (("ADD", False), ("SUB", True)):
            lines.append(f"// {name} lazy {kind.lower()}: r = a {'-' if sub else '+'} b kept in [0, 2p), 35 issue slots")
            lines.append(f"#define ZK_FP_{kind}_ASM_{name} \\")
            for t in gen_addsub(P, sub):
                lines.append(f'    "{t}\\n\\t" \\')
            lines.append('    ""')
            lines.append("")
    lines.append("#define ZK_FP_ADDSUB_CLOBBERS " + ", ".join(f'"v{T0 + i}"' for i in range(17)) + ', "vcc"')
    clob = [f'"v{PAIRS[0] + i}"' for i in range(4)] + [f'"v{M0 + i}"' for i in range(8)] + [f'"s{S0 + i}"' for i in range(9)] + ['"vcc"']
    lines.append("#define ZK_MONT_MUL_CLOBBERS " + ", ".join(clob))
    open(dst, "w").write("\n".join(lines) + "\n")
    print("wrote", os.path.normpath(dst))





# ---------------------------------------------------------------------------------------------------------------------------------
# Lazy addition / subtraction on [0, 2p): the compiler's version of these 8-limb carry chains costs ~90 instructions, 32 of them
# half-rate 64-bit adds (it avoids v_addc chains because gfx950 wants two wait states between a VALU writing a carry and the VALU
# reading it).  Here two chains are interleaved so that each fills the other's wait states:
#   add:  t = a + b (carry in an SGPR pair)      s = t - 2p (borrow in vcc, literals)      r = borrow ? t : s
#   sub:  t = a - b (borrow in an SGPR pair)     s = t + 2p (carry in vcc, literals)       r = borrow ? s : t
# 35 issue slots, all full rate (the filler slots load the 2p limbs).  Fixed temporaries t = v2..v9, s = v10..v17, v18.  Operands: %0-%7 r, %8 SGPR pair, %9-%16 a, %17-%24 b.
T0, S0R = 2, 10


def gen_addsub(P, sub):
    two_p = sum(x << (32 * i) for i, x in enumerate(P)) * 2
    L = [(two_p >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    T = lambda j: f"v{T0 + j}"
    S = lambda j: f"v{S0R + j}"
    A = lambda j: f"%{9 + j}"
    B = lambda j: f"%{17 + j}"
    K = f"v{S0R + 8}"                               # the 2p limb of the current column (a literal next to a carry-in would be a
    out = [f"v_mov_b32 {K}, 0x{L[0]:08x}"]         # second constant-bus operand, which VOP2/VOP3 do not allow): loaded in the filler slots
    for j in range(8):
        if not sub:
            out.append(f"v_add_co_u32_e64 {T(j)}, %8, {A(j)}, {B(j)}" if j == 0 else f"v_addc_co_u32_e64 {T(j)}, %8, {A(j)}, {B(j)}, %8")
            out.append(f"v_subrev_co_u32_e32 {S(j)}, vcc, {K}, {T(j)}" if j == 0 else f"v_subbrev_co_u32_e32 {S(j)}, vcc, {K}, {T(j)}, vcc")
        else:
            out.append(f"v_sub_co_u32_e64 {T(j)}, %8, {A(j)}, {B(j)}" if j == 0 else f"v_subb_co_u32_e64 {T(j)}, %8, {A(j)}, {B(j)}, %8")
            out.append(f"v_add_co_u32_e32 {S(j)}, vcc, {K}, {T(j)}" if j == 0 else f"v_addc_co_u32_e32 {S(j)}, vcc, {K}, {T(j)}, vcc")
        # third slot of the column: with the other chain's instruction it makes the two wait states a carry needs before it is read again
        out.append(f"v_mov_b32 {K}, 0x{L[j + 1]:08x}" if j < 7 else "s_nop 0")
    out.append("s_nop 0")                          # the selects read the last carries: one more slot
    for j in range(8):
        # add: borrow (vcc) set  <=>  t < 2p  -> keep t.   v_cndmask_b32_e32 D = vcc ? src1 : src0
        # sub: borrow (%8) set   <=>  a < b   -> take s.   v_cndmask_b32_e64 D = mask ? src1 : src0
        out.append(f"v_cndmask_b32_e32 %{j}, {S(j)}, {T(j)}, vcc" if not sub else f"v_cndmask_b32_e64 %{j}, {T(j)}, {S(j)}, %8")
    return out


def simulate_addsub(ins_list, a, b):
    import re
    reg = {}
    M32 = 0xFFFFFFFF
    for i in range(8):
        reg[f"%{9 + i}"] = (a >> (32 * i)) & M32
        reg[f"%{17 + i}"] = (b >> (32 * i)) & M32
    def val(tok):
        tok = tok.strip()
        return int(tok, 16) if tok.startswith("0x") else reg[tok]
    for text in ins_list:
        op, _, rest = text.partition(" ")
        args = [x.strip() for x in rest.split(",")]
        if op == "s_nop":
            continue
        if op == "v_mov_b32":
            reg[args[0]] = val(args[1]); continue
        if op in ("v_add_co_u32_e64", "v_add_co_u32_e32"):
            r = val(args[2]) + val(args[3]); reg[args[0]] = r & M32; reg[args[1]] = r >> 32
        elif op in ("v_addc_co_u32_e64", "v_addc_co_u32_e32"):
            r = val(args[2]) + val(args[3]) + reg[args[4]]; reg[args[0]] = r & M32; reg[args[1]] = r >> 32
        elif op == "v_sub_co_u32_e64":
            r = val(args[2]) - val(args[3]); reg[args[0]] = r & M32; reg[args[1]] = 1 if r < 0 else 0
        elif op == "v_subb_co_u32_e64":
            r = val(args[2]) - val(args[3]) - reg[args[4]]; reg[args[0]] = r & M32; reg[args[1]] = 1 if r < 0 else 0
        elif op == "v_subrev_co_u32_e32":           # D = src1 - src0
            r = val(args[3]) - val(args[2]); reg[args[0]] = r & M32; reg[args[1]] = 1 if r < 0 else 0
        elif op == "v_subbrev_co_u32_e32":
            r = val(args[3]) - val(args[2]) - reg[args[4]]; reg[args[0]] = r & M32; reg[args[1]] = 1 if r < 0 else 0
        elif op in ("v_cndmask_b32_e32", "v_cndmask_b32_e64"):
            reg[args[0]] = val(args[2]) if reg[args[3]] else val(args[1])
        else:
            raise ValueError(op)
    return sum(reg[f"%{j}"] << (32 * j) for j in range(8))


def selftest_addsub():
    import random
    rnd = random.Random(11)
    for name, (P, INV) in FIELDS.items():
        p = sum(x << (32 * i) for i, x in enumerate(P))
        add, sub = gen_addsub(P, False), gen_addsub(P, True)
        edge = [0, 1, p - 1, p, p + 1, 2 * p - 1]
        cases = [(x, y) for x in edge for y in edge] + [(rnd.randrange(2 * p), rnd.randrange(2 * p)) for _ in range(300)]
        for a, b in cases:
            r = simulate_addsub(add, a, b)
            assert r < 2 * p and r % p == (a + b) % p, (name, "add", a, b)
            r = simulate_addsub(sub, a, b)
            assert r < 2 * p and r % p == (a - b) % p, (name, "sub", a, b)
    print("selftest ok: interleaved add / sub streams keep [0, 2p) and agree with (a +- b) mod p for Fq and Fr")

def simulate(ins_list, a, b):
    """Interprets the generated stream on Python ints (one lane).  Returns the 8 result limbs."""
    import re
    reg = {}
    M32 = 0xFFFFFFFF

    def val(tok):
        tok = tok.strip()
        if tok.startswith("0x"):
            return int(tok, 16)
        if tok.isdigit():
            return int(tok)
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return reg.get(f"v{m.group(1)}", 0) | (reg.get(f"v{m.group(2)}", 0) << 32)
        return reg[tok]

    for i in range(8):
        reg[f"%{11 + i}"] = (a >> (32 * i)) & M32
        reg[f"%{19 + i}"] = (b >> (32 * i)) & M32
    for ins in ins_list:
        op, rest = ins.text.split(" ", 1)
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)]
        if op == "s_mov_b32" or op == "v_mov_b32":
            reg[args[0]] = val(args[1]) & M32
        elif op == "v_mad_u64_u32":
            r = val(args[2]) * val(args[3]) + val(args[4])
            m = re.fullmatch(r"v\[(\d+):(\d+)\]", args[0])
            reg[f"v{m.group(1)}"] = r & M32; reg[f"v{m.group(2)}"] = (r >> 32) & M32
            reg[args[1]] = r >> 64
            assert reg[args[1]] in (0, 1)
        elif op == "v_addc_co_u32_e64":
            r = val(args[2]) + val(args[3]) + val(args[4])
            assert r <= M32, "ex overflow"
            reg[args[0]] = r & M32; reg[args[1]] = r >> 32
        elif op == "v_mul_lo_u32":
            reg[args[0]] = (val(args[1]) * val(args[2])) & M32
        elif op == "s_nop":
            pass
        else:
            raise ValueError(op)
    return sum(reg[f"%{j}"] << (32 * j) for j in range(8))


def selftest():
    import random
    rnd = random.Random(7)
    for name, (P, INV) in FIELDS.items():
        p = sum(x << (32 * i) for i, x in enumerate(P))
        assert (-pow(p, -1, 1 << 32)) % (1 << 32) == INV
        ins = gen(P, INV)
        rinv = pow(1 << 256, -1, p)
        for t in range(200):
            a = rnd.randrange(p) if t > 4 else [0, 1, p - 1, p - 1, (1 << 254) - 1 if (1 << 254) - 1 < p else p - 2][t]
            b = rnd.randrange(p) if t > 4 else [0, p - 1, p - 1, 1, p - 1][t]
            got = simulate(ins, a, b)
            assert got < 2 * p and got % p == a * b * rinv % p, (name, t)
    print("selftest ok: simulated instruction stream == a*b*R^-1 mod p for Fq and Fr")


if __name__ == "__main__":
    import sys
    main()
    if "--check" in sys.argv:
        selftest()
        selftest_addsub()
