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
    dst29 = os.path.join(here, "..", "zklaim_amd", "csrc", "f29_asm.inc")
    l29 = ["// GENERATED by tools/gen_mont_asm.py (gen_f29) — do not edit.  9 x 29-bit Montgomery product / squaring, one 64-bit column, no carry folds.", ""]
    for name, sq, dual in (("MUL", False, False), ("SQR", True, False), ("MUL2", False, True), ("SQR2", True, True)):
        ins = gen_f29_dual(sq) if dual else gen_f29(sq)
        l29.append(f"// {name}: {len(ins)} instructions, {sum(1 for t in ins if t.startswith('v_mad'))} v_mad_u64_u32")
        l29.append(f"#define ZK_F29_{name}_ASM \\")
        for t in ins:
            l29.append(f'    "{t}\\n\\t" \\')
        l29.append('    ""')
        l29.append("")
    ins = gen_f29(False, modulus=F29_R)
    l29.append(f"// Fr MUL: {len(ins)} instructions, {sum(1 for t in ins if t.startswith('v_mad'))} v_mad_u64_u32")
    l29.append("#define ZK_F29R_MUL_ASM \\")
    for t in ins:
        l29.append(f'    "{t}\\n\\t" \\')
    l29.append('    ""')
    l29.append("")
    ins = gen_f29_dual(False, modulus=F29_R)
    l29.append(f"// Fr MUL2: {len(ins)} instructions, {sum(1 for t in ins if t.startswith('v_mad'))} v_mad_u64_u32")
    l29.append("#define ZK_F29R_MUL2_ASM \\")
    for t in ins:
        l29.append(f'    "{t}\\n\\t" \\')
    l29.append('    ""')
    l29.append("")
    l29.append("#define ZK_F29_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(2, 13)] + [f'"s{i}"' for i in range(16, 27)] + ['"vcc"']))
    l29.append("#define ZK_F29_CLOBBERS2 " + ", ".join([f'"v{i}"' for i in range(2, 13)] + [f'"v{i}"' for i in range(14, 25)] + [f'"s{i}"' for i in range(16, 27)] + ['"vcc"']))
    open(dst29, "w").write("\n".join(l29) + "\n")
    print("wrote", os.path.normpath(dst29))





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

# ---------------------------------------------------------------------------------------------------------------------------------
# 9 x 29-bit Montgomery product (R' = 2^261) for csrc/fq29.hip.hpp: product scanning with ONE 64-bit column accumulator.  18 partial
# products of 58 bits fit a column without overflowing, so there is no carry to fold — the v_addc_co_u32_e64 of the 32-bit stream costs
# as much as a multiply-add on gfx950 (every VOP3-encoded instruction ~4.2 cycles, profiles/r3_mul_variants.txt).  Written as C++ the
# compiler splits each column over two accumulators and joins them with a 64-bit add (17 v_lshl_add_u64 per product: 895 cycles); with
# one inline-asm statement per multiply-add it pads every statement with s_nop.  One block for the whole product: 162 v_mad_u64_u32,
# 9 v_mul_lo_u32, 17 v_lshrrev_b64, 17 v_and_b32 (VOP2, mask in an SGPR).
# Fixed registers: column v[2:3], quotient digits v4..v12, q's limbs s16..s24, -1/q mod 2^29 in s25, the mask in s26.
# Operands: %0-%8 result, %9-%17 a, %18-%26 b (for the squaring: b = 2 a, limb-wise; cross products i < j only, against the doubled limb).
F29_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583


F29_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617      # Fr: the NTT's butterflies (csrc/ntt.hip)


def gen_f29(square, col=2, m0=4, out0=0, a0=9, b0=18, consts=True, modulus=None):
    """one product: column accumulator v[col:col+1], quotient digits v{m0}.., operands %{out0}.. result, %{a0}.. a, %{b0}.. b"""
    q = modulus or F29_P
    P = [(q >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
    inv = (-pow(q, -1, 1 << 29)) % (1 << 29)
    A = lambda i: f"%{a0 + i}"
    B = lambda i: f"%{b0 + i}"
    M = lambda i: f"v{m0 + i}"
    SP = lambda j: f"s{16 + j}"
    COL = f"v[{col}:{col + 1}]"
    out = [f"s_mov_b32 s25, 0x{inv:08x}", "s_mov_b32 s26, 0x1fffffff", f"s_mov_b32 {SP(0)}, 0x{P[0]:08x}"] if consts else []
    first = [True]

    def mad(x, y):
        out.append(f"v_mad_u64_u32 {COL}, vcc, {x}, {y}, " + ("0" if first[0] else COL))
        first[0] = False

    def ab_terms(k):
        lo, hi = max(0, k - 8), min(k, 8)
        if not square:
            return [(A(i), B(k - i)) for i in range(lo, hi + 1)]
        t = [(B(i), A(k - i)) for i in range(lo, hi + 1) if i < k - i]             # 2 a_i a_j once, i < j
        if k % 2 == 0:
            t.append((A(k // 2), A(k // 2)))
        return t

    for k in range(17):
        for x, y in ab_terms(k):
            mad(x, y)
        if k < 9:
            for i in range(k):
                mad(M(i), SP(k - i))
            out.append(f"v_mul_lo_u32 {M(k)}, v{col}, s25")
            out.append(f"v_and_b32_e32 {M(k)}, s26, {M(k)}")
            mad(M(k), SP(0))
            if k < 8 and consts:
                out.append(f"s_mov_b32 {SP(k + 1)}, 0x{P[k + 1]:08x}")
        else:
            for i in range(k - 8, 9):
                mad(M(i), SP(k - i))
            out.append(f"v_and_b32_e32 %{out0 + k - 9}, s26, v{col}")
        out.append(f"v_lshrrev_b64 {COL}, 29, {COL}")
    out.append(f"v_mov_b32_e32 %{out0 + 8}, v{col}")
    if consts:
        out.append("s_nop 1")     # a DPP move may read a result right behind the block: two wait states the compiler cannot see into the block for
    return out


def gen_f29_dual(square, modulus=None):
    """two independent products, instruction by instruction: each chain's next multiply-add issues while the other's is in flight (a lone
    column is a chain of dependent v_mad_u64_u32; at two wavefronts per SIMD its latency shows).  Operands: %0-%8 and %9-%17 the results,
    %18-%26 a, %27-%35 b (first product), %36-%44 c, %45-%53 d (second product)."""
    q = modulus or F29_P
    P = [(q >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]
    inv = (-pow(q, -1, 1 << 29)) % (1 << 29)
    head = [f"s_mov_b32 s25, 0x{inv:08x}", "s_mov_b32 s26, 0x1fffffff"] + [f"s_mov_b32 s{16 + j}, 0x{P[j]:08x}" for j in range(9)]
    x = gen_f29(square, col=2, m0=4, out0=0, a0=18, b0=27, consts=False, modulus=modulus)
    y = gen_f29(square, col=14, m0=16, out0=9, a0=36, b0=45, consts=False, modulus=modulus)
    body = []
    for i in range(max(len(x), len(y))):
        if i < len(x):
            body.append(x[i])
        if i < len(y):
            body.append(y[i])
    return head + body + ["s_nop 1"]


def simulate_f29(ins_list, operands, n_out=9):
    """interprets a 29-bit stream on Python integers; operands: {operand index: value}; returns the list of result operands %0..%{n_out-1}"""
    import re
    reg = {f"%{k}": v for k, v in operands.items()}

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

    def put64(tok, r):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        reg[f"v{m.group(1)}"] = r & 0xFFFFFFFF; reg[f"v{m.group(2)}"] = r >> 32
    for text in ins_list:
        op, rest = text.split(" ", 1)
        args = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)]
        if op in ("s_mov_b32", "v_mov_b32_e32"):
            reg[args[0]] = val(args[1]) & 0xFFFFFFFF
        elif op == "v_mad_u64_u32":
            r = val(args[2]) * val(args[3]) + val(args[4])
            assert r < 1 << 64, "column overflow"
            put64(args[0], r)
        elif op == "v_mul_lo_u32":
            reg[args[0]] = (val(args[1]) * val(args[2])) & 0xFFFFFFFF
        elif op == "v_and_b32_e32":
            reg[args[0]] = val(args[1]) & val(args[2])
        elif op == "v_lshrrev_b64":
            put64(args[0], val(args[2]) >> val(args[1]))
        elif op == "s_nop":
            pass
        else:
            raise ValueError(op)
    return [reg[f"%{j}"] for j in range(n_out)]


def selftest_f29():
    import random
    rnd = random.Random(29)
    q = F29_P
    rinv = pow(1 << 261, -1, q)
    limbs = lambda x: [(x >> (29 * i)) & ((1 << 29) - 1) if i < 8 else x >> 232 for i in range(9)]
    value = lambda l: sum(x << (29 * i) for i, x in enumerate(l))
    ops = lambda base, l: {base + i: x for i, x in enumerate(l)}
    mul, sqr, mul2, sqr2 = gen_f29(False), gen_f29(True), gen_f29_dual(False), gen_f29_dual(True)
    mul_r = gen_f29(False, modulus=F29_R)
    rinv_r = pow(1 << 261, -1, F29_R)
    for t in range(200):                                                         # Fr: values as the NTT's butterflies feed them (one side below r, the other below 60 r)
        a = rnd.randrange(60 * F29_R); b = rnd.randrange(F29_R)
        got = value(simulate_f29(mul_r, {**ops(9, limbs(a)), **ops(18, limbs(b))}))
        assert got < 2 * F29_R and got % F29_R == a * b * rinv_r % F29_R, ("mul Fr", t)
    mul2_r = gen_f29_dual(False, modulus=F29_R)
    for t in range(200):                                                         # Fr pairs, the radix-4 butterflies' operands: limbs up to 2.5 x 2^30 on the a side (two lazy stages)
        a = [x + rnd.choice((0, 1 << 30, 3 << 29, 1 << 31)) if i < 8 else x for i, x in enumerate(limbs(rnd.randrange(2 * F29_R)))]
        c = [(5 << 29) - 1] * 8 + [1 << 27] if t == 0 else [x + (1 << 30) if i < 8 else x for i, x in enumerate(limbs(rnd.randrange(50 * F29_R)))]
        b, d = rnd.randrange(F29_R), rnd.randrange(F29_R)
        r = simulate_f29(mul2_r, {**ops(18, a), **ops(27, limbs(b)), **ops(36, c), **ops(45, limbs(d))}, 18)
        assert value(r[:9]) % F29_R == value(a) * b * rinv_r % F29_R and value(r[9:]) % F29_R == value(c) * d * rinv_r % F29_R, ("mul2 Fr", t)
        assert value(r[:9]) < 2 * F29_R and value(r[9:]) < 2 * F29_R and all(x < 1 << 29 for x in r[:8] + r[9:17]), ("mul2 Fr range", t)
    for t in range(300):
        a = rnd.randrange(13 * q) if t > 3 else [0, 1, q - 1, 13 * q - 1][t]
        b = rnd.randrange(13 * q) if t > 3 else [0, q - 1, q - 1, 13 * q - 1][t]
        got = value(simulate_f29(mul, {**ops(9, limbs(a)), **ops(18, limbs(b))}))
        assert got < 2 * q and got % q == a * b * rinv % q, ("mul", t)
        # unnormalised operand: limbs up to 2^30.6 on one side (a difference a + S - b as the mixed addition feeds it)
        big = [min(x + (3 << 28), (1 << 31) - 1) for x in limbs(a % (4 * q))]
        got = value(simulate_f29(mul, {**ops(9, big), **ops(18, limbs(b % (2 * q)))}))
        assert got % q == value(big) * (b % (2 * q)) * rinv % q, ("mul-unnormalised", t)
        a8 = a % (8 * q)
        got = value(simulate_f29(sqr, {**ops(9, limbs(a8)), **ops(18, [2 * x for x in limbs(a8)])}))
        assert got < 2 * q and got % q == a8 * a8 * rinv % q, ("sqr", t)
        # the interleaved pairs: two independent results, each equal to its single stream's
        c, d = rnd.randrange(8 * q), rnd.randrange(8 * q)
        r = simulate_f29(mul2, {**ops(18, limbs(a8)), **ops(27, limbs(b % (8 * q))), **ops(36, limbs(c)), **ops(45, limbs(d))}, 18)
        assert value(r[:9]) % q == a8 * (b % (8 * q)) * rinv % q and value(r[9:]) % q == c * d * rinv % q and value(r[:9]) < 2 * q and value(r[9:]) < 2 * q, ("mul2", t)
        r = simulate_f29(sqr2, {**ops(18, limbs(a8)), **ops(27, [2 * x for x in limbs(a8)]), **ops(36, limbs(c)), **ops(45, [2 * x for x in limbs(c)])}, 18)
        assert value(r[:9]) % q == a8 * a8 * rinv % q and value(r[9:]) % q == c * c * rinv % q, ("sqr2", t)
    print("selftest ok: 29-bit product / squaring streams (single and interleaved pairs) == a*b*2^-261 mod q, no column overflow")


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
        selftest_f29()
