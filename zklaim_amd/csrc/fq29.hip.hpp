// fq29.hip.hpp — alt_bn128 Fq in 9 limbs of 29 bits for the multi-exponentiation's accumulation kernel (device only).
//
// Replaces libff's Fp_model<4, alt_bn128_modulus_q>::mul_reduce on the hot path of multi_exp's bucket accumulation (reached from
// r1cs_gg_ppzksnark_prover, /root/reference/zklaim/snark.cpp:126).  Same field, same values; only the register representation differs.
//
// Why: on gfx950 every VOP3-encoded vector instruction costs ~4.2 cycles per wavefront and every VOP2 one ~2.3
// (profiles/r3_mul_variants.txt).  The 8 x 32-bit Montgomery product of fp.hip.hpp pays one v_mad_u64_u32 AND one v_addc_co_u32_e64
// (the carry fold, VOP3 because its carry comes from an SGPR pair) per partial product: 1173 cycles.  With 29-bit limbs a 64-bit column
// holds 18 partial products of 58 bits without overflowing, so there is no carry to fold: 162 multiply-adds, 9 quotient digits, 17 column
// shifts — 895 cycles measured, and additions / subtractions are limb-wise VOP2 instructions without any carry chain.
//
// Representation.  R' = 2^261.  An element x is held as X = x R' mod p, as 9 limbs X = sum v[i] 2^(29 i).  "Digits": limbs 0..7 below
// 2^29 (limb 8 holds the rest).  Values are NOT kept below p: a product's result is below p (1 + a b / (p R')) — below 2p whenever
// a b < 169 p^2 — and sums / differences stay unreduced; every use site states the bound it relies on (in units of p) and the limb
// bound that keeps a column sum below 2^64: 9 La Lb + 9 2^58 + 2^36 < 2^64 for operand limbs below La, Lb.
// Subtraction a - b is a + S - b, limb-wise, with S a multiple of p written with every lower limb >= depth (2^29 - 1) ("spread": each limb
// borrows `depth` units from the next), so that no limb goes negative for a subtrahend whose limbs are below depth 2^29 and whose value
// is below (K - 1) p + ...: the constants S{K}_{depth} below.
// Memory format stays libff's (8 x 32-bit limbs, R = 2^256, canonical): to29 / from29 convert with one product each.
#pragma once
#include "curve.hip.hpp"
#include "f29_asm.inc"

namespace zk {

struct Fq29 {
    uint32_t v[9];
    static constexpr uint32_t M = (1u << 29) - 1;
    static ZK_D Fq29 zero() { Fq29 r; for (int i = 0; i < 9; ++i) r.v[i] = 0; return r; }
};
namespace f29 {
// q, -q^-1 mod 2^29, q^-1 mod 2^29
__device__ static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
static constexpr uint32_t INV = 0x04866389u, PINV = 0x1b799c77u;
// R' mod q (the element one); 32 R' mod q (to29: mont(X, TO) = 32 X = x 2^261 for X = x 2^256); 2^256 mod q (from29: mont(X', FROM) = x 2^256)
__device__ static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
__device__ static constexpr uint32_t TO[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
__device__ static constexpr uint32_t FROM[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
// spread multiples of q: S{K}_{d} = K q with limbs 0..7 each increased by d 2^29 and the next limb decreased by d
__device__ static constexpr uint32_t S2_1[9] = {0x30f9fa8eu, 0x2208c16cu, 0x38e5469du, 0x25aa45a0u, 0x2b0bb2efu, 0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu};
__device__ static constexpr uint32_t S4_1[9] = {0x21f3f51cu, 0x241182dau, 0x31ca8d3bu, 0x2b548b42u, 0x361765dfu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u};
__device__ static constexpr uint32_t S6_1[9] = {0x32edefaau, 0x261a4447u, 0x2aafd3d9u, 0x30fed0e4u, 0x212318cfu, 0x31238483u, 0x23e94785u, 0x3628e537u, 0x012259d5u};
__device__ static constexpr uint32_t S4_3[9] = {0x61f3f51cu, 0x641182d8u, 0x71ca8d39u, 0x6b548b40u, 0x761765ddu, 0x6b6d02ffu, 0x629b8501u, 0x797098cdu, 0x00c19136u};

// Montgomery product a b / R' mod q (generated stream, tools/gen_mont_asm.py gen_f29: one 64-bit column accumulator, 162 multiply-adds,
// no carry folds).  Limbs: 9 La Lb < 2^63.8 (digits x digits, digits x one unnormalised sum or difference of digit vectors).
// Values: a b < 169 p^2 gives a result below 2p.  Result: digits.
// (As C++ — acc += (uint64_t) a_i * b_j — the compiler splits every column over two accumulators and joins them with a 64-bit add:
//  895 cycles against ~830 for the stream; with one asm statement per multiply-add it pads each with s_nop.)
ZK_D Fq29 mul(const Fq29 &a, const Fq29 &b) {
    Fq29 t;
    asm(ZK_F29_MUL_ASM
        : "=&v"(t.v[0]), "=&v"(t.v[1]), "=&v"(t.v[2]), "=&v"(t.v[3]), "=&v"(t.v[4]), "=&v"(t.v[5]), "=&v"(t.v[6]), "=&v"(t.v[7]), "=&v"(t.v[8])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(b.v[8])
        : ZK_F29_CLOBBERS);
    return t;
}
// a^2 / R' mod q: the 36 cross products once, against the doubled operand (a's limbs below 2^30: 4 (2 a_i) a_j + a_k^2 + 9 m p < 2^63)
ZK_D Fq29 sqr(const Fq29 &a) {
    Fq29 t; uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.v[i] << 1;
    asm(ZK_F29_SQR_ASM
        : "=&v"(t.v[0]), "=&v"(t.v[1]), "=&v"(t.v[2]), "=&v"(t.v[3]), "=&v"(t.v[4]), "=&v"(t.v[5]), "=&v"(t.v[6]), "=&v"(t.v[7]), "=&v"(t.v[8])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]),
          "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8])
        : ZK_F29_CLOBBERS);
    return t;
}
// two independent products / squarings as ONE interleaved stream: r0 = a b, r1 = c d (r0 = a^2, r1 = c^2).  A lone product is a chain
// of dependent multiply-adds; at the two wavefronts per SIMD the accumulation kernel runs at, their latency shows (measured: the single
// stream no faster than the compiler's two-accumulator code).  The mixed addition's ten products come in five independent pairs.
#define ZK_F29_OUT18(r0, r1) "=&v"(r0.v[0]), "=&v"(r0.v[1]), "=&v"(r0.v[2]), "=&v"(r0.v[3]), "=&v"(r0.v[4]), "=&v"(r0.v[5]), "=&v"(r0.v[6]), "=&v"(r0.v[7]), "=&v"(r0.v[8]), \
                             "=&v"(r1.v[0]), "=&v"(r1.v[1]), "=&v"(r1.v[2]), "=&v"(r1.v[3]), "=&v"(r1.v[4]), "=&v"(r1.v[5]), "=&v"(r1.v[6]), "=&v"(r1.v[7]), "=&v"(r1.v[8])
#define ZK_F29_IN9(a) "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8])
ZK_D void mul2(Fq29 &r0, Fq29 &r1, const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d) {
    asm(ZK_F29_MUL2_ASM : ZK_F29_OUT18(r0, r1) : ZK_F29_IN9(a.v), ZK_F29_IN9(b.v), ZK_F29_IN9(c.v), ZK_F29_IN9(d.v) : ZK_F29_CLOBBERS2);
}
ZK_D void sqr2(Fq29 &r0, Fq29 &r1, const Fq29 &a, const Fq29 &c) {
    uint32_t da[9], dc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { da[i] = a.v[i] << 1; dc[i] = c.v[i] << 1; }
    asm(ZK_F29_SQR2_ASM : ZK_F29_OUT18(r0, r1) : ZK_F29_IN9(a.v), ZK_F29_IN9(da), ZK_F29_IN9(c.v), ZK_F29_IN9(dc) : ZK_F29_CLOBBERS2);
}
// a + b, limb-wise (no carries; the caller's limb bound grows by the sum)
ZK_D Fq29 add(const Fq29 &a, const Fq29 &b) { Fq29 r; for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
ZK_D Fq29 dbl(const Fq29 &a) { Fq29 r; for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] << 1; return r; }
// a + S - b: S one of the spread constants above; b's limbs below depth 2^29 and b's value below what S's top limb covers
ZK_D Fq29 sub(const Fq29 &a, const uint32_t (&S)[9], const Fq29 &b) { Fq29 r; for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + S[i] - b.v[i]; return r; }
ZK_D Fq29 neg(const uint32_t (&S)[9], const Fq29 &b) { Fq29 r; for (int i = 0; i < 9; ++i) r.v[i] = S[i] - b.v[i]; return r; }
// carry propagation: the same value as digits (limbs 0..7 below 2^29).  Limbs below 2^32 on entry; the value below 2^261.
ZK_D Fq29 norm(const Fq29 &a) {
    Fq29 r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t t = a.v[i] + c; r.v[i] = t & Fq29::M; c = t >> 29; }
    r.v[8] = a.v[8] + c;
    return r;
}
// is the value (digits, below 16 p) a multiple of q?  k = v[0] / q mod 2^29 would be that multiple: anything but 0..15 says no at once
// (all but 2^-25 of the non-zero values); the rare candidates are compared limb by limb with k q.
ZK_D bool is_zero_mod_p(const Fq29 &a) {
    const uint32_t k = (a.v[0] * PINV) & Fq29::M;
    if (__builtin_expect(k > 15, 1)) return false;
    uint32_t diff = 0; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { c += (uint64_t)k * P[i]; const uint32_t d = i < 8 ? (uint32_t)c & Fq29::M : (uint32_t)c; diff |= d ^ a.v[i]; c >>= 29; }
    return diff == 0;
}
// libff's memory form (canonical, R = 2^256, 8 x 32 bits) -> digits of x R' (value below 1.01 p), and back (canonical)
ZK_D Fq29 unpack(const Fq &x) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, l = bit >> 5, s = bit & 31;
        uint32_t lo = x.v[l] >> s;
        if (s > 3 && l + 1 < 8) lo |= x.v[l + 1] << (32 - s);
        r.v[i] = i < 8 ? lo & Fq29::M : lo;
    }
    return r;
}
ZK_D Fq29 to29(const Fq &x) { Fq29 t; for (int i = 0; i < 9; ++i) t.v[i] = TO[i]; return mul(unpack(x), t); }
ZK_D Fq from29(const Fq29 &a) {            // a: digits, value below 13 p
    Fq29 f; for (int i = 0; i < 9; ++i) f.v[i] = FROM[i];
    const Fq29 t = mul(a, f);                // x 2^256 mod q, below 2q, digits
    uint32_t w[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {            // 32-bit word l = bits [32 l, 32 l + 32)
        const int i = (32 * l) / 29, s = 32 * l - 29 * i;
        uint32_t v = t.v[i] >> s;
        if (i + 1 < 9) v |= t.v[i + 1] << (29 - s);
        if (s > 26 && i + 2 < 9) v |= t.v[i + 2] << (58 - s);
        w[l] = v;
    }
    return Fq::reduce_once(w);
}


// ---- inversion: constant-time safegcd (Bernstein-Yang divsteps; the layout of libsecp256k1's modinv32: nine signed 30-bit limbs, 30
//      divsteps per round on the low limbs, the 2 x 2 transition matrix applied to (f, g) and — modulo q — to (d, e)).  libff's
//      Fp_model::inverse() is what the reference reaches through to_affine / batch inversion; here the inversion feeds Montgomery's trick in
//      the batched-affine accumulation (msm.hip), where ONE wavefront inverts 64 shared products for its workgroup: a Fermat chain
//      (254 squarings + ~50 products = 300 product-times, 0.13 ms of latency on a lone wavefront) would stall the workgroup for longer than
//      its additions take; 20 rounds of 30 divsteps are ~45 product-times, and every lane finishes in the same round count.
//      Proven bound for 256-bit inputs: 590 divsteps (20 rounds); random inputs are through after 18 — the loop leaves when every lane of
//      the wavefront has g = 0 (further rounds change nothing: with g = 0 a round is the identity on f and d).
__device__ static constexpr int32_t Q30[9] = {0x187cfd47, 0x3082305b, 0x071ca8d3, 0x205aa45a, 0x01585d97, 0x0116da06, 0x1a029b85, 0x139cb84c, 0x3064};
static constexpr uint32_t Q30_INV = 0x1b799c77u;                  // q^-1 mod 2^30
// R'^3 mod q: mont(X^-1, R3) = X^-1 R'^2 = (x R')^-1 R'^2 = x^-1 R', the representation of 1 / x
__device__ static constexpr uint32_t R3[9] = {0x0e2312b2u, 0x16c05ca2u, 0x0bc84389u, 0x1cdf310bu, 0x11adafddu, 0x032e568eu, 0x1d6ae48cu, 0x10d4cd1fu, 0x0026c2d2u};
struct Trans30 { int32_t u, v, q, r; };
ZK_D int32_t divsteps30(int32_t zeta, uint32_t f, uint32_t g, Trans30 &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll 6
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(zeta >> 31);
        const uint32_t c2 = 0u - (g & 1u);
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;
        zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
ZK_D void update_de30(int32_t (&d)[9], int32_t (&e)[9], const Trans30 &t) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d[8] >> 31, se = e[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0], ce = (int64_t)q * d[0] + (int64_t)r * e[0];
    md -= (int32_t)((Q30_INV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((Q30_INV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)Q30[0] * md; ce += (int64_t)Q30[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)Q30[i] * md;
        ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)Q30[i] * me;
        d[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d[8] = (int32_t)cd; e[8] = (int32_t)ce;
}
ZK_D void update_fg30(int32_t (&f)[9], int32_t (&g)[9], const Trans30 &t) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0], cg = (int64_t)q * f[0] + (int64_t)r * g[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cf += (int64_t)u * f[i] + (int64_t)v * g[i];
        cg += (int64_t)q * f[i] + (int64_t)r * g[i];
        f[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f[8] = (int32_t)cf; g[8] = (int32_t)cg;
}
// 1 / a in the same representation.  a: digits, value below 2^261 (anything a product or norm() returns); a multiple of q returns zero.
// Every lane of the wavefront must call it (the early exit is a wavefront vote).
ZK_D Fq29 inverse(const Fq29 &a) {
    constexpr int32_t M30 = 0x3fffffff;
    int32_t g[9], f[9], d[9], e[9];
    // the integer X = sum v[i] 2^(29 i), repacked into 30-bit limbs, then brought below q: X < 2^261 = 169.3 q is too large for the divstep
    // bound, so one product by R' mod q first (mont(X, ONE) = X R' / R' = X mod q, below 2 q) and a conditional subtraction
    Fq29 one; for (int i = 0; i < 9; ++i) one.v[i] = ONE[i];
    const Fq29 xr = mul(a, one);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 30 * i, j = bit / 29, sh = bit - 29 * j;
        uint32_t w = xr.v[j] >> sh;
        if (j + 1 < 9) w |= xr.v[j + 1] << (29 - sh);
        g[i] = i < 8 ? (int32_t)(w & (uint32_t)M30) : (int32_t)w;
    }
    {   // g - q if that is not negative
        int32_t t[9], c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { const int32_t s = g[i] - Q30[i] + c; t[i] = i < 8 ? s & M30 : s; c = s >> 30; }
        const bool neg = t[8] < 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = neg ? g[i] : t[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { f[i] = Q30[i]; d[i] = 0; e[i] = 0; }
    e[0] = 1;
    int32_t zeta = -1;
    for (int it = 0; it < 20; ++it) {
        Trans30 t;
        zeta = divsteps30(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
        update_de30(d, e, t);
        update_fg30(f, g, t);
        if (it >= 15) {
            int32_t any = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) any |= g[i];
            if (__all(any == 0)) break;
        }
    }
    // d = +-1 / X: add q if negative, negate by the sign of f (= +-1), propagate, add q again if needed
    {
        const int32_t add = d[8] >> 31, ngt = f[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; ++i) { int32_t x = d[i] + (Q30[i] & add); d[i] = (x ^ ngt) - ngt; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
        const int32_t add2 = d[8] >> 31;
#pragma unroll
        for (int i = 0; i < 9; ++i) d[i] += Q30[i] & add2;
#pragma unroll
        for (int i = 0; i < 8; ++i) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    }
    // a multiple of q has no inverse: f ends as +-q, d as 0 mod q — return exact zero limbs (callers test for all-zero)
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {                                               // 30-bit limbs -> 29-bit digits
        const int bit = 29 * i, j = bit / 30, sh = bit - 30 * j;
        uint32_t w = (uint32_t)d[j] >> sh;
        if (j + 1 < 9 && sh > 1) w |= (uint32_t)d[j + 1] << (30 - sh);
        r.v[i] = i < 8 ? w & Fq29::M : w;
    }
    Fq29 r3; for (int i = 0; i < 9; ++i) r3.v[i] = R3[i];
    return mul(r, r3);
}

}  // namespace f29

// a base point as the accumulation kernel holds it: x, y as digits of x R', y R' (18 words) and a flag word (1 = infinity)
struct alignas(16) Affine29 { uint32_t x[9], y[9], inf, pad; };              // a base as the accumulation holds it in registers
// ... and as it lies in memory: x R' and y R' (digits below 1.01 p: 255 bits) packed into 8 words each, the infinity flag in the top bit of
// x's last word — ONE 64-byte line per base.  The 80-byte form (the struct above stored as it is) straddled two lines, sometimes three, and
// the accumulation gathers a record per addition: a per-window table of 2^20 points is read at random, 16.8 M times per H query.
struct alignas(64) Rec64 { uint32_t xw[8], yw[8]; };
namespace f29 {
ZK_D void pack8(const Fq29 &t, uint32_t (&w)[8]) {                           // digits -> 32-bit words (as from29 does behind its product)
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const int i = (32 * l) / 29, s = 32 * l - 29 * i;
        uint32_t v = t.v[i] >> s;
        if (i + 1 < 9) v |= t.v[i + 1] << (29 - s);
        if (s > 26 && i + 2 < 9) v |= t.v[i + 2] << (58 - s);
        w[l] = v;
    }
}
ZK_D void unpack8(const uint32_t (&w)[8], uint32_t (&v)[9]) {                // 32-bit words -> digits; bit 255 is not part of the value
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, l = bit >> 5, s = bit & 31;
        uint32_t lo = w[l] >> s;
        if (s > 3 && l + 1 < 8) lo |= w[l + 1] << (32 - s);
        v[i] = i < 8 ? lo & Fq29::M : lo & 0x7fffffu;
    }
}
}  // namespace f29
ZK_D void store_rec64(Rec64 *dst, const Fq29 &x, const Fq29 &y, bool inf) {
    Rec64 o;
    f29::pack8(x, o.xw); f29::pack8(y, o.yw);
    if (inf) o.xw[7] |= 0x80000000u;
    uint4 *d = reinterpret_cast<uint4 *>(dst); const uint4 *s = reinterpret_cast<const uint4 *>(&o);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = s[j];
}
ZK_D Rec64 load_rec64_raw(const Rec64 *src) {                                // the four loads only: the caller unpacks when it USES the record, not when it asks for it
    Rec64 o; uint4 *d = reinterpret_cast<uint4 *>(&o); const uint4 *s = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = s[j];
    return o;
}
ZK_D Affine29 unpack_rec64(const Rec64 &o) {
    Affine29 a;
    f29::unpack8(o.xw, a.x); f29::unpack8(o.yw, a.y);
    a.inf = o.xw[7] >> 31; a.pad = 0;
    return a;
}
ZK_D Affine29 load_rec64(const Rec64 *src) { return unpack_rec64(load_rec64_raw(src)); }

// XYZZ accumulator of the 29-bit path.  Value bounds kept between additions (units of p): X < 5.3, Y < 3.4, ZZ, ZZZ < 1.1; X and Y digits.
struct XYZZ29 {
    Fq29 x, y, zz, zzz;
    // madd-2008-s on the 29-bit representation.  b: digits below 1.01 p, or a difference S2_1 - y (limbs below 2^30, value below 2p) for a
    // negated point.  `inf` is the accumulator's infinity flag (kept in a register instead of a test of ZZ).  Returns false when the
    // addition is one of the exceptional cases (b == +-accumulator): the caller handles those on the 32-bit path.
    ZK_D bool madd(const Fq29 &bx, const Fq29 &by, bool &inf) {
        using namespace f29;
        if (inf) { x = bx; y = norm(by); for (int i = 0; i < 9; ++i) { zz.v[i] = ONE[i]; zzz.v[i] = ONE[i]; } inf = false; return true; }
        Fq29 U2, S2; mul2(U2, S2, bx, zz, by, zzz);                         // < 1.02 each
        const Fq29 Pd = norm(sub(U2, S6_1, x));                             // U2 - X1 + 6p   < 7.1      (x < 5.3: covered by 6p - 2^232)
        const Fq29 Rd = norm(sub(S2, S4_1, y));                             // S2 - Y1 + 4p   < 5.1      (y < 3.4)
        if (__builtin_expect(is_zero_mod_p(Pd), 0)) return false;
        Fq29 PP, RR; sqr2(PP, RR, Pd, Rd);                                  // < 1 + 50.4 / 169 = 1.30;  < 1 + 26.1 / 169 = 1.16
        Fq29 PPP, Q; mul2(PPP, Q, Pd, PP, x, PP);                           // < 1.06, < 1.05
        const Fq29 D = add(PPP, dbl(Q));                                    // < 3.2, limbs below 3 2^29
        x = norm(sub(RR, S4_3, D));                                         // RR - D + 4p    < 5.2
        const Fq29 T = sub(Q, S6_1, x);                                     // Q - X3 + 6p    < 7.1, limbs below 2^30.6 (x digits)
        Fq29 RT, YP; mul2(RT, YP, Rd, T, y, PPP);                           // < 1 + 36.3 / 169 = 1.22;  < 1.03
        y = norm(sub(RT, S2_1, YP));                                        // RT - YP + 2p   < 3.3
        Fq29 z2, z3; mul2(z2, z3, zz, PP, zzz, PPP);                        // < 1.01
        zz = z2; zzz = z3;
        return true;
    }
};

// A bucket accumulator as the 29-bit kernels keep it in memory between accumulation (also from one piece of a piece-wise job to the next)
// and reduction: the four coordinates as digits under XYZZ29's invariants, 144 bytes = nine aligned 16-byte words; infinity: ZZ all zero.
struct alignas(16) Bucket29 { uint32_t x[9], y[9], zz[9], zzz[9]; };
ZK_D void store_bucket29(Bucket29 *dst, const XYZZ29 &a, bool inf) {
    Bucket29 o;
#pragma unroll
    for (int i = 0; i < 9; ++i) { o.x[i] = inf ? 0u : a.x.v[i]; o.y[i] = inf ? 0u : a.y.v[i]; o.zz[i] = inf ? 0u : a.zz.v[i]; o.zzz[i] = inf ? 0u : a.zzz.v[i]; }
    uint4 *d = reinterpret_cast<uint4 *>(dst); const uint4 *s = reinterpret_cast<const uint4 *>(&o);
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = s[i];
}
ZK_D Bucket29 load_bucket29_raw(const Bucket29 *src) {
    Bucket29 o; uint4 *d = reinterpret_cast<uint4 *>(&o); const uint4 *s = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = s[i];
    return o;
}
// the same from / to the 32-bit kernels' XYZZ<Fq> (heavy-bucket path: rare)
ZK_D void store_bucket29(Bucket29 *dst, const XYZZ<Fq> &p) {
    XYZZ29 a; const bool inf = p.is_inf();
    if (!inf) { a.x = f29::to29(p.x.normalized()); a.y = f29::to29(p.y.normalized()); a.zz = f29::to29(p.zz.normalized()); a.zzz = f29::to29(p.zzz.normalized()); }
    else a.x = a.y = a.zz = a.zzz = Fq29::zero();
    store_bucket29(dst, a, inf);
}
ZK_D XYZZ<Fq> bucket29_to_xyzz(const Bucket29 &b) {
    Fq29 x, y, zz, zzz; uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { x.v[i] = b.x[i]; y.v[i] = b.y[i]; zz.v[i] = b.zz[i]; zzz.v[i] = b.zzz[i]; any |= b.zz[i]; }
    if (!any) return XYZZ<Fq>::inf().normalized();
    return {f29::from29(x), f29::from29(y), f29::from29(zz), f29::from29(zzz)};
}

// ---- general addition on the 29-bit representation, shared by the four lanes of a DPP quad (the bucket reduction's chains) ------------
// Same scheme as xyzz_add_quad (curve.hip.hpp): the lanes hold identical copies of both operands, each multiplies a different pair and the
// products are broadcast back — 14 dependent products become 4 rounds — but a round costs a 29-bit product: 930 cycles instead of 1293 at
// the two wavefronts per SIMD these kernels run at, 951 instead of 1524 for a lone wavefront (profiles/r3_mul_variants.txt).
// Invariants of a point (units of p): X < 5.3, Y < 3.4, ZZ, ZZZ < 1.1, all four digits; infinity: ZZ == 0 (all limbs).
struct alignas(16) XYZZ29q {
    Fq29 x, y, zz, zzz;
    ZK_D bool is_inf() const { uint32_t o = 0; for (int i = 0; i < 9; ++i) o |= zz.v[i]; return o == 0; }
    static ZK_D XYZZ29q inf() { XYZZ29q r; r.x = r.y = r.zz = r.zzz = Fq29::zero(); return r; }
};
// Broadcast of lane S of each quad.  STEP names the round of the addition (0 load, 1..4 the four product rounds); bit STEP of
// ZK_F29_SHFL_MASK sends that round through ds_bpermute instead of a DPP move.  Round 4 does by default: with DPP moves there the sum came
// out wrong on hardware (tests/test_gpu_field.py::test_g1_quad_addition_29bit_vs_oracle, bisected round by round with this mask; neither
// wait states behind the product nor pinning the operands changed it), with ds_bpermute every case matches.  27 of them per addition.
#ifndef ZK_F29_SHFL_MASK
#define ZK_F29_SHFL_MASK 16
#endif
template <int S, int STEP = 0> ZK_D Fq29 quad_bcast29(const Fq29 &a) {
    Fq29 r;
    if constexpr ((ZK_F29_SHFL_MASK >> STEP) & 1) { for (int i = 0; i < 9; ++i) r.v[i] = (uint32_t)__shfl((int)a.v[i], (int)((threadIdx.x & 60u) | S), 64); }
    else { for (int i = 0; i < 9; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], S * 0x55, 0xf, 0xf, true); }
    return r;
}
ZK_D Fq29 quad_select29(uint32_t q, const Fq29 &a0, const Fq29 &a1, const Fq29 &a2, const Fq29 &a3) {
    Fq29 r;
    for (int i = 0; i < 9; ++i) { uint32_t lo = q & 1 ? a1.v[i] : a0.v[i], hi = q & 1 ? a3.v[i] : a2.v[i]; r.v[i] = q & 2 ? hi : lo; }
    return r;
}
// the point as the 32-bit kernels store it (canonical XYZZ<Fq>) -> 29-bit, one coordinate per lane of the quad, and back (lane-local)
ZK_D XYZZ29q quad_load29(const Fq &x, const Fq &y, const Fq &zz, const Fq &zzz, uint32_t q) {
    Fq mine;
    for (int i = 0; i < 8; ++i) { uint32_t lo = q & 1 ? y.v[i] : x.v[i], hi = q & 1 ? zzz.v[i] : zz.v[i]; mine.v[i] = q & 2 ? hi : lo; }
    const Fq29 c = f29::to29(mine);
    XYZZ29q r = {quad_bcast29<0, 0>(c), quad_bcast29<1, 0>(c), quad_bcast29<2, 0>(c), quad_bcast29<3, 0>(c)};
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(r.x.v[i]), "+v"(r.y.v[i]), "+v"(r.zz.v[i]), "+v"(r.zzz.v[i]));     // (executed here, quad active: see xyzz29_add_quad)
    return r;
}
// the same addition by one lane (products in independent pairs); invariants as above
ZK_D void xyzz29_add_lane(XYZZ29q &a, const XYZZ29q &b) {
    using namespace f29;
    if (b.is_inf()) return;
    if (a.is_inf()) { a = b; return; }
    Fq29 U1, U2, S1, S2; mul2(U1, U2, a.x, b.zz, b.x, a.zz); mul2(S1, S2, a.y, b.zzz, b.y, a.zzz);
    const Fq29 P = norm(sub(U2, S2_1, U1)), R = norm(sub(S2, S2_1, S1));
    if (__builtin_expect(is_zero_mod_p(P), 0)) {
        if (is_zero_mod_p(R)) {
            XYZZ<Fq> t = {from29(a.x), from29(a.y), from29(a.zz), from29(a.zzz)};
            t = t.dbl();
            a.x = to29(t.x.normalized()); a.y = to29(t.y.normalized()); a.zz = to29(t.zz.normalized()); a.zzz = to29(t.zzz.normalized());
        } else a = XYZZ29q::inf();
        return;
    }
    Fq29 PP, RR; sqr2(PP, RR, P, R);
    Fq29 ZZ12, ZZZ12; mul2(ZZ12, ZZZ12, a.zz, b.zz, a.zzz, b.zzz);
    Fq29 PPP, Q; mul2(PPP, Q, P, PP, U1, PP);
    const Fq29 X3 = norm(sub(RR, S4_3, add(PPP, dbl(Q))));
    const Fq29 T = sub(Q, S6_1, X3);
    Fq29 RT, SP; mul2(RT, SP, T, R, PPP, S1);
    a.x = X3; a.y = norm(sub(RT, S2_1, SP));
    mul2(a.zz, a.zzz, ZZ12, PP, ZZZ12, PPP);
}
ZK_D void xyzz29_add_quad(XYZZ29q &a, const XYZZ29q &b, uint32_t q) {
    using namespace f29;
    if (b.is_inf()) return;                                   // quad-uniform branches: all four lanes see the same data
    if (a.is_inf()) { a = b; return; }
    const Fq29 m1 = mul(quad_select29(q, a.x, b.x, a.y, b.y), quad_select29(q, b.zz, a.zz, b.zzz, a.zzz));
    const Fq29 U1 = quad_bcast29<0, 1>(m1), U2 = quad_bcast29<1, 1>(m1), S1 = quad_bcast29<2, 1>(m1), S2 = quad_bcast29<3, 1>(m1);      // < 1.04 each
    const Fq29 P = norm(sub(U2, S2_1, U1)), R = norm(sub(S2, S2_1, S1));                                                      // < 3.1
    if (__builtin_expect(is_zero_mod_p(P), 0)) {
        if (is_zero_mod_p(R)) {                                // a == b: the doubling, on the 32-bit path (rare)
            XYZZ<Fq> t = {from29(a.x), from29(a.y), from29(a.zz), from29(a.zzz)};
            t = t.dbl();
            a.x = to29(t.x.normalized()); a.y = to29(t.y.normalized()); a.zz = to29(t.zz.normalized()); a.zzz = to29(t.zzz.normalized());
        } else a = XYZZ29q::inf();
        return;
    }
    const Fq29 m2 = mul(quad_select29(q, P, R, a.zz, a.zzz), quad_select29(q, P, R, b.zz, b.zzz));
    const Fq29 PP = quad_bcast29<0, 2>(m2), RR = quad_bcast29<1, 2>(m2), ZZ12 = quad_bcast29<2, 2>(m2), ZZZ12 = quad_bcast29<3, 2>(m2);  // < 1.06, 1.06, 1.01, 1.01
    const Fq29 m3 = mul(quad_select29(q, P, U1, ZZ12, ZZ12), PP);           // lane 3 repeats lane 2's product
    const Fq29 PPP = quad_bcast29<0, 3>(m3), Q = quad_bcast29<1, 3>(m3), ZZ3 = quad_bcast29<2, 3>(m3);                              // < 1.02, 1.01, 1.01
    const Fq29 X3 = norm(sub(RR, S4_3, add(PPP, dbl(Q))));                  // RR - PPP - 2Q + 4p < 5.1
    const Fq29 T = sub(Q, S6_1, X3);                                        // < 7.1, limbs below 2^30.6: one side of the next product
    const Fq29 m4 = mul(quad_select29(q, T, PPP, PPP, PPP), quad_select29(q, R, S1, ZZZ12, ZZZ12));
    a.x = X3; a.y = norm(sub(quad_bcast29<0, 4>(m4), S2_1, quad_bcast29<1, 4>(m4)));                                              // R T - S1 PPP + 2p < 3.2
    a.zz = ZZ3; a.zzz = quad_bcast29<2, 4>(m4);
    // The broadcasts above must execute HERE, with the whole quad active: a caller that only uses the sum in one lane of the quad
    // (if (q == 0) store ...) otherwise lets the compiler sink these DPP moves into that branch, where lanes 1..3 are disabled and a
    // DPP read of a disabled lane returns 0 (seen on hardware: the last round's Y and ZZZ wrong, everything else right).
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(a.y.v[i]), "+v"(a.zzz.v[i]));
}


// ---- the general addition shared by a PAIR of lanes (round 4: the bucket reduction's chains) ---------------------------------------------------------
// The quad form above is bound by ISSUE, not latency (tools/r4_quad_chain.sh: 2.8 us of issue per wavefront-addition however many wavefronts
// share the SIMD): four lanes run 16 products for the 14 an addition needs, and moving operands between them (135 VOP3 selects, 99 DPP moves, 27
// ds_bpermute) costs as much again as the products.  Here lane 0 of a pair OWNS (X, ZZ) of every point and lane 1 owns (Y, ZZZ): the
// first two rounds of the addition are then the SAME code in both lanes on their own registers —
//     m0 = c0 * c1',  m1 = c0' * c1        (lane 0: U1, U2;  lane 1: S1, S2)          D = m1 - m0       (P;  R)
//     sq = D^2,  zz = c1 * c1'             (PP, ZZ12;  RR, ZZZ12)
// — round 3 runs in lane 0 only (PPP = P PP, Q = U1 PP; lane 1 repeats it on its own values and drops the result), and round 4 is one
// paired stream again (lane 0: ZZ3 = ZZ12 PP, ZZZ3 = ZZZ12 PPP;  lane 1: R T, S1 PPP) after four 9-word exchanges (PPP and T to lane 1, RR
// and ZZZ12 to lane 0) and a fifth that brings ZZZ3 home.  14 useful products in four paired streams per lane, 45 DPP moves and 54 one-condition selects
// per addition: about half the quad form's issue per addition at the same depth.  Invariants and value bounds as XYZZ29q; infinity: a lane's
// second coordinate (ZZ resp. ZZZ) all zero — both lanes see it without an exchange.
struct Half29 {
    Fq29 c0, c1;                                                  // lane 0: X, ZZ;  lane 1: Y, ZZZ
    ZK_D bool is_inf() const { uint32_t o = 0; for (int i = 0; i < 9; ++i) o |= c1.v[i]; return o == 0; }
    static ZK_D Half29 inf() { Half29 r; r.c0 = r.c1 = Fq29::zero(); return r; }
};
// the partner lane's copy (DPP quad_perm [1, 0, 3, 2]).  The two wait states a DPP read wants behind the VALU write of its source are
// spelled out: the sources are usually the last writes of a generated product stream, which the compiler's hazard pass cannot see into.
ZK_D Fq29 pair_swap29(const Fq29 &a) {
    Fq29 t = a, r;
    asm volatile("s_nop 1" : "+v"(t.v[0]), "+v"(t.v[1]), "+v"(t.v[2]), "+v"(t.v[3]), "+v"(t.v[4]), "+v"(t.v[5]), "+v"(t.v[6]), "+v"(t.v[7]), "+v"(t.v[8]));
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)t.v[i], 0xB1, 0xf, 0xf, true);
    return r;
}
ZK_D Fq29 pair_pick29(bool second, const Fq29 &a, const Fq29 &b) { Fq29 r; for (int i = 0; i < 9; ++i) r.v[i] = second ? b.v[i] : a.v[i]; return r; }
// a += b.  r: this lane's place in its pair (0 or 1).  Every branch is uniform across the pair (both lanes see the same infinity and
// equality verdicts), so both lanes of a pair are active at every exchange; callers keep pairs whole inside divergent code.
ZK_D void xyzz29_add_pair(Half29 &a, const Half29 &b, uint32_t r) {
    using namespace f29;
    if (b.is_inf()) return;
    if (a.is_inf()) { a = b; return; }
    const bool second = r != 0;
    Fq29 m0, m1; mul2(m0, m1, a.c0, b.c1, b.c0, a.c1);                       // < 1.04 each
    const Fq29 D = norm(sub(m1, S2_1, m0));                                   // P resp. R, < 3.1
    const uint32_t zero_mine = is_zero_mod_p(D) ? 1u : 0u;
    uint32_t zero_other; { uint32_t t = zero_mine; asm volatile("s_nop 1" : "+v"(t)); zero_other = (uint32_t)__builtin_amdgcn_mov_dpp((int)t, 0xB1, 0xf, 0xf, true); }
    const bool p_zero = second ? zero_other != 0 : zero_mine != 0, r_zero = second ? zero_mine != 0 : zero_other != 0;
    if (__builtin_expect(p_zero, 0)) {
        if (r_zero) {                                                          // a == b: the doubling, on the 32-bit path (rare); both lanes rebuild the whole point
            const Fq29 o0 = pair_swap29(a.c0), o1 = pair_swap29(a.c1);
            XYZZ<Fq> t = {from29(second ? o0 : a.c0), from29(second ? a.c0 : o0), from29(second ? o1 : a.c1), from29(second ? a.c1 : o1)};
            t = t.dbl();
            a.c0 = to29((second ? t.y : t.x).normalized()); a.c1 = to29((second ? t.zzz : t.zz).normalized());
        } else a = Half29::inf();
        return;
    }
    Fq29 sq, zz; mul2(sq, zz, D, D, a.c1, b.c1);                              // PP, ZZ12 resp. RR, ZZZ12: < 1.06, 1.01
    Fq29 t0, t1; mul2(t0, t1, D, sq, m0, sq);                                 // lane 0: PPP, Q (< 1.02, 1.01); lane 1: unused
    const Fq29 ppp_in = pair_swap29(t0), rr_in = pair_swap29(sq);             // lane 1 receives PPP, lane 0 receives RR
    const Fq29 X3 = norm(sub(rr_in, S4_3, add(t0, dbl(t1))));                 // lane 0: RR - PPP - 2Q + 4p < 5.1
    const Fq29 T = sub(t1, S6_1, X3);                                         // lane 0: Q - X3 + 6p < 7.1, limbs below 2^30.6
    const Fq29 t_in = pair_swap29(T), zzz12_in = pair_swap29(zz);             // lane 1 receives T, lane 0 receives ZZZ12
    Fq29 n0, n1;
    mul2(n0, n1, pair_pick29(second, zz, D), pair_pick29(second, sq, t_in), pair_pick29(second, zzz12_in, m0), pair_pick29(second, t0, ppp_in));
    // lane 0: ZZ3 = ZZ12 PP, ZZZ3 = ZZZ12 PPP;  lane 1: R T, S1 PPP
    const Fq29 zzz3_in = pair_swap29(n1);                                     // lane 1 receives ZZZ3
    const Fq29 Y3 = norm(sub(n0, S2_1, n1));                                  // lane 1: R T - S1 PPP + 2p < 3.2
    a.c0 = pair_pick29(second, X3, Y3);
    a.c1 = pair_pick29(second, n0, zzz3_in);
    // executed here, the whole pair active (see xyzz29_add_quad: a caller that only uses one lane's sum must not get these moves sunk into its branch)
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(a.c0.v[i]), "+v"(a.c1.v[i]));
}


}  // namespace zk
