// fr29.hip.hpp — alt_bn128 Fr in 9 limbs of 29 bits for the NTT's butterflies (device only; the Fq twin is fq29.hip.hpp).
//
// Replaces libff's Fp_model<4, alt_bn128_modulus_r>::mul_reduce inside libfqfft's _basic_serial_radix2_FFT (reached from
// r1cs_to_qap_witness_map, /root/reference/zklaim/snark.cpp:126).  Why 29 bits: profiles/r3_mul_variants.txt — the 8 x 32-bit product
// pays a VOP3 carry fold per partial product (1173 cycles), the 9 x 29-bit one has no carries to fold (~830), and a butterfly's
// addition and subtraction become limb-wise VOP2 instructions plus one carry propagation each.
//
// Representation: R' = 2^261, limbs v[0..8] ("digits": limbs 0..7 below 2^29).  Values are not reduced between stages: a butterfly is
//     t = v w (below 2r for v w < 169 r^2),   u' = u + t,   v' = u + 2r - t        (2r "spread": every lower limb >= 2^29 - 1)
// so a value grows by at most 2r per stage — below 60 r after the 28 stages the field's 2-adicity allows, far inside 2^261 = 169 r.
// Memory form between passes: 40-byte records (9 limbs + pad, five aligned 8-byte words).  libff's form (canonical, R = 2^256) enters
// by plain bit slicing — read as a 29-bit Montgomery residue it stands for x 2^-5, a constant factor the linear transform carries to
// the output, where slicing back yields x 2^256 again — and leaves through one last product (the post-scaling table, 1/N, or one),
// which brings the value below 2r for the single conditional subtraction.
#pragma once
#include "fp.hip.hpp"
#include "f29_asm.inc"

namespace zk {

struct Fr29 { uint32_t v[9]; static constexpr uint32_t M = (1u << 29) - 1; };
struct alignas(8) Rec29 { uint32_t w[10]; };                 // memory / LDS record of an Fr29

namespace fr29 {
__device__ static constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
__device__ static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};   // R' mod r
__device__ static constexpr uint32_t S2_1[9] = {0x20000002u, 0x3e1f593eu, 0x3cb848a0u, 0x2fa121e5u, 0x2b0ba505u, 0x25b68180u, 0x214dc281u, 0x3cb84c67u, 0x0060c89bu};  // 2r, spread

// a b / R' mod r (generated stream, tools/gen_mont_asm.py gen_f29 with Fr's modulus).  Limbs: 9 La Lb < 2^63.8; a b < 169 r^2 -> below 2r.
ZK_D Fr29 mul(const Fr29 &a, const Fr29 &b) {
    Fr29 t;
    asm(ZK_F29R_MUL_ASM
        : "=&v"(t.v[0]), "=&v"(t.v[1]), "=&v"(t.v[2]), "=&v"(t.v[3]), "=&v"(t.v[4]), "=&v"(t.v[5]), "=&v"(t.v[6]), "=&v"(t.v[7]), "=&v"(t.v[8])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(b.v[8])
        : ZK_F29_CLOBBERS);
    return t;
}
// two independent products, interleaved instruction by instruction (gen_f29_dual with Fr's modulus): a, c limbs up to 2.5 x 2^30
ZK_D void mul2(Fr29 &r0, Fr29 &r1, const Fr29 &a, const Fr29 &b, const Fr29 &c, const Fr29 &d) {
#define ZK_FR29_IN9(x) "v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]), "v"(x.v[4]), "v"(x.v[5]), "v"(x.v[6]), "v"(x.v[7]), "v"(x.v[8])
    asm(ZK_F29R_MUL2_ASM
        : "=&v"(r0.v[0]), "=&v"(r0.v[1]), "=&v"(r0.v[2]), "=&v"(r0.v[3]), "=&v"(r0.v[4]), "=&v"(r0.v[5]), "=&v"(r0.v[6]), "=&v"(r0.v[7]), "=&v"(r0.v[8]),
          "=&v"(r1.v[0]), "=&v"(r1.v[1]), "=&v"(r1.v[2]), "=&v"(r1.v[3]), "=&v"(r1.v[4]), "=&v"(r1.v[5]), "=&v"(r1.v[6]), "=&v"(r1.v[7]), "=&v"(r1.v[8])
        : ZK_FR29_IN9(a), ZK_FR29_IN9(b), ZK_FR29_IN9(c), ZK_FR29_IN9(d)
        : ZK_F29_CLOBBERS2);
#undef ZK_FR29_IN9
}
// limb-wise u + t and u + 2r - t without carries (the radix-4 butterflies normalise once per two stages): t digits below 1.36 r (a product
// of a value below 60 r with a twiddle below r), so its top limb stays below S2_1's; the results' limbs grow by 2^29 resp. 2^30.
ZK_D Fr29 add_lazy(const Fr29 &a, const Fr29 &b) { Fr29 r; for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
ZK_D Fr29 sub_lazy(const Fr29 &a, const Fr29 &b) { Fr29 r; for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + S2_1[i] - b.v[i]; return r; }
ZK_D Fr29 norm(const Fr29 &a) {                               // carry propagation: limbs below 2^32 in, digits out
    Fr29 r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t t = a.v[i] + c; r.v[i] = t & Fr29::M; c = t >> 29; }
    r.v[8] = a.v[8] + c;
    return r;
}
// u + t and u + 2r - t as digits (t: digits below 2r - 2^232; u: digits)
ZK_D Fr29 add_norm(const Fr29 &a, const Fr29 &b) {
    Fr29 r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t t = a.v[i] + b.v[i] + c; r.v[i] = t & Fr29::M; c = t >> 29; }
    r.v[8] = a.v[8] + b.v[8] + c;
    return r;
}
ZK_D Fr29 sub_norm(const Fr29 &a, const Fr29 &b) {
    Fr29 r; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t t = a.v[i] + S2_1[i] - b.v[i] + c; r.v[i] = t & Fr29::M; c = t >> 29; }
    r.v[8] = a.v[8] + S2_1[8] - b.v[8] + c;
    return r;
}
// libff's 256-bit form <-> digits, by bit position (no arithmetic)
ZK_D Fr29 slice(const Fr &x) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, l = bit >> 5, s = bit & 31;
        uint32_t lo = x.v[l] >> s;
        if (s > 3 && l + 1 < 8) lo |= x.v[l + 1] << (32 - s);
        r.v[i] = i < 8 ? lo & Fr29::M : lo;
    }
    return r;
}
ZK_D Fr unslice_reduce(const Fr29 &t) {                      // t: digits, below 2r -> canonical 8 x 32 bits
    uint32_t w[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const int i = (32 * l) / 29, s = 32 * l - 29 * i;
        uint32_t v = t.v[i] >> s;
        if (i + 1 < 9) v |= t.v[i + 1] << (29 - s);
        if (s > 26 && i + 2 < 9) v |= t.v[i + 2] << (58 - s);
        w[l] = v;
    }
    return Fr::reduce_once(w);
}
ZK_D Fr29 load_rec(const Rec29 *p) {
    Fr29 r; const uint2 *q = reinterpret_cast<const uint2 *>(p);
    const uint2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y; r.v[4] = c.x; r.v[5] = c.y; r.v[6] = d.x; r.v[7] = d.y; r.v[8] = e.x;
    return r;
}
ZK_D void store_rec(Rec29 *p, const Fr29 &r) {
    uint2 *q = reinterpret_cast<uint2 *>(p);
    q[0] = make_uint2(r.v[0], r.v[1]); q[1] = make_uint2(r.v[2], r.v[3]); q[2] = make_uint2(r.v[4], r.v[5]); q[3] = make_uint2(r.v[6], r.v[7]); q[4] = make_uint2(r.v[8], 0u);
}
}  // namespace fr29

}  // namespace zk
