// gadgets.hpp — the gadgets zklaim's credential circuit is assembled from, written for r1cs_builder.hpp.
//
// zklaim_gadget (/root/reference/zklaim/zklaim_gadget.cpp:153-784) uses libsnark gadgetlib1's multipacking_gadget,
// digest_variable, comparison_gadget and sha256_compression_function_gadget (zklaim_gadget.cpp:18-20, 442-446, 476-537).
// gadgetlib1 is part of the absent submodule, so these are written from the semantics of those gadgets, not their source:
// the statement each one enforces is the same; the variable numbering and the constraint order of a libsnark-generated key
// are not reproduced (keys made by libsnark and keys made here are not interchangeable).
// Every gadget allocates its variables, emits its constraints and — when the inputs carry witness values — assigns its
// variables in the same call, so the structure never depends on the values.
#pragma once
#include <array>
#include <cstring>
#include "r1cs_builder.hpp"

namespace zk { namespace circuit {

// a boolean wire: a constant, a variable, or the negation of a variable
struct Bit {
    Var v = 0; int8_t konst = 0; bool neg = false;          // konst: -1 variable, 0 / 1 constant
    static Bit zero() { Bit b; b.konst = 0; return b; }
    static Bit one() { Bit b; b.konst = 1; return b; }
    static Bit var(Var v) { Bit b; b.v = v; b.konst = -1; return b; }
    bool is_const() const { return konst >= 0; }
    Bit operator!() const { Bit b = *this; if (is_const()) b.konst = 1 - konst; else b.neg = !neg; return b; }
    LC lc() const { if (is_const()) return LC::constant((uint64_t)konst); return neg ? LC::constant(1) - LC(v) : LC(v); }
    bool value(const Builder &pb) const { if (is_const()) return konst != 0; bool x = pb.is_nonzero(v); return neg ? !x : x; }
};

inline Bit new_bit(Builder &pb, bool value) { Var v = pb.alloc(); pb.set_bit(v, value); return Bit::var(v); }

inline Bit bit_xor(Builder &pb, Bit a, Bit b) {
    if (a.is_const()) return a.konst ? !b : b;
    if (b.is_const()) return b.konst ? !a : a;
    Bit r = new_bit(pb, a.value(pb) != b.value(pb));
    if (pb.recording) pb.enforce(a.lc() * 2, b.lc(), a.lc() + b.lc() - r.lc());                 // 2ab = a + b - r
    return r;
}
inline Bit bit_and(Builder &pb, Bit a, Bit b) {
    if (a.is_const()) return a.konst ? b : Bit::zero();
    if (b.is_const()) return b.konst ? a : Bit::zero();
    Bit r = new_bit(pb, a.value(pb) && b.value(pb));
    if (pb.recording) pb.enforce(a.lc(), b.lc(), r.lc());
    return r;
}
inline Bit bit_or(Builder &pb, Bit a, Bit b) { return !bit_and(pb, !a, !b); }
inline Bit bit_xor3(Builder &pb, Bit a, Bit b, Bit c) { return bit_xor(pb, bit_xor(pb, a, b), c); }
// ch(e, f, g) = e ? f : g
inline Bit bit_choice(Builder &pb, Bit e, Bit f, Bit g) {
    if (e.is_const()) return e.konst ? f : g;
    if (f.is_const() && g.is_const()) { if (f.konst == g.konst) return f; return f.konst ? e : !e; }
    Bit r = new_bit(pb, e.value(pb) ? f.value(pb) : g.value(pb));
    if (pb.recording) pb.enforce(e.lc(), f.lc() - g.lc(), r.lc() - g.lc());                     // e (f - g) = r - g
    return r;
}
inline Bit bit_majority(Builder &pb, Bit a, Bit b, Bit c) {
    if (a.is_const()) return a.konst ? bit_or(pb, b, c) : bit_and(pb, b, c);
    if (b.is_const()) return b.konst ? bit_or(pb, a, c) : bit_and(pb, a, c);
    if (c.is_const()) return c.konst ? bit_or(pb, a, b) : bit_and(pb, a, b);
    Bit t = bit_and(pb, a, b);
    int s = (int)a.value(pb) + (int)b.value(pb) + (int)c.value(pb);
    Bit r = new_bit(pb, s >= 2);
    if (pb.recording) pb.enforce(c.lc(), a.lc() + b.lc() - t.lc() * 2, r.lc() - t.lc());         // r = ab + c (a + b - 2ab)
    return r;
}

// ---- 32-bit words, index 0 = most significant bit (SHA-256's big-endian convention, and the order memtobv
//      (libsnark_wrapper.cpp:65-74) produces for the bytes of a block)
typedef std::array<Bit, 32> Word;
inline Word word_const(uint32_t x) { Word w; for (int i = 0; i < 32; ++i) w[i] = ((x >> (31 - i)) & 1) ? Bit::one() : Bit::zero(); return w; }
inline Word rotr(const Word &w, int n) { Word o; for (int i = 0; i < 32; ++i) o[i] = w[(i - n + 32) % 32]; return o; }
inline Word shr(const Word &w, int n) { Word o; for (int i = 0; i < 32; ++i) o[i] = i >= n ? w[i - n] : Bit::zero(); return o; }
inline LC word_lc(const Word &w) { LC l; for (int i = 0; i < 32; ++i) l = l + w[i].lc() * ((uint64_t)1 << (31 - i)); return l; }
inline uint32_t word_value(const Builder &pb, const Word &w) { uint32_t x = 0; for (int i = 0; i < 32; ++i) x |= (uint32_t)w[i].value(pb) << (31 - i); return x; }
inline Word word_xor3(Builder &pb, const Word &a, const Word &b, const Word &c) { Word o; for (int i = 0; i < 32; ++i) o[i] = bit_xor3(pb, a[i], b[i], c[i]); return o; }

// sum of words (+ constant) mod 2^32.  The low 32 result bits are fresh boolean variables, or `out` when given (variables
// whose booleanity is enforced elsewhere); the carry bits are fresh and boolean-constrained here.
inline Word add_mod32(Builder &pb, const std::vector<Word> &terms, uint32_t konst, const Var *out = nullptr) {
    bool all_const = true;
    for (auto &w : terms) for (auto &b : w) all_const = all_const && b.is_const();
    uint64_t sum = konst; LC s;
    for (auto &w : terms) sum += word_value(pb, w);
    if (all_const && !out) return word_const((uint32_t)sum);
    const bool rec = pb.recording;
    if (rec) { s = LC::constant((uint64_t)konst); for (auto &w : terms) s = s + word_lc(w); }
    int extra = 0; while (((uint64_t)(terms.size() + 1) << 32) > ((uint64_t)1 << (32 + extra))) ++extra;   // enough carry bits for the worst case
    Word r; LC packed;
    for (int i = 0; i < 32; ++i) {                                             // weight 2^(31-i)
        bool bv = (sum >> (31 - i)) & 1;
        if (out) { r[i] = Bit::var(out[i]); }
        else { r[i] = new_bit(pb, bv); pb.enforce_boolean(r[i].v); }
        if (rec) packed = packed + r[i].lc() * ((uint64_t)1 << (31 - i));
    }
    for (int j = 0; j < extra; ++j) {
        Bit c = new_bit(pb, (sum >> (32 + j)) & 1); pb.enforce_boolean(c.v);
        if (rec) packed = packed + c.lc() * ((uint64_t)1 << (32 + j));
    }
    if (rec) pb.enforce(LC::constant(1), s, packed);
    return r;
}

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t SHA256_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

// SHA-256 compression of one 512-bit block from the default IV (zklaim hashes one block per payload:
// 384 pre-image bits + the fixed padding, zklaim_gadget.cpp:33-36,476-497).  digest_out: 256 boolean variables (big-endian
// bit order per word) that the final additions are written into.
inline void sha256_compress_from_iv(Builder &pb, const std::vector<Bit> &block /*512*/, const std::vector<Var> &digest_out /*256*/) {
    std::vector<Word> W(64);
    for (int t = 0; t < 16; ++t) for (int i = 0; i < 32; ++i) W[t][i] = block[32 * t + i];
    for (int t = 16; t < 64; ++t) {
        Word s0 = word_xor3(pb, rotr(W[t - 15], 7), rotr(W[t - 15], 18), shr(W[t - 15], 3));
        Word s1 = word_xor3(pb, rotr(W[t - 2], 17), rotr(W[t - 2], 19), shr(W[t - 2], 10));
        W[t] = add_mod32(pb, {W[t - 16], s0, W[t - 7], s1}, 0);
    }
    Word r[8];
    for (int j = 0; j < 8; ++j) r[j] = word_const(SHA256_IV[j]);
    for (int t = 0; t < 64; ++t) {
        const Word &a = r[0], &b = r[1], &c = r[2], &d = r[3], &e = r[4], &f = r[5], &g = r[6], &h = r[7];
        Word S1 = word_xor3(pb, rotr(e, 6), rotr(e, 11), rotr(e, 25));
        Word ch; for (int i = 0; i < 32; ++i) ch[i] = bit_choice(pb, e[i], f[i], g[i]);
        Word S0 = word_xor3(pb, rotr(a, 2), rotr(a, 13), rotr(a, 22));
        Word mj; for (int i = 0; i < 32; ++i) mj[i] = bit_majority(pb, a[i], b[i], c[i]);
        Word new_e = add_mod32(pb, {d, h, S1, ch, W[t]}, SHA256_K[t]);
        Word new_a = add_mod32(pb, {h, S1, ch, W[t], S0, mj}, SHA256_K[t]);
        for (int j = 7; j >= 1; --j) r[j] = r[j - 1];
        r[4] = new_e; r[0] = new_a;
    }
    for (int j = 0; j < 8; ++j) add_mod32(pb, {r[j]}, SHA256_IV[j], &digest_out[32 * j]);
}

// packing: 1 * sum_i 2^i bits[i] = packed      (libsnark packing_gadget; little-endian over the given order)
inline void enforce_packing(Builder &pb, const std::vector<Var> &bits, size_t lo, size_t hi, Var packed, bool enforce_bitness) {
    if (!pb.recording) return;
    LC s; Fr w = Fr::one();
    for (size_t i = lo; i < hi; ++i) { if (enforce_bitness) pb.enforce_boolean(bits[i]); s.add(bits[i], w); w = w.dbl(); }
    pb.enforce(LC::constant(1), s, LC(packed));
}
inline void assign_packing(Builder &pb, const std::vector<Var> &bits, size_t lo, size_t hi, Var packed) {
    Fr s = Fr::zero(), w = Fr::one();
    for (size_t i = lo; i < hi; ++i) { if (pb.is_nonzero(bits[i])) s += w; w = w.dbl(); }
    pb.set(packed, s);
}

// comparison of two n-bit values (libsnark comparison_gadget semantics): less = [A < B], less_or_eq = [A <= B].
//   alpha = 2^n + B - A decomposed into n+1 bits; less_or_eq = alpha[n]; less = less_or_eq * OR(alpha[0..n-1])
struct Comparison { std::vector<Var> alpha; Var alpha_packed, not_all_zeros, inv; };
inline Comparison comparison_alloc(Builder &pb, size_t n, Var less_or_eq) {
    Comparison c; c.alpha = pb.alloc_n(n); c.alpha.push_back(less_or_eq);
    c.alpha_packed = pb.alloc(); c.not_all_zeros = pb.alloc(); c.inv = pb.alloc();
    return c;
}
inline void comparison_constraints(Builder &pb, const Comparison &c, size_t n, Var A, Var B, Var less, Var less_or_eq) {
    if (!pb.recording) return;
    enforce_packing(pb, c.alpha, 0, n + 1, c.alpha_packed, true);
    Fr two_n = Fr::one(); for (size_t i = 0; i < n; ++i) two_n = two_n.dbl();
    pb.enforce(LC::constant(1), LC::constant(two_n) + LC(B) - LC(A), LC(c.alpha_packed));
    LC sum; for (size_t i = 0; i < n; ++i) sum = sum + LC(c.alpha[i]);
    pb.enforce(LC(c.inv), sum, LC(c.not_all_zeros));                           // disjunction: inv * sum = out
    pb.enforce(LC::constant(1) - LC(c.not_all_zeros), sum, LC());              //              (1 - out) * sum = 0
    pb.enforce(LC(less_or_eq), LC(c.not_all_zeros), LC(less));
}
inline void comparison_witness(Builder &pb, const Comparison &c, size_t n, uint64_t a, uint64_t b, Var less, Var less_or_eq) {
    // n == 64: alpha = 2^64 + b - a as a 65-bit integer
    unsigned __int128 alpha = ((unsigned __int128)1 << n) + b - a;
    uint64_t cnt = 0;
    for (size_t i = 0; i <= n; ++i) { bool bit = (alpha >> i) & 1; pb.set_bit(c.alpha[i], bit); if (i < n && bit) ++cnt; }
    Fr packed = Fr::from_u64((uint64_t)alpha);
    if ((alpha >> 64) & 1) { Fr t = Fr::one(); for (int i = 0; i < 64; ++i) t = t.dbl(); packed += t; }
    pb.set(c.alpha_packed, packed);
    pb.set_bit(c.not_all_zeros, cnt != 0);
    pb.set(c.inv, cnt ? Fr::from_u64(cnt).inverse() : Fr::zero());
    bool leq = (alpha >> n) & 1;
    pb.set_bit(less_or_eq, leq);
    pb.set_bit(less, leq && cnt);
}

}}  // namespace zk::circuit
