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
#include <initializer_list>
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
    bool value(const uint8_t *tags) const { return is_const() ? konst != 0 : ((tags[v] != 0) != neg); }      // tags: Builder::tag_data()
};

inline Bit new_bit(Builder &pb, bool value) { Var v = pb.alloc(); pb.set_bit(v, value); return Bit::var(v); }
// The witness-only pass (one per proof, on the seam's critical path) evaluates the same gates in the same order, but reads and writes
// the one-byte tags directly instead of going through the builder for every bit: 27 K gates per payload.
inline Bit new_bit_w(Builder &pb, bool value) { Var v = pb.alloc(); pb.tag_data()[v] = value; return Bit::var(v); }     // (alloc may move the storage: pointer taken after it)

inline Bit bit_xor(Builder &pb, Bit a, Bit b) {
    if (a.is_const()) return a.konst ? !b : b;
    if (b.is_const()) return b.konst ? !a : a;
    if (!pb.recording) { const uint8_t *tags = pb.tag_data(); return new_bit_w(pb, a.value(tags) != b.value(tags)); }
    Bit r = new_bit(pb, a.value(pb) != b.value(pb));
    pb.enforce(a.lc() * 2, b.lc(), a.lc() + b.lc() - r.lc());                                   // 2ab = a + b - r
    return r;
}
inline Bit bit_and(Builder &pb, Bit a, Bit b) {
    if (a.is_const()) return a.konst ? b : Bit::zero();
    if (b.is_const()) return b.konst ? a : Bit::zero();
    if (!pb.recording) { const uint8_t *tags = pb.tag_data(); return new_bit_w(pb, a.value(tags) && b.value(tags)); }
    Bit r = new_bit(pb, a.value(pb) && b.value(pb));
    pb.enforce(a.lc(), b.lc(), r.lc());
    return r;
}
inline Bit bit_or(Builder &pb, Bit a, Bit b) { return !bit_and(pb, !a, !b); }
inline Bit bit_xor3(Builder &pb, Bit a, Bit b, Bit c) { return bit_xor(pb, bit_xor(pb, a, b), c); }
// ch(e, f, g) = e ? f : g
inline Bit bit_choice(Builder &pb, Bit e, Bit f, Bit g) {
    if (e.is_const()) return e.konst ? f : g;
    if (f.is_const() && g.is_const()) { if (f.konst == g.konst) return f; return f.konst ? e : !e; }
    if (!pb.recording) { const uint8_t *tags = pb.tag_data(); return new_bit_w(pb, e.value(tags) ? f.value(tags) : g.value(tags)); }
    Bit r = new_bit(pb, e.value(pb) ? f.value(pb) : g.value(pb));
    pb.enforce(e.lc(), f.lc() - g.lc(), r.lc() - g.lc());                                       // e (f - g) = r - g
    return r;
}
inline Bit bit_majority(Builder &pb, Bit a, Bit b, Bit c) {
    if (a.is_const()) return a.konst ? bit_or(pb, b, c) : bit_and(pb, b, c);
    if (b.is_const()) return b.konst ? bit_or(pb, a, c) : bit_and(pb, a, c);
    if (c.is_const()) return c.konst ? bit_or(pb, a, b) : bit_and(pb, a, b);
    Bit t = bit_and(pb, a, b);
    if (!pb.recording) { const uint8_t *tags = pb.tag_data(); return new_bit_w(pb, (int)a.value(tags) + (int)b.value(tags) + (int)c.value(tags) >= 2); }
    int s = (int)a.value(pb) + (int)b.value(pb) + (int)c.value(pb);
    Bit r = new_bit(pb, s >= 2);
    pb.enforce(c.lc(), a.lc() + b.lc() - t.lc() * 2, r.lc() - t.lc());                           // r = ab + c (a + b - 2ab)
    return r;
}

// ---- 32-bit words, index 0 = most significant bit (SHA-256's big-endian convention, and the order memtobv
//      (libsnark_wrapper.cpp:65-74) produces for the bytes of a block)
typedef std::array<Bit, 32> Word;
inline Word word_const(uint32_t x) { Word w; for (int i = 0; i < 32; ++i) w[i] = ((x >> (31 - i)) & 1) ? Bit::one() : Bit::zero(); return w; }
inline Word rotr(const Word &w, int n) { Word o; for (int i = 0; i < 32; ++i) o[i] = w[(i - n) & 31]; return o; }
inline Word shr(const Word &w, int n) { Word o; for (int i = 0; i < 32; ++i) o[i] = i >= n ? w[i - n] : Bit::zero(); return o; }
inline LC word_lc(const Word &w) { LC l; for (int i = 0; i < 32; ++i) l = l + w[i].lc() * ((uint64_t)1 << (31 - i)); return l; }
inline uint32_t word_value(const Builder &pb, const Word &w) { const uint8_t *tags = pb.tag_data(); uint32_t x = 0; for (int i = 0; i < 32; ++i) x |= (uint32_t)w[i].value(tags) << (31 - i); return x; }
inline bool word_plain(const Word &w) { for (auto &b : w) if (b.is_const()) return false; return true; }                     // 32 variables, no constant
// Witness-only word operations.  When every input bit is a variable the gates of a word allocate their outputs in a fixed pattern (two per
// bit for a three-way xor and for a majority, one for a choice), so the word is evaluated on native 32-bit values and its outputs are
// written as one block of tags — same variables, same order, same values as the gate-by-gate path, which the recording pass and any
// word with a constant bit (shifted-in zeros, the IV, the padding) still take.
inline Word word_xor3(Builder &pb, const Word &a, const Word &b, const Word &c) {
    Word o;
    if (!pb.recording && word_plain(a) && word_plain(b)) {
        // a, b variables; c variables or constants (the zeros a shift brings in): per bit t = a xor b is a fresh variable, and t xor c is
        // another one unless c is constant there (then it is t or its negation)
        const uint32_t A = word_value(pb, a), B = word_value(pb, b), T = A ^ B, Rv = T ^ word_value(pb, c);
        uint32_t nvar_c = 0;
        for (auto &x : c) nvar_c += !x.is_const();
        Var pos = pb.alloc_block(32 + nvar_c);
        uint8_t *tags = pb.tag_data();
        for (int i = 0; i < 32; ++i) {
            tags[pos] = (T >> (31 - i)) & 1;
            const Var t = pos++;
            if (c[i].is_const()) { o[i] = Bit::var(t); if (c[i].konst) o[i] = !o[i]; }
            else { tags[pos] = (Rv >> (31 - i)) & 1; o[i] = Bit::var(pos++); }
        }
        return o;
    }
    for (int i = 0; i < 32; ++i) o[i] = bit_xor3(pb, a[i], b[i], c[i]);
    return o;
}
inline Word word_choice(Builder &pb, const Word &e, const Word &f, const Word &g) {
    Word o;
    if (!pb.recording && word_plain(e) && word_plain(f) && word_plain(g)) {
        const uint32_t E = word_value(pb, e), Rv = (E & word_value(pb, f)) | (~E & word_value(pb, g));
        const Var first = pb.alloc_block(32);
        uint8_t *tags = pb.tag_data();
        for (int i = 0; i < 32; ++i) { tags[first + i] = (Rv >> (31 - i)) & 1; o[i] = Bit::var(first + i); }
        return o;
    }
    for (int i = 0; i < 32; ++i) o[i] = bit_choice(pb, e[i], f[i], g[i]);
    return o;
}
inline Word word_majority(Builder &pb, const Word &a, const Word &b, const Word &c) {
    Word o;
    if (!pb.recording && word_plain(a) && word_plain(b) && word_plain(c)) {
        const uint32_t A = word_value(pb, a), B = word_value(pb, b), C = word_value(pb, c), T = A & B, Rv = T | (C & (A ^ B));
        const Var first = pb.alloc_block(64);                                   // per bit: t = a AND b, then the majority
        uint8_t *tags = pb.tag_data();
        for (int i = 0; i < 32; ++i) { tags[first + 2 * i] = (T >> (31 - i)) & 1; tags[first + 2 * i + 1] = (Rv >> (31 - i)) & 1; o[i] = Bit::var(first + 2 * i + 1); }
        return o;
    }
    for (int i = 0; i < 32; ++i) o[i] = bit_majority(pb, a[i], b[i], c[i]);
    return o;
}

// sum of words (+ constant) mod 2^32.  The low 32 result bits are fresh boolean variables, or `out` when given (variables
// whose booleanity is enforced elsewhere); the carry bits are fresh and boolean-constrained here.
inline Word add_mod32(Builder &pb, std::initializer_list<const Word *> terms, uint32_t konst, const Var *out = nullptr) {
    bool all_const = true;
    uint64_t sum = konst; LC s;
    {   // value and constness of every term in one pass over its bits
        const uint8_t *tags = pb.tag_data();
        for (const Word *w : terms) {
            uint32_t x = 0;
            for (int i = 0; i < 32; ++i) { const Bit &b = (*w)[i]; all_const = all_const && b.is_const(); x |= (uint32_t)b.value(tags) << (31 - i); }
            sum += x;
        }
    }
    if (all_const && !out) return word_const((uint32_t)sum);
    const bool rec = pb.recording;
    if (rec) { s = LC::constant((uint64_t)konst); for (const Word *w : terms) s = s + word_lc(*w); }
    int extra = 0; while (((uint64_t)(terms.size() + 1) << 32) > ((uint64_t)1 << (32 + extra))) ++extra;   // enough carry bits for the worst case
    Word r; LC packed;
    if (!rec) {                                                                // witness-only: result bits (unless given) and carries as one block of tags
        const uint32_t nres = out ? 0 : 32;
        const Var first = pb.alloc_block(nres + (uint32_t)extra);
        uint8_t *tags = pb.tag_data();
        for (int i = 0; i < 32; ++i) {
            if (out) r[i] = Bit::var(out[i]);
            else { tags[first + i] = (sum >> (31 - i)) & 1; r[i] = Bit::var(first + i); }
        }
        for (int j = 0; j < extra; ++j) tags[first + nres + j] = (sum >> (32 + j)) & 1;
        return r;
    }
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
        W[t] = add_mod32(pb, {&W[t - 16], &s0, &W[t - 7], &s1}, 0);
    }
    Word r[8];
    for (int j = 0; j < 8; ++j) r[j] = word_const(SHA256_IV[j]);
    for (int t = 0; t < 64; ++t) {
        const Word &a = r[0], &b = r[1], &c = r[2], &d = r[3], &e = r[4], &f = r[5], &g = r[6], &h = r[7];
        Word S1 = word_xor3(pb, rotr(e, 6), rotr(e, 11), rotr(e, 25));
        Word ch = word_choice(pb, e, f, g);
        Word S0 = word_xor3(pb, rotr(a, 2), rotr(a, 13), rotr(a, 22));
        Word mj = word_majority(pb, a, b, c);
        Word new_e = add_mod32(pb, {&d, &h, &S1, &ch, &W[t]}, SHA256_K[t]);
        Word new_a = add_mod32(pb, {&h, &S1, &ch, &W[t], &S0, &mj}, SHA256_K[t]);
        for (int j = 7; j >= 1; --j) r[j] = r[j - 1];
        r[4] = new_e; r[0] = new_a;
    }
    for (int j = 0; j < 8; ++j) add_mod32(pb, {&r[j]}, SHA256_IV[j], &digest_out[32 * j]);
}

// packing: 1 * sum_i 2^i bits[i] = packed      (libsnark packing_gadget; little-endian over the given order)
inline void enforce_packing(Builder &pb, const std::vector<Var> &bits, size_t lo, size_t hi, Var packed, bool enforce_bitness) {
    if (!pb.recording) return;
    LC s; Fr w = Fr::one();
    for (size_t i = lo; i < hi; ++i) { if (enforce_bitness) pb.enforce_boolean(bits[i]); s.add(bits[i], w); w = w.dbl(); }
    pb.enforce(LC::constant(1), s, LC(packed));
}
inline void assign_packing(Builder &pb, const std::vector<Var> &bits, size_t lo, size_t hi, Var packed) {
    // at most 253 bits (FieldT::capacity()): the integer is assembled in 32-bit limbs and converted to Montgomery form once
    Fr s = Fr::zero();
    const uint8_t *tags = pb.tag_data();
    for (size_t i = lo; i < hi && i - lo < 256; ++i) if (tags[bits[i]]) s.v[(i - lo) >> 5] |= 1u << ((i - lo) & 31);
    pb.set(packed, s.to_mont());
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
    static const std::array<Fr, 65> inv_table = [] { std::array<Fr, 65> t; t[0] = Fr::zero(); for (uint64_t i = 1; i <= 64; ++i) t[i] = Fr::from_u64(i).inverse(); return t; }();
    pb.set(c.inv, cnt <= 64 ? inv_table[cnt] : Fr::from_u64(cnt).inverse());          // 1 / (number of set bits): one of 64 values for n <= 64
    bool leq = (alpha >> n) & 1;
    pb.set_bit(less_or_eq, leq);
    pb.set_bit(less, leq && cnt);
}

}}  // namespace zk::circuit
