// serialize.hpp — libff's binary point / field (de)serialisation on the host (BINARY_OUTPUT, MONTGOMERY_OUTPUT, point
// compression: the flags libsnark's build sets; reached from operator<< / operator>> at
// /root/reference/zklaim/libsnark_wrapper.cpp:126,142,150,165,173,189).  G1 = '0'|'1' X '0'|'1' (34 B), G2 66 B.
#pragma once
#include <cstring>
#include <memory>
#include <utility>
#include <vector>
#include "../curve.hip.hpp"
#include "pairing.hpp"

namespace zk { namespace ser {

inline bool canonical_lsb(const Fq &y) { return y.from_mont().v[0] & 1u; }
inline size_t put_g1(uint8_t *out, const G1Affine &a) {
    bool inf = a.is_inf();
    Fq x = inf ? Fq::zero() : a.x, y = inf ? Fq::one() : a.y;
    out[0] = inf ? '1' : '0'; memcpy(out + 1, x.v, 32); out[33] = canonical_lsb(y) ? '1' : '0';
    return 34;
}
inline size_t put_g2(uint8_t *out, const G2Affine &a) {
    bool inf = a.is_inf();
    Fq2 x = inf ? Fq2::zero() : a.x, y = inf ? Fq2::one() : a.y;
    out[0] = inf ? '1' : '0'; memcpy(out + 1, x.c0.v, 32); memcpy(out + 33, x.c1.v, 32); out[65] = canonical_lsb(y.c0) ? '1' : '0';
    return 66;
}
inline size_t put_g1(uint8_t *out, const G1 &p) { return put_g1(out, p.to_affine()); }
inline size_t put_g2(uint8_t *out, const G2 &p) { return put_g2(out, p.to_affine()); }

inline Fq fq_sqrt_candidate(const Fq &a) {                   // a^((q+1)/4), q = 3 mod 4
    static const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    return a.pow(e, 8);
}
inline bool fq2_sqrt(const Fq2 &a, Fq2 &out) {               // complex method
    if (a.c1.is_zero()) {
        Fq s = fq_sqrt_candidate(a.c0);
        if (s.sqr() == a.c0) { out = {s, Fq::zero()}; return true; }
        Fq t = fq_sqrt_candidate(a.c0.neg());
        if (t.sqr() == a.c0.neg()) { out = {Fq::zero(), t}; return true; }
        return false;
    }
    Fq norm = a.c0.sqr() + a.c1.sqr(), s = fq_sqrt_candidate(norm);
    if (s.sqr() != norm) return false;
    Fq two_inv = Fq::from_u64(2).inverse();
    Fq d = (a.c0 + s) * two_inv, c0 = fq_sqrt_candidate(d);
    if (c0.sqr() != d) { d = (a.c0 - s) * two_inv; c0 = fq_sqrt_candidate(d); if (c0.sqr() != d) return false; }
    out = {c0, a.c1 * c0.dbl().inverse()};
    return true;
}
inline bool get_g1(const uint8_t *p, G1Affine &out) {
    if (p[0] == '1') { out = G1Affine::inf(); return true; }
    if (p[0] != '0' || (p[33] != '0' && p[33] != '1')) return false;
    Fq x; memcpy(x.v, p + 1, 32);
    Fq rhs = x.sqr() * x + Fq::from_u64(3), y = fq_sqrt_candidate(rhs);
    if (y.sqr() != rhs) return false;
    if (canonical_lsb(y) != (p[33] == '1')) y = y.neg();
    out = {x, y};
    return true;
}
inline bool get_g2(const uint8_t *p, G2Affine &out) {
    if (p[0] == '1') { out = G2Affine::inf(); return true; }
    if (p[0] != '0' || (p[65] != '0' && p[65] != '1')) return false;
    Fq2 x; memcpy(x.c0.v, p + 1, 32); memcpy(x.c1.v, p + 33, 32);
    Fq2 rhs = x.sqr() * x + pairing::fq2(3, 0) * pairing::xi().inverse(), y;
    if (!fq2_sqrt(rhs, y)) return false;
    if (canonical_lsb(y.c0) != (p[65] == '1')) y = y.neg();
    out = {x, y};
    return true;
}
// Fq12: c0.c0.c0, c0.c0.c1, c0.c1.c0, ... (libff Fp12_2over3over2 operator<<), 12 x 32 B
inline size_t put_fq12(uint8_t *out, const pairing::Fq12 &g) {
    const Fq2 *c[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
    for (int i = 0; i < 6; ++i) { memcpy(out + 64 * i, c[i]->c0.v, 32); memcpy(out + 64 * i + 32, c[i]->c1.v, 32); }
    return 384;
}
inline void get_fq12(const uint8_t *p, pairing::Fq12 &g) {
    Fq2 *c[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
    for (int i = 0; i < 6; ++i) { memcpy(c[i]->c0.v, p + 64 * i, 32); memcpy(c[i]->c1.v, p + 64 * i + 32, 32); }
}

// byte buffer whose resize() leaves new bytes uninitialised: the key writer sizes runs of hundreds of MB and fills them on the host pool
template <class T> struct UninitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef UninitAlloc<U> other; };
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new ((void *)p) U(std::forward<Args>(args)...); }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }             // default-init: no zero fill for bytes
};
typedef std::vector<uint8_t, UninitAlloc<uint8_t>> Bytes;
struct Writer {
    Bytes buf;
    void dec(size_t v) {                                     // ASCII decimal + '\n' (millions of these in a pk: no snprintf)
        char t[24]; int n = 0;
        do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) buf.push_back((uint8_t)t[--n]);
        buf.push_back('\n');
    }
    void raw(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; buf.insert(buf.end(), b, b + n); }
    void g1(const G1Affine &a) { uint8_t t[34]; put_g1(t, a); raw(t, 34); }
    void g2(const G2Affine &a) { uint8_t t[66]; put_g2(t, a); raw(t, 66); }
};
struct Reader {
    const uint8_t *p, *end; bool ok = true;
    size_t dec() {
        size_t v = 0; int nd = 0;
        while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); ++p; if (++nd > 19) { ok = false; return 0; } }
        if (nd == 0 || p >= end || *p != '\n') { ok = false; return 0; }
        ++p; return v;
    }
    const uint8_t *take(size_t n) { if (!ok || (size_t)(end - p) < n) { ok = false; return nullptr; } const uint8_t *r = p; p += n; return r; }
};

}}  // namespace zk::ser
