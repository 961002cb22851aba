// pairing.hpp — reduced optimal-ate pairing on alt_bn128, host code (the verifier is a CPU step in the reference too).
//
// Replaces libff's alt_bn128 pairing as used by r1cs_gg_ppzksnark_verifier_strong_IC (/root/reference/zklaim/snark.cpp:62,
// reached from libsnark_verify, zklaim/libsnark_wrapper.cpp:252-276).  Written from the definition rather than from libff:
//   Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u;  Fq12 = Fq6[w]/(w^2 - v);  untwist (x, y) -> (x w^2, y w^3);
//   e(P, Q) = ( f_{6z+2,Q}(P) * l_{[6z+2]Q, pi(Q)}(P) * l_{[6z+2]Q + pi(Q), -pi^2(Q)}(P) ) ^ ((q^12 - 1)/r),  z = 4965661367192848881,
// with affine line functions (one Fq2 inversion per step) and the final exponentiation as (q^6 - 1), then generic
// square-and-multiply by (q^2 + 1) and by (q^4 - q^2 + 1)/r.  ~8 ms per pairing on one host core: a verification is three
// Miller loops and one final exponentiation, milliseconds next to the signature check and key parsing around it.  Any
// bilinear, non-degenerate pairing makes the Groth16 check sound and complete; GT values are NOT claimed to equal libff's
// representation bit for bit (libff may differ by a fixed unit power), which only matters for exchanging vk blobs.
#pragma once
#include "../curve.cuh"

namespace zk { namespace pairing {

inline Fq2 fq2(uint64_t a, uint64_t b) { return {Fq::from_u64(a), Fq::from_u64(b)}; }
inline Fq2 xi() { return fq2(9, 1); }
inline Fq2 mul_xi(const Fq2 &a) { return a * xi(); }
inline Fq2 conj(const Fq2 &a) { return {a.c0, a.c1.neg()}; }
inline Fq2 scale(const Fq2 &a, const Fq &s) { return {a.c0 * s, a.c1 * s}; }

struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    bool operator==(const Fq6 &o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    Fq6 operator+(const Fq6 &o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator-(const Fq6 &o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    Fq6 operator*(const Fq6 &o) const {                      // schoolbook with v^3 = xi
        Fq2 a0 = c0 * o.c0, a1 = c1 * o.c1, a2 = c2 * o.c2;
        Fq2 t0 = a0 + mul_xi((c1 + c2) * (o.c1 + o.c2) - a1 - a2);
        Fq2 t1 = (c0 + c1) * (o.c0 + o.c1) - a0 - a1 + mul_xi(a2);
        Fq2 t2 = (c0 + c2) * (o.c0 + o.c2) - a0 - a2 + a1;
        return {t0, t1, t2};
    }
    Fq6 mul_by_v() const { return {mul_xi(c2), c0, c1}; }
    Fq6 inverse() const {
        Fq2 t0 = c0.sqr() - mul_xi(c1 * c2), t1 = mul_xi(c2.sqr()) - c0 * c1, t2 = c1.sqr() - c0 * c2;
        Fq2 d = (c0 * t0 + mul_xi(c2 * t1) + mul_xi(c1 * t2)).inverse();
        return {t0 * d, t1 * d, t2 * d};
    }
};

struct Fq12 {
    Fq6 c0, c1;                                              // c0 + c1 w
    static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    bool operator==(const Fq12 &o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fq12 &o) const { return !(*this == o); }
    Fq12 operator*(const Fq12 &o) const {
        Fq6 a = c0 * o.c0, b = c1 * o.c1;
        return {a + b.mul_by_v(), (c0 + c1) * (o.c0 + o.c1) - a - b};
    }
    Fq12 sqr() const { return (*this) * (*this); }
    Fq12 conjugate() const { return {c0, c1.neg()}; }         // = x^(q^6)
    Fq12 inverse() const { Fq6 d = (c0 * c0 - (c1 * c1).mul_by_v()).inverse(); return {c0 * d, (c1 * d).neg()}; }
    Fq12 pow(const uint32_t *e, int nlimbs) const {
        Fq12 r = one(); bool started = false;
        for (int i = nlimbs * 32 - 1; i >= 0; --i) {
            if (started) r = r.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1u) { r = started ? r * (*this) : *this; started = true; }
        }
        return r;
    }
};

// the line through T (slope lambda on the twist) evaluated at P, as a sparse Fq12: yP + (-lambda xP) w + (lambda xT - yT) w^3
inline Fq12 line(const Fq2 &lambda, const G2Affine &T, const G1Affine &P) {
    Fq12 l; l.c0 = {Fq2{P.y, Fq::zero()}, Fq2::zero(), Fq2::zero()};
    l.c1 = {scale(lambda, P.x).neg(), lambda * T.x - T.y, Fq2::zero()};
    return l;
}
inline G2Affine frobenius_twist(const G2Affine &Q, const Fq2 &gx, const Fq2 &gy, bool conjugate_coords) {
    return conjugate_coords ? G2Affine{conj(Q.x) * gx, conj(Q.y) * gy} : G2Affine{Q.x * gx, Q.y * gy};
}

// Miller function of the optimal ate pairing (no final exponentiation); P, Q finite
inline Fq12 miller_loop(const G1Affine &P, const G2Affine &Q) {
    static const uint32_t E_QM1_6[8] = {0x2414d4e1u, 0x34b01759u, 0xe6bda1c2u, 0xee9591c2u, 0xc0403964u, 0xf40d60f3u, 0xd032f006u, 0x0810b7bdu};
    const unsigned __int128 S = ((unsigned __int128)1 << 64) + 11347224129447541672ull;   // 6z+2 = 29793968203157093288 (65 bits)
    Fq12 f = Fq12::one();
    G2Affine T = Q;
    auto dbl_step = [&]() {
        Fq2 xx = T.x.sqr();
        Fq2 lambda = (xx.dbl() + xx) * T.y.dbl().inverse();
        f = f.sqr() * line(lambda, T, P);
        Fq2 x3 = lambda.sqr() - T.x.dbl();
        T = {x3, lambda * (T.x - x3) - T.y};
    };
    auto add_step = [&](const G2Affine &R_) {
        Fq2 lambda = (R_.y - T.y) * (R_.x - T.x).inverse();
        f = f * line(lambda, T, P);
        Fq2 x3 = lambda.sqr() - T.x - R_.x;
        T = {x3, lambda * (T.x - x3) - T.y};
    };
    for (int i = 63; i >= 0; --i) {                          // bit 64 is the leading one
        dbl_step();
        if ((S >> i) & 1) add_step(Q);
    }
    // gamma = xi^((q-1)/6): pi(Q) = (conj(x) gamma^2, conj(y) gamma^3); pi^2(Q) = (x N^2, y N^3) with N = gamma conj(gamma) in Fq
    Fq2 g1 = Fq2::one(); { Fq2 base = xi(); for (int i = 8 * 32 - 1; i >= 0; --i) { g1 = g1.sqr(); if ((E_QM1_6[i >> 5] >> (i & 31)) & 1u) g1 = g1 * base; } }
    Fq2 g2 = g1.sqr(), g3 = g2 * g1;
    Fq2 n1 = g1 * conj(g1), n2 = n1.sqr(), n3 = n2 * n1;
    G2Affine Q1 = frobenius_twist(Q, g2, g3, true);
    G2Affine Q2 = frobenius_twist(Q, n2, n3, false);
    add_step(Q1);
    add_step(G2Affine{Q2.x, Q2.y.neg()});
    return f;
}

inline Fq12 final_exponentiation(const Fq12 &f) {
    static const uint32_t E_Q2P1[16] = {0x275d69b2u, 0x3b5458a2u, 0x09eac101u, 0xa602072du, 0x6d96cadcu, 0x4a50189cu, 0x7a1242c8u, 0x04689e95u,
                                        0x34c6b38du, 0x26edfa5cu, 0x16375606u, 0xb00b8551u, 0x0348d21cu, 0x599a6f7cu, 0x763cbf9cu, 0x0925c4b8u};
    static const uint32_t E_HARD[24] = {0xccdf42b1u, 0xe81bb482u, 0xf49c36d4u, 0x5abf5cc4u, 0x1da014fdu, 0xf1154e7eu, 0x87cdbacfu, 0xdcc7b44cu,
                                        0x954bcf8au, 0xaaa441e3u, 0xd5095f23u, 0x6b887d56u, 0xf3fd90c6u, 0x79581e16u, 0xd189227du, 0x3b1b1355u,
                                        0x61876f6bu, 0x4e529a58u, 0xd5b12278u, 0x6c0eb522u, 0x83177fafu, 0x331ec151u, 0x0b0759adu, 0x01baaa71u};
    Fq12 g = f.conjugate() * f.inverse();                    // f^(q^6 - 1)
    g = g.pow(E_Q2P1, 16);                                    // ^(q^2 + 1)
    return g.pow(E_HARD, 24);                                 // ^((q^4 - q^2 + 1)/r)
}

inline Fq12 reduced_pairing(const G1Affine &P, const G2Affine &Q) {
    if (P.is_inf() || Q.is_inf()) return Fq12::one();
    return final_exponentiation(miller_loop(P, Q));
}

inline bool on_curve_g1(const G1Affine &P) { return P.is_inf() || P.y.sqr() == P.x.sqr() * P.x + Fq::from_u64(3); }
inline bool on_curve_g2(const G2Affine &Q) { return Q.is_inf() || Q.y.sqr() == Q.x.sqr() * Q.x + fq2(3, 0) * xi().inverse(); }

}}  // namespace zk::pairing
