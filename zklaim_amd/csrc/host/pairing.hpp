// pairing.hpp — reduced optimal-ate pairing on alt_bn128, host code (the verifier is a CPU step in the reference too).
//
// Replaces libff's alt_bn128 pairing as used by r1cs_gg_ppzksnark_verifier_strong_IC (/root/reference/zklaim/snark.cpp:62,
// reached from libsnark_verify, zklaim/libsnark_wrapper.cpp:252-276).  Written from the definition rather than from libff:
//   Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u;  Fq12 = Fq6[w]/(w^2 - v);  untwist (x, y) -> (x w^2, y w^3);
//   e(P, Q) = ( f_{6z+2,Q}(P) * l_{[6z+2]Q, pi(Q)}(P) * l_{[6z+2]Q + pi(Q), -pi^2(Q)}(P) ) ^ ((q^12 - 1)/r),  z = 4965661367192848881,
// The running points of the Miller loop are projective and the lines carry their slope's denominator (no inversion per step; the
// affine loop of the definition stays as multi_miller_loop_affine, the cross-check); the final exponentiation is (q^6 - 1)(q^2 + 1),
// then the Fuentes-Castaneda chain with cyclotomic squarings.  A three-pairing verification is ~1.0 ms of Miller loop and ~0.4 ms of
// final exponentiation on one host core.  Any bilinear, non-degenerate pairing makes the Groth16 check sound and complete; the
// final exponentiation follows libff's exponent (see below) so that alpha_g1_beta_g2 in a verification key means the same GT value.
#pragma once
#include <vector>
#include "../curve.hip.hpp"

namespace zk { namespace pairing {

// ---- host-only field inversion by the binary extended Euclidean algorithm on 4 x 64-bit limbs.  Fp::inverse() is Fermat's x^(p-2)
//      (380 products, 12 us on a host core), fine for a handful of conversions but the Miller loop below inverts once per step: 89
//      inversions were a fifth of a verification.  Montgomery in, Montgomery out: the integer inverse of aR is a^-1 R^-1; one Montgomery
//      product with R^3 turns it into a^-1 R.
namespace hostinv {
struct U256 { uint64_t w[4]; };
inline bool is_one(const U256 &a) { return a.w[0] == 1 && !(a.w[1] | a.w[2] | a.w[3]); }
inline bool geq(const U256 &a, const U256 &b) { for (int i = 3; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; return true; }
inline void sub(U256 &a, const U256 &b) { unsigned __int128 br = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - (uint64_t)br; a.w[i] = (uint64_t)d; br = (d >> 64) & 1; } }
inline void add(U256 &a, const U256 &b) { unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.w[i] + b.w[i]; a.w[i] = (uint64_t)c; c >>= 64; } }
inline void shr1(U256 &a) { for (int i = 0; i < 3; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 63); a.w[3] >>= 1; }
inline void halve_mod(U256 &x, const U256 &p) { if (x.w[0] & 1) add(x, p); shr1(x); }              // x / 2 mod p  (x < p < 2^254: no overflow)
inline void sub_mod(U256 &x, const U256 &y, const U256 &p) { if (geq(x, y)) sub(x, y); else { add(x, p); sub(x, y); } }
}  // namespace hostinv
inline Fq fq_inverse_host(const Fq &x) {
    using namespace hostinv;
    U256 u, v, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}}, p;
    for (int i = 0; i < 4; ++i) { u.w[i] = x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32); p.w[i] = FqParams::P[2 * i] | ((uint64_t)FqParams::P[2 * i + 1] << 32); }
    if (!(u.w[0] | u.w[1] | u.w[2] | u.w[3])) return Fq::zero();
    v = p;
    while (!is_one(u) && !is_one(v)) {
        while (!(u.w[0] & 1)) { shr1(u); halve_mod(x1, p); }
        while (!(v.w[0] & 1)) { shr1(v); halve_mod(x2, p); }
        if (geq(u, v)) { sub(u, v); sub_mod(x1, x2, p); } else { sub(v, u); sub_mod(x2, x1, p); }
    }
    const U256 &r = is_one(u) ? x1 : x2;                                        // (aR)^-1 as an integer
    Fq out;
    for (int i = 0; i < 4; ++i) { out.v[2 * i] = (uint32_t)r.w[i]; out.v[2 * i + 1] = (uint32_t)(r.w[i] >> 32); }
    static const Fq R3 = Fq::r2() * Fq::r2();                                   // the limbs of R^3 mod q
    return out * R3;
}
inline Fq2 fq2_inverse_host(const Fq2 &a) { Fq d = fq_inverse_host(a.c0.sqr() + a.c1.sqr()); return {a.c0 * d, (a.c1 * d).neg()}; }

inline Fq2 fq2(uint64_t a, uint64_t b) { return {Fq::from_u64(a), Fq::from_u64(b)}; }
inline Fq2 xi() { return fq2(9, 1); }
inline Fq2 mul_xi(const Fq2 &a) {                            // (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u: additions only
    Fq2 a8 = a.dbl().dbl().dbl();
    return {a8.c0 + a.c0 - a.c1, a8.c1 + a.c1 + a.c0};
}
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
        Fq2 d = fq2_inverse_host(c0 * t0 + mul_xi(c2 * t1) + mul_xi(c1 * t2));
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
    // complex squaring: (c0 + c1 w)^2 = (c0 + c1)(c0 + v c1) - ab - v ab + 2ab w, ab = c0 c1: two Fq6 products instead of three
    Fq12 sqr() const {
        Fq6 ab = c0 * c1;
        return {(c0 + c1) * (c0 + c1.mul_by_v()) - ab - ab.mul_by_v(), ab + ab};
    }
    // Granger-Scott squaring for elements of the cyclotomic subgroup (everything after the easy part of the final exponentiation):
    // three Fq4 squarings, i.e. 9 Fq2 products instead of the 12 of sqr().  Written from the published formulas; the layout is the
    // tower's (c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5) coefficients).  zkg_pairing_selfcheck compares a chain built on it with
    // plain square-and-multiply.
    Fq12 cyclotomic_sqr() const {
        Fq2 z0 = c0.c0, z4 = c0.c1, z3 = c0.c2, z2 = c1.c0, z1 = c1.c1, z5 = c1.c2;
        auto fq4_sqr = [](const Fq2 &a, const Fq2 &b, Fq2 &r0, Fq2 &r1) {       // (a + b y)^2 with y^2 = xi
            Fq2 ab = a * b;
            r0 = (a + b) * (a + mul_xi(b)) - ab - mul_xi(ab); r1 = ab + ab;
        };
        Fq2 t0, t1, t2, t3, t4, t5;
        fq4_sqr(z0, z1, t0, t1); fq4_sqr(z2, z3, t2, t3); fq4_sqr(z4, z5, t4, t5);
        z0 = t0 - z0; z0 = z0 + z0 + t0;
        z1 = t1 + z1; z1 = z1 + z1 + t1;
        Fq2 x5 = mul_xi(t5);
        z2 = x5 + z2; z2 = z2 + z2 + x5;
        z3 = t4 - z3; z3 = z3 + z3 + t4;
        z4 = t2 - z4; z4 = z4 + z4 + t2;
        z5 = t3 + z5; z5 = z5 + z5 + t3;
        return {{z0, z4, z3}, {z2, z1, z5}};
    }
    Fq12 cyclotomic_pow(const uint32_t *e, int nlimbs) const {                  // *this in the cyclotomic subgroup
        Fq12 r = one(); bool started = false;
        for (int i = nlimbs * 32 - 1; i >= 0; --i) {
            if (started) r = r.cyclotomic_sqr();
            if ((e[i >> 5] >> (i & 31)) & 1u) { r = started ? r * (*this) : *this; started = true; }
        }
        return r;
    }
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

// the line through T (slope lambda on the twist) evaluated at P is the sparse Fq12  yP + (-lambda xP) w + (lambda xT - yT) w^3
inline G2Affine frobenius_twist(const G2Affine &Q, const Fq2 &gx, const Fq2 &gy, bool conjugate_coords) {
    return conjugate_coords ? G2Affine{conj(Q.x) * gx, conj(Q.y) * gy} : G2Affine{Q.x * gx, Q.y * gy};
}

// gamma_1 = xi^((q-1)/6) (the constant the Miller loop's pi(Q) uses as well)
inline Fq2 gamma1() {
    static const uint32_t E_QM1_6[8] = {0x2414d4e1u, 0x34b01759u, 0xe6bda1c2u, 0xee9591c2u, 0xc0403964u, 0xf40d60f3u, 0xd032f006u, 0x0810b7bdu};
    Fq2 g1 = Fq2::one(), base = xi();
    for (int i = 8 * 32 - 1; i >= 0; --i) { g1 = g1.sqr(); if ((E_QM1_6[i >> 5] >> (i & 31)) & 1u) g1 = g1 * base; }
    return g1;
}
// f * l for a line l = a + b w + c w^3 (a = yP in Fq, b, c in Fq2): in the tower l = (a, 0, 0) + (b, c, 0) w.
inline Fq12 mul_by_line(const Fq12 &f, const Fq &a, const Fq2 &b, const Fq2 &c) {
    // f.c0 * (a,0,0) and f.c1 * (a,0,0) are coefficient scalings; X * (b, c, 0) is a sparse Fq6 product (v^3 = xi)
    auto sparse = [&](const Fq6 &x) {                        // x * (b + c v)
        Fq2 x0b = x.c0 * b, x1c = x.c1 * c;
        Fq2 mid = (x.c0 + x.c1) * (b + c) - x0b - x1c;       // x0 c + x1 b
        return Fq6{x0b + mul_xi(x.c2 * c), mid, x1c + x.c2 * b};
    };
    auto scl = [&](const Fq6 &x) { return Fq6{scale(x.c0, a), scale(x.c1, a), scale(x.c2, a)}; };
    // (f0 + f1 w)(A + B w) = f0 A + f1 B v + (f0 B + f1 A) w   with A = (a,0,0), B = (b,c,0), w^2 = v
    return {scl(f.c0) + sparse(f.c1).mul_by_v(), sparse(f.c0) + scl(f.c1)};
}

// The same product with affine running points, as the definition at the top of this file states it: one squaring of f per step for all
// pairs and ONE field inversion per step (the slopes' denominators are inverted together).  Kept as the cross-check of the
// inversion-free loop below (zkg_pairing_selfcheck): a step's inversion costs 9 us on a host core, 89 of them were half of the loop.
inline Fq12 multi_miller_loop_affine(const G1Affine *P, const G2Affine *Q, int n) {
    const unsigned __int128 S = ((unsigned __int128)1 << 64) + 11347224129447541672ull;   // 6z+2 = 29793968203157093288 (65 bits)
    Fq12 f = Fq12::one();
    std::vector<G2Affine> T(Q, Q + n);
    std::vector<Fq2> den(n), pre(n);
    auto invert_all = [&]() {                                // den[j] <- 1/den[j]
        Fq2 run = Fq2::one();
        for (int j = 0; j < n; ++j) { pre[j] = run; run = run * den[j]; }
        Fq2 inv = fq2_inverse_host(run);
        for (int j = n - 1; j >= 0; --j) { Fq2 d = inv * pre[j]; inv = inv * den[j]; den[j] = d; }
    };
    auto apply = [&](int j, const Fq2 &lambda, const Fq2 &x_other) {       // multiply the line in, move T_j along it
        f = mul_by_line(f, P[j].y, scale(lambda, P[j].x).neg(), lambda * T[j].x - T[j].y);
        Fq2 x3 = lambda.sqr() - T[j].x - x_other;
        T[j] = {x3, lambda * (T[j].x - x3) - T[j].y};
    };
    auto dbl_step = [&]() {
        for (int j = 0; j < n; ++j) den[j] = T[j].y.dbl();
        invert_all();
        f = f.sqr();
        for (int j = 0; j < n; ++j) { Fq2 xx = T[j].x.sqr(); apply(j, (xx.dbl() + xx) * den[j], T[j].x); }
    };
    auto add_step = [&](const std::vector<G2Affine> &R_) {
        for (int j = 0; j < n; ++j) den[j] = R_[j].x - T[j].x;
        invert_all();
        for (int j = 0; j < n; ++j) apply(j, (R_[j].y - T[j].y) * den[j], R_[j].x);
    };
    std::vector<G2Affine> Q0(Q, Q + n);
    for (int i = 63; i >= 0; --i) {                          // bit 64 is the leading one
        dbl_step();
        if ((S >> i) & 1) add_step(Q0);
    }
    // gamma = xi^((q-1)/6): pi(Q) = (conj(x) gamma^2, conj(y) gamma^3); pi^2(Q) = (x N^2, y N^3) with N = gamma conj(gamma) in Fq
    static const Fq2 g1 = gamma1();
    static const Fq2 g2 = g1.sqr(), g3 = g2 * g1, n1 = g1 * conj(g1), n2 = n1.sqr(), n3 = n2 * n1;
    std::vector<G2Affine> Q1(n), Q2(n);
    for (int j = 0; j < n; ++j) {
        Q1[j] = frobenius_twist(Q[j], g2, g3, true);
        G2Affine t = frobenius_twist(Q[j], n2, n3, false);
        Q2[j] = {t.x, t.y.neg()};
    }
    add_step(Q1);
    add_step(Q2);
    return f;
}
// f * l for a line l = a + b w + c w^3 with a, b, c in Fq2 (the inversion-free steps below scale the line by an Fq2 factor, which the
// final exponentiation removes: (q^12 - 1)/r is a multiple of q^6 - 1).  l = A + B w with A = (a,0,0), B = (b,c,0); Karatsuba over w:
// 3 + 5 + 5 = 13 Fq2 products.
inline Fq12 mul_by_line2(const Fq12 &f, const Fq2 &a, const Fq2 &b, const Fq2 &c) {
    auto sparse = [](const Fq6 &x, const Fq2 &b_, const Fq2 &c_) {             // x * (b_ + c_ v), v^3 = xi
        Fq2 x0b = x.c0 * b_, x1c = x.c1 * c_;
        Fq2 mid = (x.c0 + x.c1) * (b_ + c_) - x0b - x1c;
        return Fq6{x0b + mul_xi(x.c2 * c_), mid, x1c + x.c2 * b_};
    };
    Fq6 f0A = {f.c0.c0 * a, f.c0.c1 * a, f.c0.c2 * a}, f1B = sparse(f.c1, b, c);
    return {f0A + f1B.mul_by_v(), sparse(f.c0 + f.c1, a + b, c) - f0A - f1B};
}

// Product of the Miller functions of the optimal ate pairing over several (P_j, Q_j) (no final exponentiation); all points
// finite.  The loops run in lock-step (one squaring of f per step for all pairs), the running points T_j = (X : Y : Z) are
// homogeneous projective on the twist Y^2 Z = X^3 + b' Z^3, b' = 3/xi, and every line is the affine line of the definition above
// multiplied through by its slope's denominator, so no step inverts anything:
//   doubling (slope 3X^2 / 2YZ):   l * 2YZ  =  2YZ yP  -  3X^2 xP w  +  (Y^2 - 3b' Z^2) w^3      [3X^3 - 2Y^2 Z = Z (Y^2 - 3b' Z^2) on the curve]
//   adding Q (slope theta / mu, theta = Y - yQ Z, mu = X - xQ Z):   l * mu  =  mu yP  -  theta xP w  +  (theta xQ - mu yQ) w^3
// and the point updates are the same chord / tangent rules with denominators cleared.
// One step's line, before the G1 point enters it: the step multiplies f by  a yP  -  b xP w  +  c w^3.
struct LineCoeff { Fq2 a, b, c; };
// The running point of one pairing's Miller loop and the two kinds of step.
struct LineWalker {
    Fq2 X, Y, Z;
    explicit LineWalker(const G2Affine &Q) : X(Q.x), Y(Q.y), Z(Fq2::one()) {}
    LineCoeff dbl() {
        static const Fq2 b3 = fq2(9, 0) * fq2_inverse_host(xi());       // 3 b'
        Fq2 B = Y.sqr(), C = Z.sqr(), E = C * b3, F = E.dbl() + E, H = (Y + Z).sqr() - B - C, J = X.sqr();
        LineCoeff l = {H, J.dbl() + J, B - E};
        Fq2 E2 = E.sqr(), E4 = E2.dbl().dbl();
        Fq2 X3 = ((X * Y) * (B - F)).dbl(), Y3 = (B + F).sqr() - (E4.dbl() + E4), Z3 = (B * H).dbl().dbl();
        X = X3; Y = Y3; Z = Z3;
        return l;
    }
    LineCoeff add(const G2Affine &R) {
        Fq2 theta = Y - R.y * Z, mu = X - R.x * Z;
        LineCoeff l = {mu, theta, theta * R.x - mu * R.y};
        Fq2 C = theta.sqr(), D = mu.sqr(), E = mu * D, F = Z * C, G = X * D, H = E + F - G.dbl();
        Fq2 X3 = mu * H, Y3 = theta * (G - H) - E * Y, Z3 = Z * E;
        X = X3; Y = Y3; Z = Z3;
        return l;
    }
};
// The step schedule of the optimal ate loop for 6z + 2 = 29793968203157093288 (65 bits; bit 64 is the leading one): f(step_is_doubling,
// which_addend) is called for 64 doublings, an addition of Q after each doubling whose bit is set, then the additions of pi(Q) and -pi^2(Q).
template <class Fn> inline void miller_schedule(Fn step) {
    const unsigned __int128 S = ((unsigned __int128)1 << 64) + 11347224129447541672ull;
    for (int i = 63; i >= 0; --i) { step(true, 0); if ((S >> i) & 1) step(false, 0); }
    step(false, 1); step(false, 2);
}
// Q, pi(Q), -pi^2(Q):  gamma = xi^((q-1)/6): pi(Q) = (conj(x) gamma^2, conj(y) gamma^3); pi^2(Q) = (x N^2, y N^3) with N = gamma conj(gamma) in Fq
inline void miller_addends(const G2Affine &Q, G2Affine out[3]) {
    static const Fq2 g1 = gamma1();
    static const Fq2 g2 = g1.sqr(), g3 = g2 * g1, n1 = g1 * conj(g1), n2 = n1.sqr(), n3 = n2 * n1;
    out[0] = Q;
    out[1] = frobenius_twist(Q, g2, g3, true);
    G2Affine t = frobenius_twist(Q, n2, n3, false);
    out[2] = {t.x, t.y.neg()};
}
// every line of the loop of a fixed Q, in schedule order: what a verification key's gamma_g2 and delta_g2 contribute to every verification
inline std::vector<LineCoeff> miller_lines(const G2Affine &Q) {
    std::vector<LineCoeff> lines; lines.reserve(96);
    G2Affine addend[3]; miller_addends(Q, addend);
    LineWalker t(Q);
    miller_schedule([&](bool doubling, int which) { lines.push_back(doubling ? t.dbl() : t.add(addend[which])); });
    return lines;
}
// prepared[j] (optional, per pair): the lines of Q[j] computed earlier by miller_lines — Q[j] itself is not read then
inline Fq12 multi_miller_loop(const G1Affine *P, const G2Affine *Q, int n, const std::vector<LineCoeff> *const *prepared = nullptr) {
    Fq12 f = Fq12::one();
    std::vector<LineWalker> T; std::vector<G2Affine> addend((size_t)3 * n);
    T.reserve(n);
    for (int j = 0; j < n; ++j) {
        const bool live = !(prepared && prepared[j]);
        T.emplace_back(live ? Q[j] : G2Affine{Fq2::one(), Fq2::one()});
        if (live) miller_addends(Q[j], &addend[(size_t)3 * j]);
    }
    size_t pos = 0;
    miller_schedule([&](bool doubling, int which) {
        if (doubling) f = f.sqr();
        for (int j = 0; j < n; ++j) {
            const LineCoeff l = (prepared && prepared[j]) ? (*prepared[j])[pos] : (doubling ? T[j].dbl() : T[j].add(addend[(size_t)3 * j + which]));
            f = mul_by_line2(f, scale(l.a, P[j].y), scale(l.b, P[j].x).neg(), l.c);
        }
        ++pos;
    });
    return f;
}

inline Fq12 miller_loop(const G1Affine &P, const G2Affine &Q) { return multi_miller_loop(&P, &Q, 1); }

// x -> x^(q^k), k = 1, 2, 3.  Fq12 = Fq2[w]/(w^6 - xi): the coefficient a_i of w^i goes to conj^k(a_i) * gamma_k^i with
// gamma_k = xi^((q^k - 1)/6): gamma_2 = gamma_1 conj(gamma_1) (in Fq), gamma_3 = gamma_2 gamma_1.
inline Fq12 frobenius(const Fq12 &x, int k) {
    static const Fq2 G1c = gamma1();
    static const Fq2 G2c = G1c * conj(G1c), G3c = G2c * G1c;
    const Fq2 g = k == 1 ? G1c : k == 2 ? G2c : G3c;
    Fq2 pw[6]; pw[0] = Fq2::one(); for (int i = 1; i < 6; ++i) pw[i] = pw[i - 1] * g;
    auto m = [&](const Fq2 &a, int i) { return ((k & 1) ? conj(a) : a) * pw[i]; };
    // tower layout: c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5)
    return {{m(x.c0.c0, 0), m(x.c0.c1, 2), m(x.c0.c2, 4)}, {m(x.c1.c0, 1), m(x.c1.c1, 3), m(x.c1.c2, 5)}};
}
// elt^z for the curve parameter z = 4965661367192848881 (libff alt_bn128_final_exponent_z; q and r are polynomials in z)
inline Fq12 exp_by_z(const Fq12 &x) { static const uint32_t Z[2] = {0x4a6909f1u, 0x44e992b4u}; return x.cyclotomic_pow(Z, 2); }   // x in the cyclotomic subgroup
inline Fq12 exp_by_neg_z(const Fq12 &x) { return exp_by_z(x).conjugate(); }     // unitary inverse: x is in the cyclotomic subgroup here

// libff alt_bn128_final_exponentiation [UPSTREAM-RECALL]: first chunk f^((q^6 - 1)(q^2 + 1)), then the last chunk by the
// Fuentes-Castaneda et al. chain, which raises to  lambda_0 + lambda_1 q + lambda_2 q^2 + lambda_3 q^3,
//   lambda_0 = 12z^3 + 12z^2 + 6z + 1, lambda_1 = 12z^3 + 6z^2 + 4z, lambda_2 = 12z^3 + 6z^2 + 6z, lambda_3 = 12z^3 + 6z^2 + 4z - 1,
// = 2z(6z^2 + 3z + 1) * (q^4 - q^2 + 1)/r: a fixed multiple of the exact hard exponent, so GT values agree with libff's
// (the alpha_g1_beta_g2 element of a verification key) and not with a textbook reduced pairing.
// tests/test_verifier.py checks the chain against a plain square-and-multiply by that integer.
inline Fq12 final_exponentiation_first_chunk(const Fq12 &f) {
    Fq12 a = f.conjugate() * f.inverse();                    // f^(q^6 - 1)
    return frobenius(a, 2) * a;                              // ^(q^2 + 1)
}
inline Fq12 final_exponentiation_last_chunk(const Fq12 &elt) {
    Fq12 A = exp_by_neg_z(elt), B = A.cyclotomic_sqr(), C = B.cyclotomic_sqr(), D = C * B;
    Fq12 E = exp_by_neg_z(D), F = E.cyclotomic_sqr(), G = exp_by_neg_z(F);
    Fq12 H = D.conjugate(), I = G.conjugate();
    Fq12 J = I * E, K = J * H, L = K * B, M = K * E, N = M * elt;
    Fq12 O = frobenius(L, 1), P = O * N, Q = frobenius(K, 2), R = Q * P;
    Fq12 S = elt.conjugate(), T = S * L, U = frobenius(T, 3);
    return U * R;
}
inline Fq12 final_exponentiation(const Fq12 &f) { return final_exponentiation_last_chunk(final_exponentiation_first_chunk(f)); }

inline Fq12 reduced_pairing(const G1Affine &P, const G2Affine &Q) {
    if (P.is_inf() || Q.is_inf()) return Fq12::one();
    return final_exponentiation(miller_loop(P, Q));
}

inline bool on_curve_g1(const G1Affine &P) { return P.is_inf() || P.y.sqr() == P.x.sqr() * P.x + Fq::from_u64(3); }
inline bool on_curve_g2(const G2Affine &Q) { return Q.is_inf() || Q.y.sqr() == Q.x.sqr() * Q.x + fq2(3, 0) * xi().inverse(); }

}}  // namespace zk::pairing
