/*
 * zkoracle.cpp — CPU restatement of the Groth16 prove path zklaim reaches through
 *     r1cs_gg_ppzksnark_prover<ppT>(pk, primary, auxiliary)      /root/reference/zklaim/snark.cpp:126
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under zklaim_amd/ links, imports or
 * calls it, and the product fails loudly if its HIP library is missing.
 *
 * PARITY UNPINNED.  The arithmetic behind snark.cpp:126 lives in scipr-lab/libsnark and its
 * depends/libff, depends/libfqfft — an un-vendored submodule with no recoverable pinned commit
 * (/root/reference/.gitmodules:1-6; lib/libsnark is an empty directory), and the reference's
 * own tests assert return codes only (zklaim/tests/zklaim.cpp).  The functions below restate
 * the published algorithms of those libraries; each names the upstream function it follows and
 * the reference call site that reaches it.  The restatement is pinned instead by the
 * independent big-integer definitions in oracle/pyref.py (tests/golden/).
 *
 * Deliberately shares no code with zklaim_amd/csrc: this file uses 4x64-bit limbs with
 * unsigned __int128 products, the HIP path uses 8x32-bit limbs.
 */
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/zkg.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;

/* ------------------------------------------------------------------ field parameters */
struct FpParams {
    u64 p[4];      /* modulus                                  */
    u64 inv;       /* -p^{-1} mod 2^64                         */
    u64 one[4];    /* R mod p                                  */
    u64 r2[4];     /* R^2 mod p                                */
};

static inline bool ge4(const u64 *a, const u64 *b) {
    for (int i = 3; i >= 0; --i) { if (a[i] != b[i]) return a[i] > b[i]; }
    return true;
}
static inline u64 sub4(u64 *r, const u64 *a, const u64 *b) {
    u64 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - br; r[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return br;
}
static inline u64 add4(u64 *r, const u64 *a, const u64 *b) {
    u64 c = 0;
    for (int i = 0; i < 4; ++i) { u128 s = (u128)a[i] + b[i] + c; r[i] = (u64)s; c = (u64)(s >> 64); }
    return c;
}

static FpParams make_params(const u64 p[4]) {
    FpParams P; memcpy(P.p, p, 32);
    u64 x = 1;                                       /* Newton: x = p^{-1} mod 2^64 */
    for (int i = 0; i < 6; ++i) x *= 2 - p[0] * x;
    P.inv = (u64)0 - x;
    /* R mod p and R^2 mod p by 256 / 512 modular doublings of 1 */
    u64 v[4] = {1, 0, 0, 0};
    for (int i = 0; i < 512; ++i) {
        u64 t[4]; u64 c = add4(t, v, v);
        if (c || ge4(t, p)) sub4(t, t, p);
        memcpy(v, t, 32);
        if (i == 255) memcpy(P.one, v, 32);
    }
    memcpy(P.r2, v, 32);
    return P;
}

static const u64 Q_LIMBS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 R_LIMBS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const FpParams PQ = make_params(Q_LIMBS);
static const FpParams PR = make_params(R_LIMBS);

/* ------------------------------------------------------------------ Fp (libff Fp_model<4,modulus>) */
template <const FpParams &P>
struct Fp {
    u64 v[4];
    static Fp zero() { Fp r; memset(r.v, 0, 32); return r; }
    static Fp one() { Fp r; memcpy(r.v, P.one, 32); return r; }
    static Fp from_limbs(const u64 *l) { Fp r; memcpy(r.v, l, 32); return r; }        /* already Montgomery */
    static Fp from_canonical(const u64 *l) { Fp a; memcpy(a.v, l, 32); Fp r2; memcpy(r2.v, P.r2, 32); return a * r2; }
    static Fp from_u64(u64 x) { u64 l[4] = {x, 0, 0, 0}; return from_canonical(l); }
    void to_canonical(u64 *out) const { Fp o; o.v[0] = 1; o.v[1] = o.v[2] = o.v[3] = 0; Fp r = (*this) * o; memcpy(out, r.v, 32); }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
    bool operator==(const Fp &o) const { return memcmp(v, o.v, 32) == 0; }
    bool operator!=(const Fp &o) const { return !(*this == o); }
    Fp operator+(const Fp &o) const { Fp r; u64 c = add4(r.v, v, o.v); if (c || ge4(r.v, P.p)) sub4(r.v, r.v, P.p); return r; }
    Fp operator-(const Fp &o) const { Fp r; if (sub4(r.v, v, o.v)) add4(r.v, r.v, P.p); return r; }
    Fp operator-() const { if (is_zero()) return *this; Fp r; sub4(r.v, P.p, v); return r; }
    /* Montgomery product, CIOS (libff Fp_model::mul_reduce) */
    Fp operator*(const Fp &o) const {
        u64 t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u64 c = 0;
            for (int j = 0; j < 4; ++j) { u128 x = (u128)v[j] * o.v[i] + t[j] + c; t[j] = (u64)x; c = (u64)(x >> 64); }
            u128 x = (u128)t[4] + c; t[4] = (u64)x; t[5] = (u64)(x >> 64);
            u64 m = t[0] * P.inv;
            x = (u128)m * P.p[0] + t[0]; c = (u64)(x >> 64);
            for (int j = 1; j < 4; ++j) { x = (u128)m * P.p[j] + t[j] + c; t[j - 1] = (u64)x; c = (u64)(x >> 64); }
            x = (u128)t[4] + c; t[3] = (u64)x; t[4] = t[5] + (u64)(x >> 64);
        }
        Fp r; memcpy(r.v, t, 32);
        if (t[4] || ge4(r.v, P.p)) sub4(r.v, r.v, P.p);
        return r;
    }
    Fp sqr() const { return (*this) * (*this); }
    Fp &operator+=(const Fp &o) { *this = *this + o; return *this; }
    Fp &operator-=(const Fp &o) { *this = *this - o; return *this; }
    Fp &operator*=(const Fp &o) { *this = *this * o; return *this; }
    Fp dbl() const { return *this + *this; }
    Fp pow(const u64 *e, int limbs) const {
        Fp r = one(); bool started = false;
        for (int i = limbs * 64 - 1; i >= 0; --i) {
            if (started) r = r.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) { r = started ? r * (*this) : *this; started = true; }
        }
        return r;
    }
    Fp pow_u64(u64 e) const { return pow(&e, 1); }
    Fp inverse() const { u64 e[4]; u64 two[4] = {2, 0, 0, 0}; sub4(e, P.p, two); return pow(e, 4); }   /* Fermat */
    bool canonical_lsb() const { u64 c[4]; to_canonical(c); return c[0] & 1; }
};
typedef Fp<PQ> Fq;
typedef Fp<PR> Fr;

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2 + 1)  (libff Fp2_model, non_residue = -1) */
struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    static Fq2 from_limbs(const u64 *l) { return {Fq::from_limbs(l), Fq::from_limbs(l + 4)}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fq2 &o) const { return !(*this == o); }
    Fq2 operator+(const Fq2 &o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2 &o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2 operator-() const { return {-c0, -c1}; }
    Fq2 operator*(const Fq2 &o) const {            /* Karatsuba, as libff */
        Fq aA = c0 * o.c0, bB = c1 * o.c1;
        return {aA - bB, (c0 + c1) * (o.c0 + o.c1) - aA - bB};
    }
    Fq2 sqr() const { Fq ab = c0 * c1; return {(c0 + c1) * (c0 - c1), ab + ab}; }
    Fq2 dbl() const { return *this + *this; }
    Fq2 inverse() const { Fq d = (c0.sqr() + c1.sqr()).inverse(); return {c0 * d, -(c1 * d)}; }
    bool canonical_lsb() const { return c0.canonical_lsb(); }
};

/* ------------------------------------------------------------------ short-Weierstrass a=0 Jacobian points (libff alt_bn128_G1 / alt_bn128_G2) */
template <class F> struct Aff { F x, y; bool inf; };

template <class F>
struct Jac {
    F X, Y, Z;
    static Jac zero() { return {F::zero(), F::one(), F::zero()}; }
    static Jac from_affine(const Aff<F> &a) { return a.inf ? zero() : Jac{a.x, a.y, F::one()}; }
    bool is_zero() const { return Z.is_zero(); }
    Jac dbl() const {                                /* dbl-2009-l */
        if (is_zero()) return *this;
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = ((X + B).sqr() - A - C).dbl();
        F E = A + A + A, Fv = E.sqr();
        F X3 = Fv - D.dbl();
        F eightC = C.dbl().dbl().dbl();
        F Y3 = E * (D - X3) - eightC;
        F Z3 = (Y * Z).dbl();
        return {X3, Y3, Z3};
    }
    Jac add(const Jac &o) const {                    /* add-2007-bl with the doubling / inverse cases handled */
        if (is_zero()) return o;
        if (o.is_zero()) return *this;
        F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        F U1 = X * Z2Z2, U2 = o.X * Z1Z1;
        F S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
        if (U1 == U2) { if (S1 == S2) return dbl(); return zero(); }
        F H = U2 - U1, I = H.dbl().sqr(), J = H * I;
        F r = (S2 - S1).dbl(), V = U1 * I;
        F X3 = r.sqr() - J - V.dbl();
        F Y3 = r * (V - X3) - (S1 * J).dbl();
        F Z3 = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
        return {X3, Y3, Z3};
    }
    Jac mixed_add(const Aff<F> &o) const {           /* madd-2007-bl */
        if (o.inf) return *this;
        if (is_zero()) return from_affine(o);
        F Z1Z1 = Z.sqr();
        F U2 = o.x * Z1Z1, S2 = o.y * Z * Z1Z1;
        if (X == U2) { if (Y == S2) return dbl(); return zero(); }
        F H = U2 - X, HH = H.sqr(), I = HH.dbl().dbl(), J = H * I;
        F r = (S2 - Y).dbl(), V = X * I;
        F X3 = r.sqr() - J - V.dbl();
        F Y3 = r * (V - X3) - (Y * J).dbl();
        F Z3 = (Z + H).sqr() - Z1Z1 - HH;
        return {X3, Y3, Z3};
    }
    Jac neg() const { return {X, -Y, Z}; }
    Aff<F> to_affine() const {
        if (is_zero()) return {F::zero(), F::zero(), true};
        F zi = Z.inverse(), zi2 = zi.sqr();
        return {X * zi2, Y * zi2 * zi, false};
    }
    /* double-and-add over canonical scalar limbs (libff scalar_mul / operator*) */
    Jac mul(const u64 *k, int limbs = 4) const {
        Jac r = zero(); bool started = false;
        for (int i = limbs * 64 - 1; i >= 0; --i) {
            if (started) r = r.dbl();
            if ((k[i / 64] >> (i % 64)) & 1) { r = r.add(*this); started = true; }
        }
        return r;
    }
};
typedef Jac<Fq> G1; typedef Aff<Fq> G1A;
typedef Jac<Fq2> G2; typedef Aff<Fq2> G2A;

/* ------------------------------------------------------------------ ABI (de)coding, see include/zkg.h */
static inline bool all_zero(const u64 *p, int n) { u64 o = 0; for (int i = 0; i < n; ++i) o |= p[i]; return o == 0; }
static G1A load_g1(const u64 *p) { if (all_zero(p, 8)) return {Fq::zero(), Fq::zero(), true}; return {Fq::from_limbs(p), Fq::from_limbs(p + 4), false}; }
static G2A load_g2(const u64 *p) { if (all_zero(p, 16)) return {Fq2::zero(), Fq2::zero(), true}; return {Fq2::from_limbs(p), Fq2::from_limbs(p + 8), false}; }
static void store_fq(u64 *p, const Fq &a) { memcpy(p, a.v, 32); }
static void store_fq2(u64 *p, const Fq2 &a) { memcpy(p, a.c0.v, 32); memcpy(p + 4, a.c1.v, 32); }
static void store_aff(u64 *p, const G1A &a) { if (a.inf) memset(p, 0, 64); else { store_fq(p, a.x); store_fq(p + 4, a.y); } }
static void store_aff(u64 *p, const G2A &a) { if (a.inf) memset(p, 0, 128); else { store_fq2(p, a.x); store_fq2(p + 8, a.y); } }
static void store_jac_norm(u64 *p, const G1 &g) {
    G1A a = g.to_affine();
    if (a.inf) { store_fq(p, Fq::zero()); store_fq(p + 4, Fq::one()); store_fq(p + 8, Fq::zero()); }
    else { store_fq(p, a.x); store_fq(p + 4, a.y); store_fq(p + 8, Fq::one()); }
}
static void store_jac_norm(u64 *p, const G2 &g) {
    G2A a = g.to_affine();
    if (a.inf) { store_fq2(p, Fq2::zero()); store_fq2(p + 8, Fq2::one()); store_fq2(p + 16, Fq2::zero()); }
    else { store_fq2(p, a.x); store_fq2(p + 8, a.y); store_fq2(p + 16, Fq2::one()); }
}
static G1 load_jac_g1(const u64 *p) { return {Fq::from_limbs(p), Fq::from_limbs(p + 4), Fq::from_limbs(p + 8)}; }
static G2 load_jac_g2(const u64 *p) { return {Fq2::from_limbs(p), Fq2::from_limbs(p + 8), Fq2::from_limbs(p + 16)}; }
template <class G> struct AffOf;
template <> struct AffOf<G1> { typedef G1A type; static G1A load(const u64 *p) { return load_g1(p); } enum { LIMBS = 8 }; };
template <> struct AffOf<G2> { typedef G2A type; static G2A load(const u64 *p) { return load_g2(p); } enum { LIMBS = 16 }; };

/* ------------------------------------------------------------------ libff::log2 (ceil) and bitreverse */
static size_t ceil_log2(size_t n) { size_t r = ((n & (n - 1)) == 0 ? 0 : 1); while (n > 1) { n >>= 1; r++; } return r; }
static size_t bitreverse(size_t n, size_t l) { size_t r = 0; for (size_t k = 0; k < l; ++k) { r = (r << 1) | (n & 1); n >>= 1; } return r; }

/* ------------------------------------------------------------------ libfqfft basic_radix2_domain<Fr> */
static const int FR_S = 28;
static Fr fr_root_of_unity() {                       /* 5^((r-1)/2^28): libff alt_bn128 init, Fr::root_of_unity */
    u64 e[4]; u64 one[4] = {1, 0, 0, 0}; sub4(e, R_LIMBS, one);
    for (int s = 0; s < FR_S; ++s) { for (int i = 0; i < 3; ++i) e[i] = (e[i] >> 1) | (e[i + 1] << 63); e[3] >>= 1; }
    return Fr::from_u64(5).pow(e, 4);
}
static Fr get_root_of_unity(size_t n) {              /* libff::get_root_of_unity */
    size_t logn = ceil_log2(n);
    Fr w = fr_root_of_unity();
    for (size_t i = FR_S; i > logn; --i) w = w.sqr();
    return w;
}
/* libfqfft _basic_serial_radix2_FFT */
static void serial_radix2_fft(Fr *a, size_t n, const Fr &omega) {
    size_t logn = ceil_log2(n);
    for (size_t k = 0; k < n; ++k) { size_t rk = bitreverse(k, logn); if (k < rk) std::swap(a[k], a[rk]); }
    size_t m = 1;
    for (size_t s = 1; s <= logn; ++s) {
        u64 e = n / (2 * m);
        Fr w_m = omega.pow_u64(e);
        for (size_t k = 0; k < n; k += 2 * m) {
            Fr w = Fr::one();
            for (size_t j = 0; j < m; ++j) {
                Fr t = w * a[k + j + m];
                a[k + j + m] = a[k + j] - t;
                a[k + j] += t;
                w *= w_m;
            }
        }
        m *= 2;
    }
}
/* libfqfft _multiply_by_coset */
static void multiply_by_coset(Fr *a, size_t n, const Fr &g) { Fr u = g; for (size_t i = 1; i < n; ++i) { a[i] *= u; u *= g; } }
static Fr coset_gen() { return Fr::from_u64(5); }    /* Fr::multiplicative_generator */

static void domain_FFT(Fr *a, size_t n) { serial_radix2_fft(a, n, get_root_of_unity(n)); }
static void domain_iFFT(Fr *a, size_t n) {
    serial_radix2_fft(a, n, get_root_of_unity(n).inverse());
    Fr sconst = Fr::from_u64(n).inverse();
    for (size_t i = 0; i < n; ++i) a[i] *= sconst;
}
static void domain_cosetFFT(Fr *a, size_t n, const Fr &g) { multiply_by_coset(a, n, g); domain_FFT(a, n); }
static void domain_icosetFFT(Fr *a, size_t n, const Fr &g) { domain_iFFT(a, n); multiply_by_coset(a, n, g.inverse()); }
/* basic_radix2_domain::divide_by_Z_on_coset */
static void divide_by_Z_on_coset(Fr *a, size_t n) {
    Fr zinv = (coset_gen().pow_u64(n) - Fr::one()).inverse();
    for (size_t i = 0; i < n; ++i) a[i] *= zinv;
}
/* libfqfft get_evaluation_domain(min_size) for min_size <= 2^28: the order of attempts is basic_radix2(min_size),
 * extended_radix2(min_size) (needs log m = s + 1: never here), step_radix2(min_size), then the same three on
 * big + rounded_small.  [UPSTREAM-RECALL]  step = true <=> step_radix2_domain. */
struct Domain {
    size_t m = 0; bool step = false;
    size_t big_m = 0, small_m = 0;            /* step_radix2_domain members */
    Fr omega, big_omega, small_omega;
};
static bool make_domain(size_t min_size, Domain &d) {
    if (min_size <= 1) return false;
    size_t lg = ceil_log2(min_size);
    if (lg > (size_t)FR_S) return false;
    size_t m; bool step;
    if (min_size == ((size_t)1 << lg)) { m = min_size; step = false; }
    else {
        size_t big = (size_t)1 << (lg - 1), small = min_size - big;
        size_t rounded_small = (size_t)1 << ceil_log2(small);
        if (small == rounded_small) { m = min_size; step = true; }
        else if (big == rounded_small) { m = big + rounded_small; step = false; }
        else { m = big + rounded_small; step = true; }
    }
    d.m = m; d.step = step;
    if (!step) { d.omega = get_root_of_unity(m); return true; }
    /* step_radix2_domain(m): big_m = 2^(log2(m)-1), small_m = m - big_m, omega = root of unity of order 2^log2(m) */
    d.big_m = (size_t)1 << (ceil_log2(m) - 1); d.small_m = m - d.big_m;
    d.omega = get_root_of_unity((size_t)1 << ceil_log2(m));
    d.big_omega = d.omega.sqr();
    d.small_omega = get_root_of_unity(d.small_m);
    return true;
}
static size_t evaluation_domain_size(size_t min_size) { Domain d; return make_domain(min_size, d) ? d.m : 0; }

/* step_radix2_domain::FFT: the polynomial is reduced mod (x^big - 1) (vector c) and, after the substitution
 * x -> omega x, mod (x^small - 1) (vector e); one radix-2 FFT each */
static void step_FFT(const Domain &D, Fr *a) {
    const size_t big = D.big_m, small = D.small_m;
    std::vector<Fr> c(big, Fr::zero()), d(big, Fr::zero());
    Fr omega_i = Fr::one();
    for (size_t i = 0; i < big; ++i) {
        c[i] = (i < small ? a[i] + a[i + big] : a[i]);
        d[i] = omega_i * (i < small ? a[i] - a[i + big] : a[i]);
        omega_i *= D.omega;
    }
    std::vector<Fr> e(small, Fr::zero());
    const size_t compr = (size_t)1 << (ceil_log2(big) - ceil_log2(small));
    for (size_t i = 0; i < small; ++i)
        for (size_t j = 0; j < compr; ++j) e[i] += d[i + j * small];
    serial_radix2_fft(c.data(), big, D.omega.sqr());
    if (small > 1) serial_radix2_fft(e.data(), small, get_root_of_unity(small));
    for (size_t i = 0; i < big; ++i) a[i] = c[i];
    for (size_t i = 0; i < small; ++i) a[i + big] = e[i];
}
/* step_radix2_domain::iFFT */
static void step_iFFT(const Domain &D, Fr *a) {
    const size_t big = D.big_m, small = D.small_m;
    std::vector<Fr> U0(a, a + big), U1(a + big, a + big + small);
    serial_radix2_fft(U0.data(), big, D.omega.sqr().inverse());
    if (small > 1) serial_radix2_fft(U1.data(), small, get_root_of_unity(small).inverse());
    const Fr U0_size_inv = Fr::from_u64(big).inverse();
    for (size_t i = 0; i < big; ++i) U0[i] *= U0_size_inv;
    const Fr U1_size_inv = Fr::from_u64(small).inverse();
    for (size_t i = 0; i < small; ++i) U1[i] *= U1_size_inv;
    std::vector<Fr> tmp = U0;
    Fr omega_i = Fr::one();
    for (size_t i = 0; i < big; ++i) { tmp[i] *= omega_i; omega_i *= D.omega; }
    for (size_t i = small; i < big; ++i) a[i] = U0[i];                       /* save A_suffix */
    const size_t compr = (size_t)1 << (ceil_log2(big) - ceil_log2(small));
    for (size_t i = 0; i < small; ++i)
        for (size_t j = 1; j < compr; ++j) U1[i] = U1[i] - tmp[i + j * small];
    const Fr omega_inv = D.omega.inverse();
    Fr omega_inv_i = Fr::one();
    for (size_t i = 0; i < small; ++i) { U1[i] *= omega_inv_i; omega_inv_i *= omega_inv; }
    const Fr over_two = Fr::from_u64(2).inverse();
    for (size_t i = 0; i < small; ++i) a[i] = (U0[i] + U1[i]) * over_two;   /* A_prefix */
    for (size_t i = 0; i < small; ++i) a[big + i] = (U0[i] - U1[i]) * over_two;   /* B2 */
}
/* step_radix2_domain::divide_by_Z_on_coset, Z(x) = (x^big - 1)(x^small - omega^small) */
static void step_divide_by_Z_on_coset(const Domain &D, Fr *P) {
    const size_t big = D.big_m, small = D.small_m;
    const Fr coset = coset_gen();
    const Fr Z0 = coset.pow_u64(big) - Fr::one();
    const Fr coset_to_small_m_times_Z0 = coset.pow_u64(small) * Z0;
    const Fr omega_to_small_m_times_Z0 = D.omega.pow_u64(small) * Z0;
    const Fr omega_to_2small_m = D.omega.pow_u64(2 * small);
    Fr elt = Fr::one();
    for (size_t i = 0; i < big; ++i) {
        P[i] *= (coset_to_small_m_times_Z0 * elt - omega_to_small_m_times_Z0).inverse();
        elt *= omega_to_2small_m;
    }
    const Fr co = coset * D.omega;
    const Fr Z1 = (co.pow_u64(big) - Fr::one()) * (co.pow_u64(small) - D.omega.pow_u64(small));
    const Fr Z1_inverse = Z1.inverse();
    for (size_t i = 0; i < small; ++i) P[big + i] *= Z1_inverse;
}
/* _basic_radix2_evaluate_all_lagrange_polynomials(m, t) */
static std::vector<Fr> basic_lagrange(size_t m, const Fr &t) {
    std::vector<Fr> u(m, Fr::zero());
    if (m == 1) { u[0] = Fr::one(); return u; }
    const Fr omega = get_root_of_unity(m);
    if (t.pow_u64(m) == Fr::one()) {                       /* t is in the domain: a unit vector */
        Fr omega_i = Fr::one();
        for (size_t i = 0; i < m; ++i) { if (omega_i == t) { u[i] = Fr::one(); return u; } omega_i *= omega; }
    }
    const Fr Z = t.pow_u64(m) - Fr::one();
    Fr l = Z * Fr::from_u64(m).inverse(), r = Fr::one();
    for (size_t i = 0; i < m; ++i) { u[i] = l * (t - r).inverse(); l *= omega; r *= omega; }
    return u;
}
/* evaluate_all_lagrange_polynomials(t) of the chosen domain */
static std::vector<Fr> domain_lagrange(const Domain &D, const Fr &t) {
    if (!D.step) return basic_lagrange(D.m, t);
    const size_t big = D.big_m, small = D.small_m;
    std::vector<Fr> inner_big = basic_lagrange(big, t), inner_small = basic_lagrange(small, t * D.omega.inverse());
    std::vector<Fr> result(D.m, Fr::zero());
    const Fr omega_to_small_m = D.omega.pow_u64(small);
    const Fr L0 = t.pow_u64(small) - omega_to_small_m;
    const Fr big_omega_to_small_m = D.big_omega.pow_u64(small);
    Fr elt = Fr::one();
    for (size_t i = 0; i < big; ++i) { result[i] = inner_big[i] * L0 * (elt - omega_to_small_m).inverse(); elt *= big_omega_to_small_m; }
    const Fr L1 = (t.pow_u64(big) - Fr::one()) * (D.omega.pow_u64(big) - Fr::one()).inverse();
    for (size_t i = 0; i < small; ++i) result[big + i] = L1 * inner_small[i];
    return result;
}
/* compute_vanishing_polynomial(t) */
static Fr domain_vanishing(const Domain &D, const Fr &t) {
    if (!D.step) return t.pow_u64(D.m) - Fr::one();
    return (t.pow_u64(D.big_m) - Fr::one()) * (t.pow_u64(D.small_m) - D.omega.pow_u64(D.small_m));
}
/* the four transforms on whichever domain was chosen (a has D.m entries) */
static void dom_FFT(const Domain &D, Fr *a) { if (D.step) step_FFT(D, a); else domain_FFT(a, D.m); }
static void dom_iFFT(const Domain &D, Fr *a) { if (D.step) step_iFFT(D, a); else domain_iFFT(a, D.m); }
static void dom_cosetFFT(const Domain &D, Fr *a, const Fr &g) { multiply_by_coset(a, D.m, g); dom_FFT(D, a); }
static void dom_icosetFFT(const Domain &D, Fr *a, const Fr &g) { dom_iFFT(D, a); multiply_by_coset(a, D.m, g.inverse()); }
static void dom_divide_by_Z_on_coset(const Domain &D, Fr *a) { if (D.step) step_divide_by_Z_on_coset(D, a); else divide_by_Z_on_coset(a, D.m); }

/* ------------------------------------------------------------------ libff multi_exp_inner<T, FieldT, multi_exp_method_BDLO12> */
static inline bool test_bit(const u64 *k, size_t b) { return b < 256 && ((k[b / 64] >> (b % 64)) & 1); }
static size_t num_bits(const u64 *k) { for (int i = 255; i >= 0; --i) if (test_bit(k, i)) return i + 1; return 0; }

template <class G>
static G multi_exp_bdlo12(const u64 *bases, const u64 *scalars, size_t length) {
    typedef typename AffOf<G>::type A;
    const int L = AffOf<G>::LIMBS;
    if (length == 0) return G::zero();
    size_t log2_length = ceil_log2(length);
    size_t c = log2_length - (log2_length / 3 - 2);          /* wraps exactly like libff for tiny lengths */
    if (c == 0) c = 1;
    size_t nb = 0;
    for (size_t i = 0; i < length; ++i) nb = std::max(nb, num_bits(scalars + 4 * i));
    size_t num_groups = (nb + c - 1) / c;
    G result = G::zero(); bool result_nonzero = false;
    std::vector<G> buckets((size_t)1 << c);
    std::vector<char> bucket_nonzero((size_t)1 << c);
    for (size_t k = num_groups - 1; k <= num_groups; k--) {
        if (result_nonzero) for (size_t i = 0; i < c; i++) result = result.dbl();
        std::fill(bucket_nonzero.begin(), bucket_nonzero.end(), 0);
        for (size_t i = 0; i < length; i++) {
            size_t id = 0;
            for (size_t j = 0; j < c; j++) if (test_bit(scalars + 4 * i, k * c + j)) id |= (size_t)1 << j;
            if (id == 0) continue;
            A b = AffOf<G>::load(bases + (size_t)L * i);
            if (b.inf) continue;
            if (bucket_nonzero[id]) buckets[id] = buckets[id].mixed_add(b);
            else { buckets[id] = G::from_affine(b); bucket_nonzero[id] = 1; }
        }
        G running_sum = G::zero(); bool running_sum_nonzero = false;
        for (size_t i = ((size_t)1 << c) - 1; i > 0; i--) {
            if (bucket_nonzero[i]) {
                if (running_sum_nonzero) running_sum = running_sum.add(buckets[i]);
                else { running_sum = buckets[i]; running_sum_nonzero = true; }
            }
            if (running_sum_nonzero) {
                if (result_nonzero) result = result.add(running_sum);
                else { result = running_sum; result_nonzero = true; }
            }
        }
    }
    return result;
}

/* libff multi_exp(..., chunks): split the range, one inner multi_exp per chunk, sum (MULTICORE build) */
template <class G>
static G multi_exp(const u64 *bases, const u64 *scalars, size_t length, int chunks) {
    const int L = AffOf<G>::LIMBS;
    if (chunks <= 1 || length < (size_t)chunks) return multi_exp_bdlo12<G>(bases, scalars, length);
    size_t one = length / chunks;
    std::vector<G> part(chunks);
#pragma omp parallel for num_threads(chunks)
    for (int i = 0; i < chunks; ++i) {
        size_t lo = i * one, hi = (i == chunks - 1 ? length : (i + 1) * one);
        part[i] = multi_exp_bdlo12<G>(bases + (size_t)L * lo, scalars + 4 * lo, hi - lo);
    }
    G r = G::zero();
    for (int i = 0; i < chunks; ++i) r = r.add(part[i]);
    return r;
}

/* libff multi_exp_with_mixed_addition: skip 0, add bases with scalar 1 directly, defer the rest */
template <class G>
static G multi_exp_with_mixed_addition(const u64 *bases, const u64 *scalars, size_t length, int chunks) {
    typedef typename AffOf<G>::type A;
    const int L = AffOf<G>::LIMBS;
    G acc = G::zero();
    std::vector<u64> p, g;
    for (size_t i = 0; i < length; ++i) {
        const u64 *s = scalars + 4 * i;
        if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
        if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) { A b = AffOf<G>::load(bases + (size_t)L * i); acc = acc.mixed_add(b); continue; }
        p.insert(p.end(), s, s + 4);
        g.insert(g.end(), bases + (size_t)L * i, bases + (size_t)L * (i + 1));
    }
    return acc.add(multi_exp<G>(g.data(), p.data(), p.size() / 4, chunks));
}

template <class G>
static G msm_naive(const u64 *bases, const u64 *scalars, size_t n) {
    const int L = AffOf<G>::LIMBS;
    G acc = G::zero();
    for (size_t i = 0; i < n; ++i) acc = acc.add(G::from_affine(AffOf<G>::load(bases + (size_t)L * i)).mul(scalars + 4 * i));
    return acc;
}

/* ------------------------------------------------------------------ R1CS helpers */
static Fr row_eval(const u32 *rowptr, const u32 *col, const u64 *val, u32 row, const std::vector<Fr> &z) {
    Fr acc = Fr::zero();
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; ++k) acc += Fr::from_limbs(val + 4 * (size_t)k) * z[col[k]];
    return acc;
}
static std::vector<Fr> padded_assignment(const zkg_r1cs &cs, const u64 *w) {
    std::vector<Fr> z(cs.num_variables + 1);
    z[0] = Fr::one();
    for (u32 i = 0; i < cs.num_variables; ++i) z[i + 1] = Fr::from_limbs(w + 4 * (size_t)i);
    return z;
}

/* libsnark r1cs_to_qap_witness_map(cs, primary, auxiliary, 0, 0, 0) -> coefficients_for_H[m+1] */
static int qap_witness_map(const zkg_r1cs &cs, const u64 *w, std::vector<Fr> &H, size_t &m_out) {
    Domain D;
    if (!make_domain((size_t)cs.num_constraints + cs.num_inputs + 1, D)) return 1;
    const size_t m = D.m;
    m_out = m;
    std::vector<Fr> z = padded_assignment(cs, w);
    std::vector<Fr> aA(m, Fr::zero()), aB(m, Fr::zero());
    for (u32 i = 0; i <= cs.num_inputs; ++i) aA[i + cs.num_constraints] = z[i];
    for (u32 i = 0; i < cs.num_constraints; ++i) {
        aA[i] += row_eval(cs.a_rowptr, cs.a_col, cs.a_val, i, z);
        aB[i] += row_eval(cs.b_rowptr, cs.b_col, cs.b_val, i, z);
    }
    dom_iFFT(D, aA.data()); dom_iFFT(D, aB.data());
    Fr g = coset_gen();
    dom_cosetFFT(D, aA.data(), g); dom_cosetFFT(D, aB.data(), g);
    std::vector<Fr> &H_tmp = aA;
    for (size_t i = 0; i < m; ++i) H_tmp[i] = aA[i] * aB[i];
    std::vector<Fr> aC(m, Fr::zero());
    for (u32 i = 0; i < cs.num_constraints; ++i) aC[i] += row_eval(cs.c_rowptr, cs.c_col, cs.c_val, i, z);
    dom_iFFT(D, aC.data()); dom_cosetFFT(D, aC.data(), g);
    for (size_t i = 0; i < m; ++i) H_tmp[i] = H_tmp[i] - aC[i];
    dom_divide_by_Z_on_coset(D, H_tmp.data());
    dom_icosetFFT(D, H_tmp.data(), g);
    H.assign(m + 1, Fr::zero());
    for (size_t i = 0; i < m; ++i) H[i] = H_tmp[i];
    return 0;
}

/* ------------------------------------------------------------------ serialisation (libff operator<< under BINARY_OUTPUT, MONTGOMERY_OUTPUT, point compression) */
static size_t ser_g1(uint8_t *out, const G1 &g) {
    G1A a = g.to_affine();
    Fq x = a.inf ? Fq::zero() : a.x, y = a.inf ? Fq::one() : a.y;
    out[0] = a.inf ? '1' : '0'; memcpy(out + 1, x.v, 32); out[33] = y.canonical_lsb() ? '1' : '0';
    return 34;
}
static size_t ser_g2(uint8_t *out, const G2 &g) {
    G2A a = g.to_affine();
    Fq2 x = a.inf ? Fq2::zero() : a.x, y = a.inf ? Fq2::one() : a.y;
    out[0] = a.inf ? '1' : '0'; memcpy(out + 1, x.c0.v, 32); memcpy(out + 33, x.c1.v, 32); out[65] = y.canonical_lsb() ? '1' : '0';
    return 66;
}

/* ------------------------------------------------------------------ fixed-base windowed batch (libff get_window_table / batch_exp, used by the generator) */
template <class G>
static void fixed_base_batch(const G &base, const u64 *scalars, size_t n, u64 *out_affine) {
    typedef typename AffOf<G>::type A;
    const int L = AffOf<G>::LIMBS;
    const int W = 8, NW = 32;
    std::vector<A> table((size_t)NW << W);
    G outer = base;
    for (int w = 0; w < NW; ++w) {
        G inner = G::zero();
        for (int i = 0; i < (1 << W); ++i) { table[((size_t)w << W) + i] = inner.to_affine(); inner = inner.add(outer); }
        outer = inner;                                        /* 2^W * previous outer */
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const u64 *k = scalars + 4 * (size_t)i;
        G acc = G::zero();
        for (int w = 0; w < NW; ++w) { unsigned d = (k[w / 8] >> ((w % 8) * 8)) & 0xFF; if (d) acc = acc.mixed_add(table[((size_t)w << W) + d]); }
        store_aff(out_affine + (size_t)L * i, acc.to_affine());
    }
}

static const u64 G1_ONE_CANON[2][4] = {{1, 0, 0, 0}, {2, 0, 0, 0}};
static const u64 G2_ONE_CANON[4][4] = {
    {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL},
    {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL},
    {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL},
    {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}};
static G1 g1_one() { return {Fq::from_canonical(G1_ONE_CANON[0]), Fq::from_canonical(G1_ONE_CANON[1]), Fq::one()}; }
static G2 g2_one() {
    return {{Fq::from_canonical(G2_ONE_CANON[0]), Fq::from_canonical(G2_ONE_CANON[1])},
            {Fq::from_canonical(G2_ONE_CANON[2]), Fq::from_canonical(G2_ONE_CANON[3])}, Fq2::one()};
}

/* ================================================================== C interface (ctypes) */
extern "C" {

/* field ops for golden-vector tests: field 0 = Fq, 1 = Fr; op 0 mul, 1 add, 2 sub, 3 inverse(a), 4 to_mont(a), 5 from_mont(a), 6 neg(a), 7 sqr(a) */
int zko_fp_op(int field, int op, const u64 *a, const u64 *b, u64 *out) {
#define DO(F)                                                                                        \
    {                                                                                                \
        F x = F::from_limbs(a), y = b ? F::from_limbs(b) : F::zero(), r;                             \
        switch (op) {                                                                                \
        case 0: r = x * y; break; case 1: r = x + y; break; case 2: r = x - y; break;                \
        case 3: r = x.inverse(); break; case 4: r = F::from_canonical(a); break;                     \
        case 5: x.to_canonical(out); return 0; case 6: r = -x; break; case 7: r = x.sqr(); break;    \
        default: return 1;                                                                           \
        }                                                                                            \
        memcpy(out, r.v, 32); return 0;                                                              \
    }
    if (field == 0) DO(Fq) else DO(Fr)
#undef DO
}
int zko_fq2_op(int op, const u64 *a, const u64 *b, u64 *out) {
    Fq2 x = Fq2::from_limbs(a), y = b ? Fq2::from_limbs(b) : Fq2::zero(), r;
    switch (op) { case 0: r = x * y; break; case 1: r = x + y; break; case 2: r = x - y; break; case 3: r = x.inverse(); break; case 7: r = x.sqr(); break; default: return 1; }
    store_fq2(out, r); return 0;
}
void zko_constants(u64 *q, u64 *r, u64 *q_inv, u64 *r_inv, u64 *q_one, u64 *r_one, u64 *q_r2, u64 *r_r2, u64 *root) {
    memcpy(q, PQ.p, 32); memcpy(r, PR.p, 32); *q_inv = PQ.inv; *r_inv = PR.inv;
    memcpy(q_one, PQ.one, 32); memcpy(r_one, PR.one, 32); memcpy(q_r2, PQ.r2, 32); memcpy(r_r2, PR.r2, 32);
    Fr w = fr_root_of_unity(); w.to_canonical(root);
}
void zko_g1_generator(u64 out[8]) { store_aff(out, g1_one().to_affine()); }
void zko_g2_generator(u64 out[16]) { store_aff(out, g2_one().to_affine()); }

int zko_g1_add(const u64 a[8], const u64 b[8], u64 out[12]) { store_jac_norm(out, G1::from_affine(load_g1(a)).add(G1::from_affine(load_g1(b)))); return 0; }
int zko_g2_add(const u64 a[16], const u64 b[16], u64 out[24]) { store_jac_norm(out, G2::from_affine(load_g2(a)).add(G2::from_affine(load_g2(b)))); return 0; }
int zko_g1_mixed_add(const u64 a[8], const u64 b[8], u64 out[12]) { store_jac_norm(out, G1::from_affine(load_g1(a)).dbl().mixed_add(load_g1(b))); return 0; }   /* (2a)+b through madd */
int zko_g1_scalar_mul(const u64 base[8], const u64 k[4], u64 out[12]) { store_jac_norm(out, G1::from_affine(load_g1(base)).mul(k)); return 0; }
int zko_g2_scalar_mul(const u64 base[16], const u64 k[4], u64 out[24]) { store_jac_norm(out, G2::from_affine(load_g2(base)).mul(k)); return 0; }
int zko_g1_on_curve(const u64 a[8]) { G1A p = load_g1(a); if (p.inf) return 1; return p.y.sqr() == p.x.sqr() * p.x + Fq::from_u64(3); }
int zko_g2_on_curve(const u64 a[16]) {
    G2A p = load_g2(a); if (p.inf) return 1;
    Fq2 b = Fq2{Fq::from_u64(3), Fq::zero()} * Fq2{Fq::from_u64(9), Fq::one()}.inverse();
    return p.y.sqr() == p.x.sqr() * p.x + b;
}
int zko_g1_sum(const u64 *pts_jac, size_t n, u64 out[12]) { G1 a = G1::zero(); for (size_t i = 0; i < n; ++i) a = a.add(load_jac_g1(pts_jac + 12 * i)); store_jac_norm(out, a); return 0; }
int zko_g2_sum(const u64 *pts_jac, size_t n, u64 out[24]) { G2 a = G2::zero(); for (size_t i = 0; i < n; ++i) a = a.add(load_jac_g2(pts_jac + 24 * i)); store_jac_norm(out, a); return 0; }

int zko_g1_fixed_base(const u64 base[8], const u64 *scalars, size_t n, u64 *out_affine) { fixed_base_batch<G1>(G1::from_affine(load_g1(base)), scalars, n, out_affine); return 0; }
int zko_g2_fixed_base(const u64 base[16], const u64 *scalars, size_t n, u64 *out_affine) { fixed_base_batch<G2>(G2::from_affine(load_g2(base)), scalars, n, out_affine); return 0; }

/* libfqfft basic_radix2_domain FFT / iFFT / cosetFFT / icosetFFT (g = 5) on Montgomery Fr, in place */
int zko_fft(u64 *a, unsigned logn, int inverse, int coset) {
    if (logn > (unsigned)FR_S) return 1;
    size_t n = (size_t)1 << logn; Fr *v = reinterpret_cast<Fr *>(a);
    if (n == 1) return 0;
    Fr g = coset_gen();
    if (!inverse) { if (coset) domain_cosetFFT(v, n, g); else domain_FFT(v, n); }
    else { if (coset) domain_icosetFFT(v, n, g); else domain_iFFT(v, n); }
    return 0;
}
size_t zko_evaluation_domain_size(size_t min_size) { return evaluation_domain_size(min_size); }
int zko_evaluation_domain_is_step(size_t min_size) { Domain d; return make_domain(min_size, d) && d.step; }
/* the same four transforms on get_evaluation_domain(m) for an m the rule maps to itself (a power of two or 2^a + 2^b) */
int zko_domain_fft(u64 *a, size_t m, int inverse, int coset) {
    Domain D; if (!make_domain(m, D) || D.m != m) return 1;
    Fr *v = reinterpret_cast<Fr *>(a); Fr g = coset_gen();
    if (!inverse) { if (coset) dom_cosetFFT(D, v, g); else dom_FFT(D, v); }
    else { if (coset) dom_icosetFFT(D, v, g); else dom_iFFT(D, v); }
    return 0;
}
/* evaluate_all_lagrange_polynomials(t) (out: m Montgomery Fr) and compute_vanishing_polynomial(t); t canonical limbs */
int zko_domain_lagrange(size_t m, const u64 t_[4], u64 *out, u64 *z_out) {
    Domain D; if (!make_domain(m, D) || D.m != m) return 1;
    Fr t = Fr::from_canonical(t_);
    std::vector<Fr> u = domain_lagrange(D, t);
    memcpy(out, u.data(), m * 32);
    Fr z = domain_vanishing(D, t); memcpy(z_out, z.v, 32);
    return 0;
}

/* method: 0 naive double-and-add, 1 BDLO12 bucket method (chunks = threads), 2 multi_exp_with_mixed_addition */
int zko_msm_g1(const u64 *bases, const u64 *scalars, size_t n, u64 out[12], int method, int chunks) {
    G1 r = method == 0 ? msm_naive<G1>(bases, scalars, n) : method == 1 ? multi_exp<G1>(bases, scalars, n, chunks) : multi_exp_with_mixed_addition<G1>(bases, scalars, n, chunks);
    store_jac_norm(out, r); return 0;
}
int zko_msm_g2(const u64 *bases, const u64 *scalars, size_t n, u64 out[24], int method, int chunks) {
    G2 r = method == 0 ? msm_naive<G2>(bases, scalars, n) : method == 1 ? multi_exp<G2>(bases, scalars, n, chunks) : multi_exp_with_mixed_addition<G2>(bases, scalars, n, chunks);
    store_jac_norm(out, r); return 0;
}

int zko_r1cs_is_satisfied(const zkg_r1cs *cs, const u64 *w) {
    std::vector<Fr> z = padded_assignment(*cs, w);
    for (u32 i = 0; i < cs->num_constraints; ++i)
        if (row_eval(cs->a_rowptr, cs->a_col, cs->a_val, i, z) * row_eval(cs->b_rowptr, cs->b_col, cs->b_val, i, z) != row_eval(cs->c_rowptr, cs->c_col, cs->c_val, i, z)) return 0;
    return 1;
}
int zko_qap_witness_h(const zkg_r1cs *cs, const u64 *w, u64 *h_out) {
    std::vector<Fr> H; size_t m;
    if (qap_witness_map(*cs, w, H, m)) return 1;
    memcpy(h_out, H.data(), (m + 1) * 32); return 0;
}

/* libsnark r1cs_gg_ppzksnark_prover (snark.cpp:126) with explicit (r, s) */
int zko_groth16_prove(const zkg_pk *pk, const u64 *w, const u64 r_[4], const u64 s_[4], int check_satisfied,
                      uint8_t *proof_out, size_t *proof_len, int chunks) {
    const zkg_r1cs &cs = pk->cs;
    if (check_satisfied && !zko_r1cs_is_satisfied(&cs, w)) return 1;          // the reference's own return value at this gate (libsnark_wrapper.cpp:233-240)
    std::vector<Fr> H; size_t m;
    if (qap_witness_map(cs, w, H, m)) return 2;
    if (m != (pk->domain_size ? (size_t)pk->domain_size : ((size_t)1 << pk->log_m))) return 2;
    Fr r = Fr::from_limbs(r_), s = Fr::from_limbs(s_);
    size_t n = cs.num_variables, l = cs.num_inputs;
    std::vector<u64> zc((n + 1) * 4), hc((m - 1) * 4);           /* as_bigint() of the scalars */
    { std::vector<Fr> z = padded_assignment(cs, w); for (size_t i = 0; i <= n; ++i) z[i].to_canonical(&zc[4 * i]); }
    for (size_t i = 0; i + 1 < m; ++i) H[i].to_canonical(&hc[4 * i]);
    G1 At = multi_exp_with_mixed_addition<G1>(pk->A_query, zc.data(), n + 1, chunks);
    G1 Bt1 = multi_exp_with_mixed_addition<G1>(pk->B_g1, zc.data(), n + 1, chunks);
    G2 Bt2 = multi_exp_with_mixed_addition<G2>(pk->B_g2, zc.data(), n + 1, chunks);
    G1 Ht = multi_exp<G1>(pk->H_query, hc.data(), m - 1, chunks);
    G1 Lt = multi_exp_with_mixed_addition<G1>(pk->L_query, zc.data() + 4 * (l + 1), n - l, chunks);
    u64 rc[4], sc[4], rsc[4];
    r.to_canonical(rc); s.to_canonical(sc); (r * s).to_canonical(rsc);
    G1 alpha = G1::from_affine(load_g1(pk->alpha_g1)), beta1 = G1::from_affine(load_g1(pk->beta_g1)), delta1 = G1::from_affine(load_g1(pk->delta_g1));
    G2 beta2 = G2::from_affine(load_g2(pk->beta_g2)), delta2 = G2::from_affine(load_g2(pk->delta_g2));
    G1 gA = alpha.add(At).add(delta1.mul(rc));
    G1 gB1 = beta1.add(Bt1).add(delta1.mul(sc));
    G2 gB2 = beta2.add(Bt2).add(delta2.mul(sc));
    G1 gC = Ht.add(Lt).add(gA.mul(sc)).add(gB1.mul(rc)).add(delta1.mul(rsc).neg());
    size_t off = 0;
    off += ser_g1(proof_out + off, gA); off += ser_g2(proof_out + off, gB2); off += ser_g1(proof_out + off, gC);
    *proof_len = off;
    return 0;
}

/* Toy generator with KNOWN trapdoor, shaped like libsnark r1cs_gg_ppzksnark_generator (snark.cpp:91):
 * the caller passes the (already A/B-swapped if beneficial) system; outputs are caller-allocated flat
 * arrays as in zkg_pk.  td = t, alpha, beta, gamma, delta (canonical limbs, 5 x 4).
 * Also emits the QAP evaluations At/Bt/Ct (n+1 Montgomery Fr each) and Zt for known-trapdoor checks. */
int zko_groth16_setup(const zkg_r1cs *cs, const u64 *td, u64 *alpha_g1, u64 *beta_g1, u64 *delta_g1, u64 *beta_g2, u64 *delta_g2,
                      u64 *A_query, u64 *B_g1, u64 *B_g2, u64 *H_query, u64 *L_query, u64 *At_out, u64 *Bt_out, u64 *Ct_out, u64 *Zt_out) {
    Domain D;
    if (!make_domain((size_t)cs->num_constraints + cs->num_inputs + 1, D)) return 1;
    size_t m = D.m;
    size_t n = cs->num_variables, l = cs->num_inputs, C = cs->num_constraints;
    Fr t = Fr::from_canonical(td), alpha = Fr::from_canonical(td + 4), beta = Fr::from_canonical(td + 8), delta = Fr::from_canonical(td + 16);
    Fr Zt = domain_vanishing(D, t);
    std::vector<Fr> u = domain_lagrange(D, t);
    std::vector<Fr> At(n + 1, Fr::zero()), Bt(n + 1, Fr::zero()), Ct(n + 1, Fr::zero());
    for (size_t i = 0; i <= l; ++i) At[i] = u[C + i];
    for (size_t i = 0; i < C; ++i) {
        for (u32 k = cs->a_rowptr[i]; k < cs->a_rowptr[i + 1]; ++k) At[cs->a_col[k]] += u[i] * Fr::from_limbs(cs->a_val + 4 * (size_t)k);
        for (u32 k = cs->b_rowptr[i]; k < cs->b_rowptr[i + 1]; ++k) Bt[cs->b_col[k]] += u[i] * Fr::from_limbs(cs->b_val + 4 * (size_t)k);
        for (u32 k = cs->c_rowptr[i]; k < cs->c_rowptr[i + 1]; ++k) Ct[cs->c_col[k]] += u[i] * Fr::from_limbs(cs->c_val + 4 * (size_t)k);
    }
    Fr dinv = delta.inverse();
    auto canon = [](const std::vector<Fr> &v) { std::vector<u64> o(v.size() * 4); for (size_t i = 0; i < v.size(); ++i) v[i].to_canonical(&o[4 * i]); return o; };
    G1 g1 = g1_one(); G2 g2 = g2_one();
    store_aff(alpha_g1, g1.mul(td + 4).to_affine()); store_aff(beta_g1, g1.mul(td + 8).to_affine()); store_aff(delta_g1, g1.mul(td + 16).to_affine());
    store_aff(beta_g2, g2.mul(td + 8).to_affine()); store_aff(delta_g2, g2.mul(td + 16).to_affine());
    std::vector<u64> Ac = canon(At), Bc = canon(Bt);
    fixed_base_batch<G1>(g1, Ac.data(), n + 1, A_query);
    fixed_base_batch<G1>(g1, Bc.data(), n + 1, B_g1);
    fixed_base_batch<G2>(g2, Bc.data(), n + 1, B_g2);
    std::vector<Fr> Hs(m - 1), Ls(n - l);
    { Fr ti = Fr::one(), zd = Zt * dinv; for (size_t i = 0; i + 1 < m; ++i) { Hs[i] = ti * zd; ti *= t; } }
    for (size_t i = l + 1; i <= n; ++i) Ls[i - l - 1] = (beta * At[i] + alpha * Bt[i] + Ct[i]) * dinv;
    std::vector<u64> Hc = canon(Hs), Lc = canon(Ls);
    fixed_base_batch<G1>(g1, Hc.data(), m - 1, H_query);
    fixed_base_batch<G1>(g1, Lc.data(), n - l, L_query);
    if (At_out) memcpy(At_out, At.data(), (n + 1) * 32);
    if (Bt_out) memcpy(Bt_out, Bt.data(), (n + 1) * 32);
    if (Ct_out) memcpy(Ct_out, Ct.data(), (n + 1) * 32);
    if (Zt_out) memcpy(Zt_out, Zt.v, 32);
    return 0;
}

/* Writer of the libsnark proving-key byte format (operator<< of r1cs_gg_ppzksnark_proving_key under BINARY_OUTPUT,
 * MONTGOMERY_OUTPUT and point compression; exported by libsnark_export_pk, libsnark_wrapper.cpp:146-157) from flat arrays.
 * Returns the number of bytes needed; writes only if cap is large enough.  Test infrastructure for zkg_crs_upload_blob. */
size_t zko_pk_write_blob(const zkg_pk *pk, uint8_t *out, size_t cap) {
    std::vector<uint8_t> buf;
    auto dec = [&](size_t v) { char t[32]; int n = snprintf(t, sizeof t, "%zu\n", v); buf.insert(buf.end(), t, t + n); };
    auto g1 = [&](const u64 *p) { uint8_t t[34]; ser_g1(t, G1::from_affine(load_g1(p))); buf.insert(buf.end(), t, t + 34); };
    auto g2 = [&](const u64 *p) { uint8_t t[66]; ser_g2(t, G2::from_affine(load_g2(p))); buf.insert(buf.end(), t, t + 66); };
    const zkg_r1cs &cs = pk->cs;
    size_t n = cs.num_variables, l = cs.num_inputs, m = pk->domain_size ? (size_t)pk->domain_size : ((size_t)1 << pk->log_m);
    g1(pk->alpha_g1); g1(pk->beta_g1); g2(pk->beta_g2); g1(pk->delta_g1); g2(pk->delta_g2);
    dec(n + 1); for (size_t i = 0; i <= n; ++i) g1(pk->A_query + 8 * i);
    std::vector<size_t> idx;
    for (size_t i = 0; i <= n; ++i) if (!all_zero(pk->B_g2 + 16 * i, 16) || !all_zero(pk->B_g1 + 8 * i, 8)) idx.push_back(i);
    dec(n + 1); dec(idx.size()); for (size_t i : idx) dec(i);
    dec(idx.size()); for (size_t i : idx) { g2(pk->B_g2 + 16 * i); g1(pk->B_g1 + 8 * i); }
    dec(m - 1); for (size_t i = 0; i + 1 < m; ++i) g1(pk->H_query + 8 * i);
    dec(n - l); for (size_t i = 0; i < n - l; ++i) g1(pk->L_query + 8 * i);
    dec(l); dec(n - l); dec(cs.num_constraints);
    for (u32 c = 0; c < cs.num_constraints; ++c) {
        const u32 *rp[3] = {cs.a_rowptr, cs.b_rowptr, cs.c_rowptr}, *col[3] = {cs.a_col, cs.b_col, cs.c_col};
        const u64 *val[3] = {cs.a_val, cs.b_val, cs.c_val};
        for (int k = 0; k < 3; ++k) {
            dec(rp[k][c + 1] - rp[k][c]);
            for (u32 t = rp[k][c]; t < rp[k][c + 1]; ++t) { dec(col[k][t]); const uint8_t *b = (const uint8_t *)(val[k] + 4 * (size_t)t); buf.insert(buf.end(), b, b + 32); }
        }
    }
    if (out && cap >= buf.size()) memcpy(out, buf.data(), buf.size());
    return buf.size();
}

int zko_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} /* extern "C" */
