// fp.hip.hpp — alt_bn128 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Replaces libff's Fp_model<4, modulus> / Fp2_model (alt_bn128_Fq, alt_bn128_Fr, alt_bn128_Fq2),
// reached from r1cs_gg_ppzksnark_prover at /root/reference/zklaim/snark.cpp:126.
//
// Memory format == libff's: 4 x u64 little-endian limbs, Montgomery form, R = 2^256.
// Register format on the device: 8 x u32 limbs.  CDNA4 has no 64x64 multiplier; the widest
// integer multiply-add is v_mad_u64_u32 (32x32+64 -> 64), so "64-bit Montgomery limbs" are
// physically pairs of 32-bit limbs and the multiplication below is written for that unit.
// Host code (final window combine, proof assembly) uses the same struct with a 64-bit-limb
// multiplication via unsigned __int128.
//
// Lazy reduction on the device: p < 2^254, so 4p < R = 2^256 and the Montgomery product of two values below 2p is again
// below 2p WITHOUT the final conditional subtraction.  Device values therefore live in [0, 2p) ("lazy"); add/sub fold
// with 2p; is_zero()/== know that 0 is represented by 0 or p; normalized() brings a value to [0, p) and is applied
// wherever a value leaves the registers for global memory, so everything in HBM and everything the host sees is canonical.
#pragma once
#include <stdint.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mont_asm.inc"

#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__

// Issue priority of a wavefront (s_setprio, 0 .. 3).  A proof's kernels share compute units across streams: the witness multi-exponentiations
// run beside the transforms and the H query's multi-exponentiation, which are the critical path.  The oldest wavefront wins the arbitration by
// default; the kernels on the critical path raise theirs (2; the digit sort's small dependent launches 3) so that the work with slack waits.
__device__ __forceinline__ void crit_wave_priority(int on) { if (on) __builtin_amdgcn_s_setprio(2); }

namespace zk {

struct FqParams {
    static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t TWO_P[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
    static constexpr uint32_t INV32 = 0xe4866389u;
    static constexpr uint64_t INV64 = 0x87d20782e4866389ull;
};
struct FrParams {
    static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t TWO_P[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
    static constexpr uint32_t INV32 = 0xefffffffu;
    static constexpr uint64_t INV64 = 0xc2e1f593efffffffull;
};

template <class PR>
struct alignas(16) Fp {
    uint32_t v[8];

    static ZK_HD Fp zero() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = 0; return r; }
    static ZK_HD Fp one() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = PR::ONE[i]; return r; }
    static ZK_HD Fp r2() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = PR::R2[i]; return r; }
#if defined(__HIP_DEVICE_COMPILE__)
    ZK_HD bool is_zero() const {                     // lazy representation: 0 or p
        uint32_t o = 0, q = 0;
        for (int i = 0; i < 8; ++i) { o |= v[i]; q |= v[i] ^ PR::P[i]; }
        return o == 0 || q == 0;
    }
    ZK_HD bool operator==(const Fp &b) const { return (*this - b).is_zero(); }
    ZK_HD Fp normalized() const { return reduce_once(v); }
#else
    ZK_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= v[i]; return o == 0; }
    ZK_HD bool operator==(const Fp &b) const { uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= v[i] ^ b.v[i]; return o == 0; }
    ZK_HD Fp normalized() const { return *this; }
#endif
    ZK_HD bool operator!=(const Fp &b) const { return !(*this == b); }

    static constexpr uint64_t p64(int j) { return PR::P[2 * j] | ((uint64_t)PR::P[2 * j + 1] << 32); }      // host arithmetic: 64-bit limbs of p
    // r = (t >= p) ? t - p : t      (t < 2p)
    static ZK_HD Fp reduce_once(const uint32_t t[8]) {
        uint32_t s[8]; uint32_t br = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { uint64_t d = (uint64_t)t[j] - PR::P[j] - br; s[j] = (uint32_t)d; br = (uint32_t)(d >> 32) & 1u; }
        Fp r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[j] = br ? t[j] : s[j];
        return r;
    }
    friend ZK_HD Fp operator+(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        // lazy: a, b < 2p, a + b < 4p < 2^256; folded once with 2p.  Two interleaved carry chains (tools/gen_mont_asm.py): 34 full-rate
        // issue slots against the compiler's ~90 instructions with 32 half-rate 64-bit adds.
        Fp r; uint64_t c;
        if constexpr (std::is_same<PR, FqParams>::value) {
            asm(ZK_FP_ADD_ASM_FQ
                : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&s"(c)
                : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
                  "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
                : ZK_FP_ADDSUB_CLOBBERS);
        } else {
            asm(ZK_FP_ADD_ASM_FR
                : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&s"(c)
                : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
                  "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
                : ZK_FP_ADDSUB_CLOBBERS);
        }
        return r;
#else
        // host: canonical values, four 64-bit limbs with add-with-carry chains.  p < 2^254: a + b < 2^255 never carries out
        uint64_t A[4], B[4], t[4], s[4]; memcpy(A, a.v, 32); memcpy(B, b.v, 32);
        unsigned long long c = 0, br = 0;
        for (int j = 0; j < 4; ++j) t[j] = __builtin_addcll(A[j], B[j], c, &c);
        for (int j = 0; j < 4; ++j) s[j] = __builtin_subcll(t[j], p64(j), br, &br);
        const uint64_t keep = 0ull - (uint64_t)br;                               // borrowed: t < p already (branch-free select)
        for (int j = 0; j < 4; ++j) s[j] ^= (s[j] ^ t[j]) & keep;
        Fp r; memcpy(r.v, s, 32);
        return r;
#endif
    }
    friend ZK_HD Fp operator-(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r; uint64_t c;                                // lazy: a - b, plus 2p when it borrowed; same interleaved scheme as operator+
        if constexpr (std::is_same<PR, FqParams>::value) {
            asm(ZK_FP_SUB_ASM_FQ
                : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&s"(c)
                : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
                  "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
                : ZK_FP_ADDSUB_CLOBBERS);
        } else {
            asm(ZK_FP_SUB_ASM_FR
                : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&s"(c)
                : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
                  "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
                : ZK_FP_ADDSUB_CLOBBERS);
        }
        return r;
#else
        uint64_t A[4], B[4], t[4]; memcpy(A, a.v, 32); memcpy(B, b.v, 32);
        unsigned long long br = 0, c = 0;
        for (int j = 0; j < 4; ++j) t[j] = __builtin_subcll(A[j], B[j], br, &br);
        const uint64_t mask = 0ull - (uint64_t)br;                               // borrowed: add p back
        for (int j = 0; j < 4; ++j) t[j] = __builtin_addcll(t[j], p64(j) & mask, c, &c);
        Fp r; memcpy(r.v, t, 32);
        return r;
#endif
    }
    ZK_HD Fp neg() const { return zero() - *this; }      // 0 - 0 = 0; otherwise p - a (host) / 2p - a (device, lazy)
    ZK_HD Fp dbl() const { return *this + *this; }

    // Montgomery product a*b*R^-1 mod p.
    friend ZK_HD Fp operator*(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Hand-scheduled gfx950 stream (tools/gen_mont_asm.py): product scanning with a 96-bit column accumulator,
        // one v_mad_u64_u32 + one v_addc_co_u32 per partial product.  Result in [0, 2p).
        uint32_t t[8]; uint64_t c0, c1, c2;
        if constexpr (std::is_same<PR, FqParams>::value) {
            asm(ZK_MONT_MUL_ASM_FQ
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&s"(c0), "=&s"(c1), "=&s"(c2)
                : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
                  "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
                : ZK_MONT_MUL_CLOBBERS);
        } else {
            asm(ZK_MONT_MUL_ASM_FR
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&s"(c0), "=&s"(c1), "=&s"(c2)
                : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
                  "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
                : ZK_MONT_MUL_CLOBBERS);
        }
        Fp r;                                            // lazy: a, b < 2p  =>  t < 2p, no final subtraction
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[j] = t[j];
        return r;
#else
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], t[4] = {0, 0, 0, 0}; memcpy(A, a.v, 32); memcpy(B, b.v, 32);
        constexpr uint64_t P4[4] = {p64(0), p64(1), p64(2), p64(3)};
        for (int i = 0; i < 4; ++i) {
            u128 x = (u128)A[0] * B[i] + t[0];
            uint64_t m = (uint64_t)x * PR::INV64;
            u128 y = (u128)m * P4[0] + (uint64_t)x;
            uint64_t c = (uint64_t)(x >> 64), c2 = (uint64_t)(y >> 64);
            for (int j = 1; j < 4; ++j) {
                x = (u128)A[j] * B[i] + t[j] + c; c = (uint64_t)(x >> 64);
                y = (u128)m * P4[j] + (uint64_t)x + c2; c2 = (uint64_t)(y >> 64);
                t[j - 1] = (uint64_t)y;
            }
            t[3] = c + c2;
        }
        uint64_t s[4]; unsigned long long br = 0;
        for (int j = 0; j < 4; ++j) s[j] = __builtin_subcll(t[j], P4[j], br, &br);
        const uint64_t keep = 0ull - (uint64_t)br;                               // borrowed: t < p already (branch-free select)
        for (int j = 0; j < 4; ++j) s[j] ^= (s[j] ^ t[j]) & keep;
        Fp r; memcpy(r.v, s, 32);
        return r;
#endif
    }
    ZK_HD Fp sqr() const { return (*this) * (*this); }
    ZK_HD Fp from_mont() const { Fp o = zero(); o.v[0] = 1; return ((*this) * o).normalized(); }      // canonical value
    ZK_HD Fp to_mont() const { return (*this) * r2(); }
    ZK_HD Fp &operator+=(const Fp &b) { *this = *this + b; return *this; }
    ZK_HD Fp &operator-=(const Fp &b) { *this = *this - b; return *this; }
    ZK_HD Fp &operator*=(const Fp &b) { *this = *this * b; return *this; }

    // x^e, e given as little-endian u32 limbs (host helpers and the one-off device inversions)
    __host__ __device__ __attribute__((noinline)) Fp pow(const uint32_t *e, int nlimbs) const {
        Fp r = one();
        for (int i = nlimbs * 32 - 1; i >= 0; --i) {
            r = r.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1u) r = r * (*this);
        }
        return r;
    }
    ZK_HD Fp pow_u64(uint64_t e) const { uint32_t l[2] = {(uint32_t)e, (uint32_t)(e >> 32)}; return pow(l, 2); }
    ZK_HD Fp inverse() const {                              // Fermat: x^(p-2)
        uint32_t e[8];
        for (int i = 0; i < 8; ++i) e[i] = PR::P[i];
        e[0] -= 2;                                          // P[0] is odd and > 2: no borrow
        return pow(e, 8);
    }
    static ZK_HD Fp from_u64(uint64_t x) { Fp r = zero(); r.v[0] = (uint32_t)x; r.v[1] = (uint32_t)(x >> 32); return r.to_mont(); }
};

typedef Fp<FqParams> Fq;
typedef Fp<FrParams> Fr;

// Fq2 = Fq[u]/(u^2 + 1)   (libff Fp2_model with non_residue = -1)
struct Fq2 {
    Fq c0, c1;
    static ZK_HD Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static ZK_HD Fq2 one() { return {Fq::one(), Fq::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool operator==(const Fq2 &b) const { return c0 == b.c0 && c1 == b.c1; }
    ZK_HD bool operator!=(const Fq2 &b) const { return !(*this == b); }
    friend ZK_HD Fq2 operator+(const Fq2 &a, const Fq2 &b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
    friend ZK_HD Fq2 operator-(const Fq2 &a, const Fq2 &b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
    friend ZK_HD Fq2 operator*(const Fq2 &a, const Fq2 &b) {          // Karatsuba: 3 base multiplications
        Fq aa = a.c0 * b.c0, bb = a.c1 * b.c1;
        return {aa - bb, (a.c0 + a.c1) * (b.c0 + b.c1) - aa - bb};
    }
    ZK_HD Fq2 sqr() const { Fq ab = c0 * c1; return {(c0 + c1) * (c0 - c1), ab + ab}; }
    ZK_HD Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    ZK_HD Fq2 normalized() const { return {c0.normalized(), c1.normalized()}; }
    ZK_HD Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    ZK_HD Fq2 inverse() const { Fq d = (c0.sqr() + c1.sqr()).inverse(); return {c0 * d, (c1 * d).neg()}; }
    ZK_HD Fq2 &operator+=(const Fq2 &b) { *this = *this + b; return *this; }
    ZK_HD Fq2 &operator-=(const Fq2 &b) { *this = *this - b; return *this; }
    ZK_HD Fq2 &operator*=(const Fq2 &b) { *this = *this * b; return *this; }
};

}  // namespace zk
