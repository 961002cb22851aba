// curve.hip.hpp — alt_bn128 G1 (over Fq) and G2 (over Fq2) group law, y^2 = x^3 + b, a = 0.
//
// Replaces libff's alt_bn128_G1 / alt_bn128_G2 add, mixed_add, dbl, to_affine_coordinates
// (reached from the multi_exp calls inside r1cs_gg_ppzksnark_prover, /root/reference/zklaim/snark.cpp:126).
//
// libff keeps Jacobian (X,Y,Z).  The accumulators here use extended Jacobian "XYZZ"
// coordinates (X, Y, ZZ, ZZZ with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): the mixed addition that
// dominates the MSM costs 8M+2S instead of 7M+4S and needs no field doubling chain.  Group
// elements are exact, so any coordinate system gives the same affine result; outputs cross the
// ABI only in normalised form (include/zkg.h).  Infinity: ZZ == 0 (XYZZ), x == y == 0 (affine).
#pragma once
#include "fp.hip.hpp"

// Device code inlines every group operation: a G2 point is 64 registers, more than the calling convention passes in
// VGPRs, so an out-of-line add or dbl forces its operands (and, through `this`, whole accumulators) into scratch
// memory — measured as 0.4-1.5 KB of private segment per lane in the G2 reduction kernels.  Host code keeps the G2
// operations out of line (code size); scalar multiplication and to_affine stay out of line everywhere.
#define ZK_HD_NOINLINE __host__ __device__ __attribute__((noinline))

namespace zk {

template <class F> struct Affine {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static ZK_HD Affine inf() { return {F::zero(), F::zero()}; }
    ZK_HD Affine neg() const { return {x, y.neg()}; }
    ZK_HD Affine normalized() const { return {x.normalized(), y.normalized()}; }       // canonical limbs, for stores to global memory
};

template <class F> struct XYZZ {
    F x, y, zz, zzz;
#if defined(__HIP_DEVICE_COMPILE__)
    static constexpr bool kInlineAll = true;
#else
    static constexpr bool kInlineAll = sizeof(F) == sizeof(Fq);       // host: G1 inline, G2 group ops out of line
#endif
    static ZK_HD XYZZ inf() { return {F::zero(), F::one(), F::zero(), F::zero()}; }
    static ZK_HD XYZZ from_affine(const Affine<F> &a) { return a.is_inf() ? inf() : XYZZ{a.x, a.y, F::one(), F::one()}; }
    ZK_HD bool is_inf() const { return zz.is_zero(); }
    ZK_HD XYZZ neg() const { return {x, y.neg(), zz, zzz}; }
    ZK_HD XYZZ normalized() const { return {x.normalized(), y.normalized(), zz.normalized(), zzz.normalized()}; }

    // dbl-2008-s-1
    ZK_HD XYZZ dbl_inl() const {
        if (is_inf()) return *this;
        F U = y.dbl(), V = U.sqr(), W = U * V, S = x * V;
        F xx = x.sqr(), M = xx.dbl() + xx;
        F X3 = M.sqr() - S.dbl();
        F Y3 = M * (S - X3) - W * y;
        return {X3, Y3, V * zz, W * zzz};
    }
    ZK_HD_NOINLINE XYZZ dbl_out() const { return dbl_inl(); }
    ZK_HD XYZZ dbl() const { if constexpr (kInlineAll) return dbl_inl(); else return dbl_out(); }

    // doubling of an affine point (mdbl-2008-s-1)
    static ZK_HD XYZZ dbl_affine_inl(const Affine<F> &a) {
        F U = a.y.dbl(), V = U.sqr(), W = U * V, S = a.x * V;
        F xx = a.x.sqr(), M = xx.dbl() + xx;
        F X3 = M.sqr() - S.dbl();
        F Y3 = M * (S - X3) - W * a.y;
        return {X3, Y3, V, W};
    }
    static ZK_HD_NOINLINE XYZZ dbl_affine_out(const Affine<F> &a) { return dbl_affine_inl(a); }

    // madd-2008-s, with the exceptional cases (this == inf, b == inf, b == +-this) handled.  The accumulator must stay in
    // registers across the hot loop, so the rare doubling branch works on private copies: nothing that lives across
    // iterations ever has its address taken (an escaped address parks the whole accumulator in scratch memory —
    // measured as 2 GB of extra HBM writes per 2^20-point MSM).
    ZK_HD void madd(const Affine<F> &b) {
        if (b.is_inf()) return;
        if (is_inf()) { x = b.x; y = b.y; zz = F::one(); zzz = F::one(); return; }
        F U2 = b.x * zz, S2 = b.y * zzz;
        F P = U2 - x, R = S2 - y;
        if (__builtin_expect(P.is_zero(), 0)) {
            if (R.is_zero()) {
                XYZZ t;
                if constexpr (kInlineAll) t = dbl_affine_inl(b);
                else { Affine<F> bc = {b.x, b.y}; t = dbl_affine_out(bc); }
                x = t.x; y = t.y; zz = t.zz; zzz = t.zzz;
            } else { x = F::zero(); y = F::one(); zz = F::zero(); zzz = F::zero(); }
            return;
        }
        F PP = P.sqr(), PPP = P * PP, Q = x * PP;
        F X3 = R.sqr() - PPP - Q.dbl();
        F Y3 = R * (Q - X3) - y * PPP;
        x = X3; y = Y3; zz = zz * PP; zzz = zzz * PPP;
    }
    // add-2008-s, exceptional cases handled
    ZK_HD void add_inl(const XYZZ &b) {
        if (b.is_inf()) return;
        if (is_inf()) { *this = b; return; }
        F U1 = x * b.zz, U2 = b.x * zz, S1 = y * b.zzz, S2 = b.y * zzz;
        F P = U2 - U1, R = S2 - S1;
        if (__builtin_expect(P.is_zero(), 0)) {
            if (R.is_zero()) { XYZZ t = dbl(); x = t.x; y = t.y; zz = t.zz; zzz = t.zzz; } else *this = inf();
            return;
        }
        F PP = P.sqr(), PPP = P * PP, Q = U1 * PP;
        F X3 = R.sqr() - PPP - Q.dbl();
        F Y3 = R * (Q - X3) - S1 * PPP;
        x = X3; y = Y3; zz = zz * b.zz * PP; zzz = zzz * b.zzz * PPP;
    }
    ZK_HD_NOINLINE void add_out(const XYZZ &b) { add_inl(b); }
    ZK_HD void add(const XYZZ &b) { if constexpr (kInlineAll) add_inl(b); else add_out(b); }

    ZK_HD_NOINLINE Affine<F> to_affine() const {
        if (is_inf()) return Affine<F>::inf();
        F zi = zzz.inverse();              // 1/ZZZ
        F t = zi * zz;                     // ZZ/ZZZ = 1/Z
        F zi2 = t.sqr();                   // 1/ZZ
        return {x * zi2, y * zi};
    }
    // k * this for a canonical little-endian u32 scalar (host: proof assembly; device: fixed-base tables)
    ZK_HD_NOINLINE XYZZ mul(const uint32_t *k, int nlimbs) const {
        XYZZ r = inf();
        for (int i = nlimbs * 32 - 1; i >= 0; --i) {
            r = r.dbl();
            if ((k[i >> 5] >> (i & 31)) & 1u) r.add(*this);
        }
        return r;
    }
};

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// ---- quad-cooperative addition (device) ---------------------------------------------------------------------------
// The tree / scan phases of the bucket reduction are dependency chains of additions with most lanes idle.  Here the four
// lanes of a DPP quad hold IDENTICAL copies of both operands and share one addition: each lane multiplies a different pair
// of operands (selected with v_cndmask, so the multiplication itself is one converged instruction stream) and the products
// are broadcast back with v_mov_b32_dpp quad_perm.  14 dependent multiplications become 4 rounds.
template <int S> ZK_D uint32_t quad_bcast_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, S * 0x55, 0xf, 0xf, true); }
template <int S> ZK_D Fq quad_bcast(const Fq &a) { Fq r; for (int i = 0; i < 8; ++i) r.v[i] = quad_bcast_u32<S>(a.v[i]); return r; }
template <int S> ZK_D Fq2 quad_bcast(const Fq2 &a) { return {quad_bcast<S>(a.c0), quad_bcast<S>(a.c1)}; }
ZK_D Fq quad_select(uint32_t q, const Fq &a0, const Fq &a1, const Fq &a2, const Fq &a3) {
    Fq r;
    for (int i = 0; i < 8; ++i) { uint32_t lo = q & 1 ? a1.v[i] : a0.v[i], hi = q & 1 ? a3.v[i] : a2.v[i]; r.v[i] = q & 2 ? hi : lo; }
    return r;
}
ZK_D Fq2 quad_select(uint32_t q, const Fq2 &a0, const Fq2 &a1, const Fq2 &a2, const Fq2 &a3) {
    return {quad_select(q, a0.c0, a1.c0, a2.c0, a3.c0), quad_select(q, a0.c1, a1.c1, a2.c1, a3.c1)};
}
// a += b; every lane of the quad passes the same a, b and receives the same sum.  q = lane & 3.
template <class F> ZK_D void xyzz_add_quad(XYZZ<F> &a, const XYZZ<F> &b, uint32_t q) {
    if (b.is_inf()) return;                                   // quad-uniform branches: all four lanes see the same data
    if (a.is_inf()) { a = b; return; }
    F m1 = quad_select(q, a.x, b.x, a.y, b.y) * quad_select(q, b.zz, a.zz, b.zzz, a.zzz);
    F U1 = quad_bcast<0>(m1), U2 = quad_bcast<1>(m1), S1 = quad_bcast<2>(m1), S2 = quad_bcast<3>(m1);
    F P = U2 - U1, R = S2 - S1;
    if (__builtin_expect(P.is_zero(), 0)) {
        if (R.is_zero()) { XYZZ<F> t = a.dbl(); a = t; } else a = XYZZ<F>::inf();
        return;
    }
    F m2 = quad_select(q, P, R, a.zz, a.zzz) * quad_select(q, P, R, b.zz, b.zzz);
    F PP = quad_bcast<0>(m2), RR = quad_bcast<1>(m2), ZZ12 = quad_bcast<2>(m2), ZZZ12 = quad_bcast<3>(m2);
    F m3 = quad_select(q, P, U1, ZZ12, ZZ12) * PP;           // lane 3 repeats lane 2's product
    F PPP = quad_bcast<0>(m3), Q = quad_bcast<1>(m3), ZZ3 = quad_bcast<2>(m3);
    F X3 = RR - PPP - Q.dbl();
    F m4 = quad_select(q, R, S1, ZZZ12, ZZZ12) * quad_select(q, Q - X3, PPP, PPP, PPP);
    F Y3 = quad_bcast<0>(m4) - quad_bcast<1>(m4);
    a.x = X3; a.y = Y3; a.zz = ZZ3; a.zzz = quad_bcast<2>(m4);
}
#endif

typedef Affine<Fq> G1Affine;  typedef XYZZ<Fq> G1;
typedef Affine<Fq2> G2Affine; typedef XYZZ<Fq2> G2;

}  // namespace zk
