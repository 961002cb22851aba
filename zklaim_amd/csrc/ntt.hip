// ntt.hip — radix-2 NTT over alt_bn128 Fr for gfx950.
//
// Replaces libfqfft basic_radix2_domain<Fr>::FFT / iFFT / cosetFFT / icosetFFT
// (_basic_serial_radix2_FFT, _multiply_by_coset), reached from r1cs_to_qap_witness_map inside
// r1cs_gg_ppzksnark_prover (/root/reference/zklaim/snark.cpp:126).
//
// Shape: the log2(N) butterfly stages are cut into passes of R <= 10 stages (R <= 8 below 2^18).  One workgroup
// stages a tile of 2^R rows x CW columns (1024 elements = 40 KiB of 29-bit records, 512 below 2^18: several tiles share a CU) in LDS, runs
// its R stages there and writes the tile back, so a 2^20 transform touches HBM twice instead of 20 times.  The
// bit-reversal permutation is folded into the first pass's gather (reads stay CW*32 B contiguous),
// coset / 1/N scalings are folded into the first load / last store.  Twiddles come from a
// per-domain table omega^i (i < N/2) that stays L2/MALL resident.  Workgroups are dealt round-robin over the 8 XCDs, so tile t is given
// to workgroup 8 (t mod tiles/8) + t / (tiles/8): the tiles one XCD's L2 sees are neighbours in memory.
// MFMA is not used: the work is 254-bit modular multiplication on the integer VALU.
#include "common.hpp"
#include "fr29.hip.hpp"
#include "../../include/zkg.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

namespace zk {

static constexpr int NTT_THREADS = 512;
static const int NTT_TILE_LOG_FORCE = getenv("ZKG_NTT_TILE_LOG") ? atoi(getenv("ZKG_NTT_TILE_LOG")) : 0;   // tuning aid
static const int NTT_MAX_R_FORCE = getenv("ZKG_NTT_MAX_R") ? atoi(getenv("ZKG_NTT_MAX_R")) : 0;                          // stages per pass (tuning aid)

struct alignas(16) U4 { uint32_t a, b, c, d; };

struct NttPassArgs {
    const Fr *src; Fr *dst; const Fr *tw; const Fr *pre; const Fr *post;
    Fr post_scalar;
    uint32_t n_log, s0, R, cw_log, first, has_post_scalar;
    size_t src_batch_stride, dst_batch_stride;      // elements between the vectors of a batch (blockIdx.y)
};

ZK_D uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(NttPassArgs A) {
    extern __shared__ U4 smem[];
    const uint32_t rows = 1u << A.R, CW = 1u << A.cw_log, stride = 2 * CW + 1;   // +1 x 16 B pad per row
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, nthr = blockDim.x;         // one thread per butterfly of a stage
    A.src += (size_t)blockIdx.y * A.src_batch_stride; A.dst += (size_t)blockIdx.y * A.dst_batch_stride;
    const uint32_t s1 = A.s0 + A.R;
    const uint32_t lo_mask = (1u << A.s0) - 1;
    auto lds_ld = [&](uint32_t row, uint32_t c) { Fr r; const U4 *p = &smem[row * stride + 2 * c]; *reinterpret_cast<U4 *>(&r.v[0]) = p[0]; *reinterpret_cast<U4 *>(&r.v[4]) = p[1]; return r; };
    auto lds_st = [&](uint32_t row, uint32_t c, const Fr &r) { U4 *p = &smem[row * stride + 2 * c]; p[0] = *reinterpret_cast<const U4 *>(&r.v[0]); p[1] = *reinterpret_cast<const U4 *>(&r.v[4]); };

    // ---- load tile (c fastest: CW*32 B contiguous per row)
    for (uint32_t e = tid; e < rows * CW; e += nthr) {
        uint32_t c = e & (CW - 1), mid = e >> A.cw_log;
        size_t idx; uint32_t row;
        if (A.first) { idx = (size_t)mid * ((size_t)1 << (A.n_log - A.R)) + (size_t)tile * CW + c; row = bitrev(mid, A.R); }
        else { uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); row = mid; }
        Fr v = A.src[idx];
        if (A.pre) v = v * A.pre[idx];
        lds_st(row, c, v);
    }
    __syncthreads();

    // ---- R butterfly stages in LDS
    for (uint32_t q = 0; q < A.R; ++q) {
        const uint32_t s = A.s0 + q, half = 1u << q;
        for (uint32_t bf = tid; bf < (rows >> 1) * CW; bf += nthr) {
            uint32_t c = bf & (CW - 1), k = bf >> A.cw_log;
            uint32_t j = k & (half - 1), r0 = ((k >> q) << (q + 1)) | j, r1 = r0 + half;
            Fr u = lds_ld(r0, c), v = lds_ld(r1, c);
            if (s != 0) {
                uint32_t lo = A.first ? 0u : ((tile * CW + c) & lo_mask);
                size_t e = ((size_t)((j << A.s0) + lo)) << (A.n_log - 1 - s);
                v = v * A.tw[e];
            }
            lds_st(r0, c, u + v);
            lds_st(r1, c, u - v);
        }
        __syncthreads();
    }

    // ---- store tile
    for (uint32_t e = tid; e < rows * CW; e += nthr) {
        uint32_t c, mid; size_t idx;
        if (A.first) { mid = e & (rows - 1); c = e >> A.R; idx = ((size_t)bitrev(tile * CW + c, A.n_log - A.R) << A.R) + mid; }
        else { c = e & (CW - 1); mid = e >> A.cw_log; uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); }
        Fr v = lds_ld(mid, c);
        if (A.post) v = v * A.post[idx];
        else if (A.has_post_scalar) v = v * A.post_scalar;
        A.dst[idx] = v.normalized();
    }
}

// ---- the same pass on the 29-bit representation (fr29.hip.hpp): 40-byte records in LDS and between passes, libff's form at the
//      transform's two ends only.  Tables (twiddles, pre / post scalings) are 29-bit records made once per domain (k_table29).
struct NttPassArgs29 {
    const void *src; void *dst; const Rec29 *tw, *pre, *post;
    uint32_t post_scalar[9];
    uint32_t n_log, s0, R, cw_log, first, last, has_post_scalar, row_pad;      // row_pad: records of padding per LDS row
    uint32_t xcd_tiles;                                                        // tiles per XCD (0 = identity map)
    uint32_t critical;                                                         // raise the wavefronts' issue priority (crit_wave_priority)
    uint32_t norm_stores;                                                      // radix-4 steps: carry propagation on the four stores (round 3) instead of on the two added-to loads
    size_t src_batch_stride, dst_batch_stride;      // elements between the vectors of a batch (blockIdx.y), in the units of src / dst
};
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass29(NttPassArgs29 A) {
    crit_wave_priority((int)A.critical);
    extern __shared__ U4 smem[];
    Rec29 *lds = reinterpret_cast<Rec29 *>(smem);
    const uint32_t rows = 1u << A.R, CW = 1u << A.cw_log, stride = CW + A.row_pad;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    // workgroups are dealt round-robin over the 8 XCDs: with xcd_tiles set, the tiles one XCD works on are neighbours in memory
    const uint32_t tile = A.xcd_tiles ? (blockIdx.x & 7u) * A.xcd_tiles + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t s1 = A.s0 + A.R;
    const uint32_t lo_mask = (1u << A.s0) - 1;
    const Fr *src_abi = reinterpret_cast<const Fr *>(A.src) + (size_t)blockIdx.y * A.src_batch_stride;
    const Rec29 *src_rec = reinterpret_cast<const Rec29 *>(A.src) + (size_t)blockIdx.y * A.src_batch_stride;
    Fr *dst_abi = reinterpret_cast<Fr *>(A.dst) + (size_t)blockIdx.y * A.dst_batch_stride;
    Rec29 *dst_rec = reinterpret_cast<Rec29 *>(A.dst) + (size_t)blockIdx.y * A.dst_batch_stride;

    // ---- load tile
    for (uint32_t e = tid; e < rows * CW; e += nthr) {
        uint32_t c = e & (CW - 1), mid = e >> A.cw_log;
        size_t idx; uint32_t row;
        if (A.first) { idx = (size_t)mid * ((size_t)1 << (A.n_log - A.R)) + (size_t)tile * CW + c; row = bitrev(mid, A.R); }
        else { uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); row = mid; }
        Fr29 v = A.first ? fr29::slice(src_abi[idx]) : fr29::load_rec(src_rec + idx);
        if (A.pre) v = fr29::mul(v, fr29::load_rec(A.pre + idx));
        fr29::store_rec(lds + row * stride + c, v);
    }
    __syncthreads();

    // ---- R butterfly stages in LDS
    for (uint32_t q = 0; q < A.R; ++q) {
        const uint32_t s = A.s0 + q, half = 1u << q;
        for (uint32_t bf = tid; bf < (rows >> 1) * CW; bf += nthr) {
            uint32_t c = bf & (CW - 1), k = bf >> A.cw_log;
            uint32_t j = k & (half - 1), r0 = ((k >> q) << (q + 1)) | j, r1 = r0 + half;
            const Fr29 u = fr29::load_rec(lds + r0 * stride + c);
            Fr29 v = fr29::load_rec(lds + r1 * stride + c);
            if (s != 0) {
                uint32_t lo = A.first ? 0u : ((tile * CW + c) & lo_mask);
                size_t e = ((size_t)((j << A.s0) + lo)) << (A.n_log - 1 - s);
                v = fr29::mul(v, fr29::load_rec(A.tw + e));
            }
            fr29::store_rec(lds + r0 * stride + c, fr29::add_norm(u, v));
            fr29::store_rec(lds + r1 * stride + c, fr29::sub_norm(u, v));
        }
        __syncthreads();
    }

    // ---- store tile
    for (uint32_t e = tid; e < rows * CW; e += nthr) {
        uint32_t c, mid; size_t idx;
        if (A.first) { mid = e & (rows - 1); c = e >> A.R; idx = ((size_t)bitrev(tile * CW + c, A.n_log - A.R) << A.R) + mid; }
        else { c = e & (CW - 1); mid = e >> A.cw_log; uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); }
        Fr29 v = fr29::load_rec(lds + mid * stride + c);
        if (!A.last) { fr29::store_rec(dst_rec + idx, v); continue; }
        Fr29 f;                                                      // the last product: post table, 1/N, or one — below 2r either way
        if (A.post) f = fr29::load_rec(A.post + idx);
        else if (A.has_post_scalar) { for (int i = 0; i < 9; ++i) f.v[i] = A.post_scalar[i]; }
        else { for (int i = 0; i < 9; ++i) f.v[i] = fr29::ONE[i]; }
        dst_abi[idx] = fr29::unslice_reduce(fr29::mul(v, f));
    }
}
// ---- the same pass with radix-4 steps: a thread takes four rows r, r + h, r + 2h, r + 3h of one column through stages q and q + 1 in
//      registers (four products as two interleaved pairs, three twiddles), so a tile makes R / 2 round trips through LDS instead of R, with
//      half the barriers, and the additions / subtractions between the two stages are limb-wise: one carry propagation per element per
//      step instead of one per stage.  Limb bounds: stage q: x + t < 2^30, x + 2r - t < 1.5 x 2^30 (these feed stage q + 1's products: the
//      stream takes limbs up to 2.5 x 2^30, tools/gen_mont_asm.py selftest_f29); stage q + 1: below 2.5 x 2^30.  An odd R ends with one radix-2 step.
__global__ __launch_bounds__(256) void k_ntt_pass29_r4(NttPassArgs29 A) {
    crit_wave_priority((int)A.critical);
    extern __shared__ U4 smem[];
    Rec29 *lds = reinterpret_cast<Rec29 *>(smem);
    const uint32_t rows = 1u << A.R, CW = 1u << A.cw_log, stride = CW + A.row_pad;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t tile = A.xcd_tiles ? (blockIdx.x & 7u) * A.xcd_tiles + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t s1 = A.s0 + A.R;
    const uint32_t lo_mask = (1u << A.s0) - 1;
    const Fr *src_abi = reinterpret_cast<const Fr *>(A.src) + (size_t)blockIdx.y * A.src_batch_stride;
    const Rec29 *src_rec = reinterpret_cast<const Rec29 *>(A.src) + (size_t)blockIdx.y * A.src_batch_stride;
    Fr *dst_abi = reinterpret_cast<Fr *>(A.dst) + (size_t)blockIdx.y * A.dst_batch_stride;
    Rec29 *dst_rec = reinterpret_cast<Rec29 *>(A.dst) + (size_t)blockIdx.y * A.dst_batch_stride;

    for (uint32_t e = tid; e < rows * CW; e += nthr) {               // load tile
        uint32_t c = e & (CW - 1), mid = e >> A.cw_log;
        size_t idx; uint32_t row;
        if (A.first) { idx = (size_t)mid * ((size_t)1 << (A.n_log - A.R)) + (size_t)tile * CW + c; row = bitrev(mid, A.R); }
        else { uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); row = mid; }
        Fr29 v = A.first ? fr29::slice(src_abi[idx]) : fr29::load_rec(src_rec + idx);
        if (A.pre) v = fr29::mul(v, fr29::load_rec(A.pre + idx));
        fr29::store_rec(lds + row * stride + c, v);
    }
    __syncthreads();

    uint32_t q = 0;
    for (; q + 1 < A.R; q += 2) {                                     // radix-4 steps: stages s0 + q and s0 + q + 1
        const uint32_t s = A.s0 + q, h = 1u << q;
        for (uint32_t qd = tid; qd < (rows >> 2) * CW; qd += nthr) {
            const uint32_t c = qd & (CW - 1), k = qd >> A.cw_log;
            const uint32_t j = k & (h - 1), r0 = ((k >> q) << (q + 2)) | j;
            Rec29 *p0 = lds + r0 * stride + c, *p1 = p0 + h * stride, *p2 = p1 + h * stride, *p3 = p2 + h * stride;
            const uint32_t lo = A.first ? 0u : ((tile * CW + c) & lo_mask);
            const size_t eb = ((size_t)((j << A.s0) + lo)) << (A.n_log - 2 - s);              // stage s + 1, row bits j
            const Fr29 wb = fr29::load_rec(A.tw + eb), wc = fr29::load_rec(A.tw + eb + ((size_t)1 << (A.n_log - 2)));   // ... and row bits j + h: (h << s0) << (n - 2 - s) = N / 4 further
            // Records in LDS and between passes hold LAZY limbs (below 2.5 x 2^30, see the bounds above): the two rows that only get added to
            // (x0, x2) are carried here, on load; the two that get multiplied (x1, x3) go into the product as they are — two carry
            // propagations per butterfly instead of four on the stores (ZKG_NTT_NORM_STORES=1: the old placement, for A/B).
            Fr29 x0 = fr29::load_rec(p0), x2 = fr29::load_rec(p2);
            if (!A.norm_stores) { x0 = fr29::norm(x0); x2 = fr29::norm(x2); }
            Fr29 t1 = fr29::load_rec(p1), t3 = fr29::load_rec(p3);
            if (s != 0) { const Fr29 wa = fr29::load_rec(A.tw + 2 * eb); fr29::mul2(t1, t3, t1, wa, t3, wa); }
            else if (!A.norm_stores) { t1 = fr29::norm(t1); t3 = fr29::norm(t3); }      // (stage 0 has no product to bring them back to digits)
            const Fr29 y0 = fr29::add_lazy(x0, t1), y1 = fr29::sub_lazy(x0, t1), y2 = fr29::add_lazy(x2, t3), y3 = fr29::sub_lazy(x2, t3);
            Fr29 u2, u3;
            fr29::mul2(u2, u3, y2, wb, y3, wc);
            if (A.norm_stores) {
                fr29::store_rec(p0, fr29::norm(fr29::add_lazy(y0, u2)));
                fr29::store_rec(p2, fr29::norm(fr29::sub_lazy(y0, u2)));
                fr29::store_rec(p1, fr29::norm(fr29::add_lazy(y1, u3)));
                fr29::store_rec(p3, fr29::norm(fr29::sub_lazy(y1, u3)));
            } else {
                fr29::store_rec(p0, fr29::add_lazy(y0, u2));
                fr29::store_rec(p2, fr29::sub_lazy(y0, u2));
                fr29::store_rec(p1, fr29::add_lazy(y1, u3));
                fr29::store_rec(p3, fr29::sub_lazy(y1, u3));
            }
        }
        __syncthreads();
    }
    if (q < A.R) {                                                    // odd R: the last stage alone
        const uint32_t s = A.s0 + q, half = 1u << q;
        for (uint32_t bf = tid; bf < (rows >> 1) * CW; bf += nthr) {
            uint32_t c = bf & (CW - 1), k = bf >> A.cw_log;
            uint32_t j = k & (half - 1), r0 = ((k >> q) << (q + 1)) | j, r1 = r0 + half;
            const Fr29 u = fr29::norm(fr29::load_rec(lds + r0 * stride + c));                 // (lazy limbs in: see the radix-4 steps)
            Fr29 v = fr29::load_rec(lds + r1 * stride + c);
            if (s != 0) {
                uint32_t lo = A.first ? 0u : ((tile * CW + c) & lo_mask);
                size_t e = ((size_t)((j << A.s0) + lo)) << (A.n_log - 1 - s);
                v = fr29::mul(v, fr29::load_rec(A.tw + e));
            } else v = fr29::norm(v);
            fr29::store_rec(lds + r0 * stride + c, fr29::add_norm(u, v));
            fr29::store_rec(lds + r1 * stride + c, fr29::sub_norm(u, v));
        }
        __syncthreads();
    }

    for (uint32_t e = tid; e < rows * CW; e += nthr) {               // store tile
        uint32_t c, mid; size_t idx;
        if (A.first) { mid = e & (rows - 1); c = e >> A.R; idx = ((size_t)bitrev(tile * CW + c, A.n_log - A.R) << A.R) + mid; }
        else { c = e & (CW - 1); mid = e >> A.cw_log; uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); }
        Fr29 v = fr29::load_rec(lds + mid * stride + c);
        if (!A.last) { fr29::store_rec(dst_rec + idx, v); continue; }
        Fr29 f;
        if (A.post) f = fr29::load_rec(A.post + idx);
        else if (A.has_post_scalar) { for (int i = 0; i < 9; ++i) f.v[i] = A.post_scalar[i]; }
        else { for (int i = 0; i < 9; ++i) f.v[i] = fr29::ONE[i]; }
        dst_abi[idx] = fr29::unslice_reduce(fr29::mul(v, f));
    }
}
// a table of Fr elements in libff's form -> 29-bit records of x R' (x 2^256 times 32 = x 2^261, sliced)
__global__ __launch_bounds__(256) void k_table29(const Fr *in, size_t n, Fr c32, Rec29 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr29::store_rec(out + i, fr29::slice((in[i] * c32).normalized()));
}
int ntt_table29(DevBuf &out, const Fr *d_in, size_t n, hipStream_t s) {
    if (out.reserve(std::max<size_t>(1, n) * sizeof(Rec29))) return ZKG_ERROR;
    if (n) hipLaunchKernelGGL(k_table29, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, n, Fr::from_u64(32), out.as<Rec29>());
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}

// out[i] = scale * base^i : each thread seeds base^(64 t) by square-and-multiply, then walks 64 entries
__global__ void k_powers(Fr *out, size_t n, Fr base, Fr scale) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = t * 64;
    if (i0 >= n) return;
    Fr cur = scale * base.pow_u64(i0);
    for (size_t i = i0; i < i0 + 64 && i < n; ++i) { out[i] = cur.normalized(); cur = cur * base; }
}

int powers_table(Fr *d_out, size_t n, const Fr &base, const Fr &scale, hipStream_t s) {
    size_t threads = (n + 63) / 64;
    hipLaunchKernelGGL(k_powers, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_out, n, base, scale);
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}

// Fr::root_of_unity (2^28-th primitive root) = 5^((r-1)/2^28), Montgomery form
static Fr fr_root_of_unity() {
    Fr r; const uint32_t l[8] = {0x80d13d9cu, 0x636e7355u, 0x2445ffd6u, 0xa22bf374u, 0x1eb203d8u, 0x56452ac0u, 0x2963f9e7u, 0x1860ef94u};
    for (int i = 0; i < 8; ++i) r.v[i] = l[i];
    return r;
}

Fr fr_root_of_unity_pow2(unsigned logn) {                               // libff::get_root_of_unity(2^logn)
    Fr omega = fr_root_of_unity();
    for (unsigned i = 28; i > logn; --i) omega = omega.sqr();
    return omega;
}

int NttDomain::init(unsigned logn_, hipStream_t s) {
    logn = logn_;
    size_t N = (size_t)1 << logn;
    Fr omega = fr_root_of_unity_pow2(logn);
    Fr g = Fr::from_u64(5);                                            // Fr::multiplicative_generator
    n_inv = Fr::from_u64(N).inverse();
    size_t half = N > 1 ? N / 2 : 1;
    if (tw_fwd.reserve(half * sizeof(Fr)) || tw_inv.reserve(half * sizeof(Fr)) || coset_pre.reserve(N * sizeof(Fr)) ||
        icoset_post.reserve(N * sizeof(Fr)) || scratch.reserve(N * sizeof(Rec29))) return ZKG_ERROR;
    if (powers_table(tw_fwd.as<Fr>(), half, omega, Fr::one(), s)) return ZKG_ERROR;
    if (powers_table(tw_inv.as<Fr>(), half, omega.inverse(), Fr::one(), s)) return ZKG_ERROR;
    if (powers_table(coset_pre.as<Fr>(), N, g, Fr::one(), s)) return ZKG_ERROR;
    if (powers_table(icoset_post.as<Fr>(), N, g.inverse(), n_inv, s)) return ZKG_ERROR;
    if (ntt_table29(tw_fwd29, tw_fwd.as<Fr>(), half, s) || ntt_table29(tw_inv29, tw_inv.as<Fr>(), half, s) ||
        ntt_table29(coset_pre29, coset_pre.as<Fr>(), N, s) || ntt_table29(icoset_post29, icoset_post.as<Fr>(), N, s)) return ZKG_ERROR;
    return ZKG_OK;
}
void NttDomain::release() {
    tw_fwd.release(); tw_inv.release(); coset_pre.release(); icoset_post.release(); scratch.release();
    tw_fwd29.release(); tw_inv29.release(); coset_pre29.release(); icoset_post29.release();
    for (auto &kv : stream_scratch) kv.second.release();
    stream_scratch.clear();
}
// transforms of one size on different streams must not share the inter-pass scratch vector
Fr *NttDomain::scratch_for(hipStream_t s) {
    std::lock_guard<std::mutex> lk(mu);
    if (first_stream_set && s == first_stream) return scratch.as<Fr>();
    if (!first_stream_set) { first_stream_set = true; first_stream = s; return scratch.as<Fr>(); }
    DevBuf &b = stream_scratch[s];
    if (b.reserve(((size_t)1 << logn) * sizeof(Rec29))) return nullptr;
    return b.as<Fr>();
}

static std::mutex g_dom_mu;
static std::map<unsigned, NttDomain *> g_domains;
static std::map<size_t, StepDomain *> g_step_domains;
NttDomain *ntt_domain(unsigned logn, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_dom_mu);
    auto it = g_domains.find(logn);
    if (it != g_domains.end()) return it->second;
    NttDomain *d = new NttDomain();
    if (d->init(logn, s) != ZKG_OK) { d->release(); delete d; return nullptr; }
    g_domains[logn] = d;
    return d;
}
int ntt_configure() {
    return hipFuncSetAttribute((const void *)k_ntt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void *)k_ntt_pass29, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void *)k_ntt_pass29_r4, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess ? ZKG_OK : ZKG_ERROR;
}
void ntt_release_all() {
    std::lock_guard<std::mutex> lk(g_dom_mu);
    for (auto &kv : g_domains) { kv.second->release(); delete kv.second; }
    g_domains.clear();
    for (auto &kv : g_step_domains) { kv.second->release(); delete kv.second; }
    g_step_domains.clear();
}

int ntt_run_ex(NttDomain *d, Fr *d_a, bool inverse, const Fr *pre, const Fr *post, const Fr *post_scalar, hipStream_t s, Fr *scratch, unsigned batch, size_t batch_stride,
               const void *pre29, const void *post29) {
    const unsigned n = d->logn;
    if (n == 0) return ZKG_OK;                 // N = 1: every variant is the identity (g^0 = 1, 1/N = 1)
    if (batch > 1 && !scratch) { set_error("ntt: a batched transform needs its own scratch"); return ZKG_ERROR; }
    const size_t N = (size_t)1 << n;
    // the domain's own scaling tables have 29-bit twins; a caller's table needs its twin passed along (ntt_table29), else the 32-bit pass runs
    if (pre && !pre29 && pre == d->coset_pre.as<Fr>()) pre29 = d->coset_pre29.p;
    if (post && !post29 && post == d->icoset_post.as<Fr>()) post29 = d->icoset_post29.p;
    static const bool ntt32 = getenv("ZKG_NTT_32") != nullptr;                                          // A/B switch
    const bool use29 = !ntt32 && (!pre || pre29) && (!post || post29);
    // geometry, measured (MI355X, profiles/r3_ntt_geometry.txt): passes of up to 10 stages over 1024-element tiles (40 KiB of LDS records, four
    // workgroups per CU), radix-4 steps, XCD-contiguous tile map — 2^20 in two passes: 0.131 ms against 0.145 ms for round 2's three passes
    // of 7 + 7 + 6 radix-2 stages over 512-element tiles (ZKG_NTT_MAX_R=8 ZKG_NTT_TILE_LOG=9 ZKG_NTT_RADIX2=1 ZKG_NTT_XCD=0)
    // Below 2^18 there are too few 1024-element tiles for the chip (2^16: 0.029 ms with 512-element tiles and a thread per butterfly, 0.035 ms
    // with the large geometry): those keep round 2's shape.
    const bool large = n >= 18;
    const int NTT_TILE_LOG = NTT_TILE_LOG_FORCE ? NTT_TILE_LOG_FORCE : (large ? 10 : 9);
    const unsigned NTT_MAX_R = NTT_MAX_R_FORCE ? (unsigned)NTT_MAX_R_FORCE : (large ? 10u : 8u);
    unsigned npass = (n + NTT_MAX_R - 1) / NTT_MAX_R;
    if (n <= (unsigned)NTT_TILE_LOG) npass = 1;
    unsigned base = n / npass, extra = n % npass;
    Fr *tmp = scratch ? scratch : d->scratch_for(s);
    if (!tmp) return ZKG_ERROR;
    unsigned s0 = 0;
    for (unsigned p = 0; p < npass; ++p) {
        unsigned R = base + (p < extra ? 1 : 0);
        NttPassArgs A;
        A.first = (p == 0);
        A.src = (p == 0) ? d_a : tmp;
        A.dst = (p == npass - 1) ? d_a : tmp;
        A.tw = (inverse ? d->tw_inv : d->tw_fwd).as<Fr>();
        A.pre = (p == 0) ? pre : nullptr;
        A.post = (p == npass - 1) ? post : nullptr;
        A.has_post_scalar = (p == npass - 1 && post_scalar && !post) ? 1 : 0;
        A.post_scalar = A.has_post_scalar ? *post_scalar : Fr::zero();
        A.n_log = n; A.s0 = s0; A.R = R;
        A.src_batch_stride = A.dst_batch_stride = batch_stride ? batch_stride : N;     // vectors and their scratch share one layout
        unsigned cols_log = n - R;                                   // columns in total
        const unsigned room = (unsigned)NTT_TILE_LOG > R ? (unsigned)NTT_TILE_LOG - R : 0u;   // (a forced tile smaller than 2^R rows: one column per tile)
        A.cw_log = cols_log < room ? cols_log : room;
        if (npass == 1) A.cw_log = 0;
        size_t tiles = (N >> R) >> A.cw_log;
        size_t rows = (size_t)1 << R, CW = (size_t)1 << A.cw_log;
        size_t lds = rows * (2 * CW + 1) * 16;
        unsigned threads = (unsigned)std::min<size_t>(NTT_THREADS, std::max<size_t>(64, rows * CW / 2));
        if (use29) {
            NttPassArgs29 B;
            B.first = A.first; B.last = (p == npass - 1);
            B.src = A.src; B.dst = A.dst;                                                     // (tmp holds 40-byte records on this path)
            B.tw = (inverse ? d->tw_inv29 : d->tw_fwd29).as<Rec29>();
            B.pre = (p == 0) ? reinterpret_cast<const Rec29 *>(pre29) : nullptr;
            B.post = B.last ? reinterpret_cast<const Rec29 *>(post29) : nullptr;
            B.has_post_scalar = A.has_post_scalar;
            if (B.has_post_scalar) {                                                          // x 2^256 -> x 2^261, sliced (host)
                const Fr c = (*post_scalar * Fr::from_u64(32));
                for (int i = 0; i < 9; ++i) { const int bit = 29 * i, l = bit >> 5, sh = bit & 31; uint64_t w = c.v[l]; if (l + 1 < 8) w |= (uint64_t)c.v[l + 1] << 32; B.post_scalar[i] = (uint32_t)(w >> sh) & (i < 8 ? Fr29::M : 0xffffffffu); }
            }
            B.n_log = n; B.s0 = s0; B.R = R; B.cw_log = A.cw_log;
            B.src_batch_stride = A.src_batch_stride; B.dst_batch_stride = A.dst_batch_stride;
            // no row padding: a 512-element tile is 20 KiB of 40-byte records, and eight of them are exactly a CU's 160 KiB — the eight tiles per
            // CU of a 2^20 transform's pass stay one round of workgroups (with a padding record per row six fit: a second, thin round)
            static const uint32_t pad29 = getenv("ZKG_NTT29_PAD") ? (uint32_t)atoi(getenv("ZKG_NTT29_PAD")) : 0;              // tuning aid
            B.row_pad = pad29;
            static const bool xcd_map = !getenv("ZKG_NTT_XCD") || atoi(getenv("ZKG_NTT_XCD")) != 0;
            B.xcd_tiles = (xcd_map && tiles % 8 == 0) ? (uint32_t)(tiles / 8) : 0;
            // the transforms are the prover's critical path at 2^20 (beside the witness multi-exponentiations); below, the G2 witness chain ends
            // as late as the H query and raised priorities here cost more there (CRIT_PRIORITY_MIN_LOG)
            B.critical = crit_priority_for(2, n) ? 1u : 0u;
            static const bool norm_stores = getenv("ZKG_NTT_NORM_STORES") != nullptr;                                  // A/B switch
            B.norm_stores = norm_stores ? 1u : 0u;
            static const int radix_force = getenv("ZKG_NTT_RADIX2") ? (atoi(getenv("ZKG_NTT_RADIX2")) ? 2 : 4) : 0;          // A/B switch
            if (radix_force ? radix_force == 4 : large) {
                const unsigned threads4 = (unsigned)std::min<size_t>(256, std::max<size_t>(64, rows * CW / 4));             // one thread per four rows of a column
                hipLaunchKernelGGL(k_ntt_pass29_r4, dim3((unsigned)tiles, batch), dim3(threads4), rows * (CW + pad29) * sizeof(Rec29), s, B);
            } else
            hipLaunchKernelGGL(k_ntt_pass29, dim3((unsigned)tiles, batch), dim3(threads), rows * (CW + pad29) * sizeof(Rec29), s, B);
        } else
        hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)tiles, batch), dim3(threads), lds, s, A);
        if (hipGetLastError() != hipSuccess) { set_error("ntt pass launch failed"); return ZKG_ERROR; }
        s0 += R;
    }
    return ZKG_OK;
}

int ntt_run(NttDomain *d, Fr *d_a, int inverse, int coset, hipStream_t s) {
    if (!inverse) return ntt_run_ex(d, d_a, false, coset ? d->coset_pre.as<Fr>() : nullptr, nullptr, nullptr, s);
    if (coset) return ntt_run_ex(d, d_a, true, nullptr, d->icoset_post.as<Fr>(), nullptr, s);
    return ntt_run_ex(d, d_a, true, nullptr, nullptr, &d->n_inv, s);
}


// ======================================================================================================
// step_radix2_domain (libfqfft step_radix2_domain<Fr>): m = big + small.  The domain is the big-th roots of unity
// followed by the coset omega * (small-th roots of unity), omega of order 2 big.  Both transforms are one pass of
// elementwise work around a radix-2 transform of size big and one of size small; lane i owns every index congruent
// to i mod small, so the fold / unfold kernels work in place without atomics.
// ======================================================================================================
static unsigned ceil_log2(size_t x) { unsigned r = 0; while (((size_t)1 << r) < x) ++r; return r; }

bool evaluation_domain_shape(size_t min_size, DomainShape &d) {
    if (min_size <= 1) return false;
    unsigned lg = ceil_log2(min_size);
    if (lg > 28) return false;
    d = DomainShape();
    d.log_m = lg;
    if (min_size == ((size_t)1 << lg)) { d.m = min_size; return true; }                    // basic_radix2_domain(min_size)
    size_t big = (size_t)1 << (lg - 1), small = min_size - big, rounded_small = (size_t)1 << ceil_log2(small);
    if (small != rounded_small && big == rounded_small) { d.m = big + rounded_small; return true; }   // basic_radix2_domain(big + rounded_small)
    d.step = true; d.big = big; d.small = rounded_small; d.m = big + rounded_small;         // step_radix2_domain(min_size | big + rounded_small)
    return true;
}
bool domain_shape_of(size_t m, DomainShape &d) { return evaluation_domain_shape(m, d) && d.m == m; }

// evaluate_all_lagrange_polynomials(t), compute_vanishing_polynomial(t).  Closed forms with ONE batched inversion:
//   basic:  u_i = Z(t) w^i / (m (t - w^i)),                                   Z(t) = t^m - 1
//   step:   u_i = Zb(t) wb^i (t^small - omega^small) / (big (t - wb^i)(wb^(i small) - omega^small))   for i < big
//           u_(big+i) = (t^big - 1)/(omega^big - 1) * Zs(t') ws^i / (small (t' - ws^i)),  t' = t / omega
static int batch_invert_range(Fr *den, size_t n) {
    std::vector<Fr> pre(n);
    Fr run = Fr::one();
    for (size_t i = 0; i < n; ++i) { if (den[i].is_zero()) return ZKG_ERROR; pre[i] = run; run = run * den[i]; }
    Fr inv = run.inverse();
    for (size_t i = n; i-- > 0;) { Fr di = inv * pre[i]; inv = inv * den[i]; den[i] = di; }
    return ZKG_OK;
}
// Montgomery's trick, one inversion per chunk, chunks on the host pool
static int batch_invert(std::vector<Fr> &den) {
    const size_t n = den.size();
    const int chunks = (int)std::min<size_t>(64, (n + 8191) / 8192);
    if (chunks <= 1) return batch_invert_range(den.data(), n);
    std::vector<int> rc(chunks, 0);
    host_parallel_for(chunks, [&](int c) { size_t lo = n * (size_t)c / chunks, hi = n * (size_t)(c + 1) / chunks; rc[c] = batch_invert_range(den.data() + lo, hi - lo); });
    for (int r : rc) if (r) return ZKG_ERROR;
    return ZKG_OK;
}
// f(lo, hi) over [0, n) in chunks on the host pool (a key generator's loops over the domain: independent per index once a chunk has
// recomputed its starting power)
template <class Fn> static void chunked(size_t n, Fn f) {
    const int chunks = (int)std::min<size_t>(64, (n + 8191) / 8192);
    if (chunks <= 1) { f((size_t)0, n); return; }
    host_parallel_for(chunks, [&](int c) { f(n * (size_t)c / chunks, n * (size_t)(c + 1) / chunks); });
}
int domain_lagrange(const DomainShape &d, const Fr &t, std::vector<Fr> &u, Fr &Zt) {
    const size_t m = d.m;
    u.assign(m, Fr::zero());
    std::vector<Fr> den(m);
    if (!d.step) {
        Fr omega = fr_root_of_unity_pow2(d.log_m), wi = Fr::one(), mf = Fr::from_u64(m);
        Zt = t.pow_u64(m) - Fr::one();
        (void)wi;
        chunked(m, [&](size_t lo, size_t hi) {                                             // each chunk restarts the running power at omega^lo
            Fr w = omega.pow_u64(lo);
            for (size_t i = lo; i < hi; ++i) { u[i] = Zt * w; den[i] = mf * (t - w); w = w * omega; }
        });
    } else {
        const size_t big = d.big, small = d.small;
        Fr omega = fr_root_of_unity_pow2(d.log_m), wb = omega.sqr(), ws = fr_root_of_unity_pow2(ceil_log2(small));
        Fr omega_s = omega.pow_u64(small), Zb = t.pow_u64(big) - Fr::one(), L0 = t.pow_u64(small) - omega_s;
        Zt = Zb * L0;
        Fr wbs = wb.pow_u64(small), wi = Fr::one(), elt = Fr::one(), bf = Fr::from_u64(big), num = Zb * L0;
        (void)wi; (void)elt;
        chunked(big, [&](size_t lo, size_t hi) {
            Fr w = wb.pow_u64(lo), e = wbs.pow_u64(lo);
            for (size_t i = lo; i < hi; ++i) { u[i] = num * w; den[i] = bf * (t - w) * (e - omega_s); w = w * wb; e = e * wbs; }
        });
        Fr tp = t * omega.inverse(), Zs = tp.pow_u64(small) - Fr::one(), sf = Fr::from_u64(small);
        Fr L1num = Zb * Zs, L1den = omega.pow_u64(big) - Fr::one();
        const Fr L1s = L1den * sf;
        chunked(small, [&](size_t lo, size_t hi) {
            Fr w = ws.pow_u64(lo);
            for (size_t i = lo; i < hi; ++i) { u[big + i] = L1num * w; den[big + i] = L1s * (tp - w); w = w * ws; }
        });
    }
    if (batch_invert(den)) { set_error("domain_lagrange: t is a domain point"); return ZKG_ERROR; }
    chunked(m, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) u[i] = u[i] * den[i]; });
    return ZKG_OK;
}

struct StepArgs {
    Fr *a; const Fr *w, *winv_half, *g;      // g: g^i (fold, coset) or g^-i (unfold, icoset); null for the plain transforms
    size_t big, small, stride; uint32_t compr;
    Fr half;
};
// coefficients -> (c | e): c = p mod (x^big - 1) in a[0, big), e = p(omega x) mod (x^small - 1) in a[big, m)
__global__ __launch_bounds__(256) void k_step_fold(StepArgs A) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.small) return;
    Fr *a = A.a + (size_t)blockIdx.y * A.stride;
    Fr hi = a[A.big + i];
    if (A.g) hi = hi * A.g[A.big + i];
    Fr e = Fr::zero();
    for (uint32_t j = 0; j < A.compr; ++j) {
        size_t k = i + (size_t)j * A.small;
        Fr x = a[k];
        if (A.g) x = x * A.g[k];
        Fr c = x, d = x;
        if (j == 0) { c = x + hi; d = x - hi; }
        a[k] = c.normalized();
        e += A.w[k] * d;
    }
    a[A.big + i] = e.normalized();
}
// (U0 | U1) = (p mod (x^big - 1) | p(omega x) mod (x^small - 1)) -> the m coefficients of p
__global__ __launch_bounds__(256) void k_step_unfold(StepArgs A) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.small) return;
    Fr *a = A.a + (size_t)blockIdx.y * A.stride;
    Fr u1 = a[A.big + i];
    for (uint32_t j = 1; j < A.compr; ++j) {                 // the suffix coefficients are U0's own; remove their image from U1
        size_t k = i + (size_t)j * A.small;
        Fr x = a[k];
        u1 -= x * A.w[k];
        if (A.g) a[k] = (x * A.g[k]).normalized();
    }
    Fr u1h = u1 * A.winv_half[i], u0h = a[i] * A.half;
    Fr lo = u0h + u1h, hi = u0h - u1h;
    if (A.g) { lo = lo * A.g[i]; hi = hi * A.g[A.big + i]; }
    a[i] = lo.normalized(); a[A.big + i] = hi.normalized();
}

// big/small > STEP_J: the walk over the big/small indices congruent to i is cut into chunks of STEP_J, one lane per (i, chunk); the chunks'
// partial sums of w^k d_k land in `part` (chunk-major) and a second pass adds them up.  (At 19 payloads zklaim's domain is 2^19 + 2^11:
// 2048 lanes walking 256 indices each made that proof the outlier of the k = 1..20 sweep.)
static constexpr uint32_t STEP_J = 16;
__global__ __launch_bounds__(256) void k_step_fold_chunk(StepArgs A, Fr *part, size_t part_stride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nch = A.compr / STEP_J;
    if (t >= A.small * nch) return;
    const size_t i = t % A.small, ch = t / A.small;
    Fr *a = A.a + (size_t)blockIdx.y * A.stride;
    Fr hi = Fr::zero();
    if (ch == 0) { hi = a[A.big + i]; if (A.g) hi = hi * A.g[A.big + i]; }
    Fr e = Fr::zero();
    for (uint32_t jj = 0; jj < STEP_J; ++jj) {
        const size_t j = ch * STEP_J + jj, k = i + j * A.small;
        Fr x = a[k];
        if (A.g) x = x * A.g[k];
        Fr c = x, d = x;
        if (j == 0) { c = x + hi; d = x - hi; }
        a[k] = c.normalized();
        e += A.w[k] * d;
    }
    part[(size_t)blockIdx.y * part_stride + ch * A.small + i] = e.normalized();
}
__global__ __launch_bounds__(256) void k_step_fold_finish(StepArgs A, const Fr *part, size_t part_stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nch = A.compr / STEP_J;
    if (i >= A.small) return;
    Fr e = Fr::zero();
    for (size_t ch = 0; ch < nch; ++ch) e += part[(size_t)blockIdx.y * part_stride + ch * A.small + i];
    A.a[(size_t)blockIdx.y * A.stride + A.big + i] = e.normalized();
}
__global__ __launch_bounds__(256) void k_step_unfold_chunk(StepArgs A, Fr *part, size_t part_stride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nch = A.compr / STEP_J;
    if (t >= A.small * nch) return;
    const size_t i = t % A.small, ch = t / A.small;
    Fr *a = A.a + (size_t)blockIdx.y * A.stride;
    Fr sum = Fr::zero();
    for (uint32_t jj = 0; jj < STEP_J; ++jj) {
        const size_t j = ch * STEP_J + jj, k = i + j * A.small;
        if (j == 0) continue;                                   // the two overlapping coefficients are solved for in the finishing pass
        Fr x = a[k];
        sum += x * A.w[k];
        if (A.g) a[k] = (x * A.g[k]).normalized();
    }
    part[(size_t)blockIdx.y * part_stride + ch * A.small + i] = sum.normalized();
}
__global__ __launch_bounds__(256) void k_step_unfold_finish(StepArgs A, const Fr *part, size_t part_stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nch = A.compr / STEP_J;
    if (i >= A.small) return;
    Fr *a = A.a + (size_t)blockIdx.y * A.stride;
    Fr u1 = a[A.big + i];
    for (size_t ch = 0; ch < nch; ++ch) u1 -= part[(size_t)blockIdx.y * part_stride + ch * A.small + i];
    Fr u1h = u1 * A.winv_half[i], u0h = a[i] * A.half;
    Fr lo = u0h + u1h, hi = u0h - u1h;
    if (A.g) { lo = lo * A.g[i]; hi = hi * A.g[A.big + i]; }
    a[i] = lo.normalized(); a[A.big + i] = hi.normalized();
}

int StepDomain::init(const DomainShape &sh, hipStream_t s) {
    shape = sh;
    const size_t big = sh.big, small = sh.small, m = sh.m;
    dbig = ntt_domain(ceil_log2(big), s); dsmall = ntt_domain(ceil_log2(small), s);
    if (!dbig || !dsmall) return ZKG_ERROR;
    Fr omega = fr_root_of_unity_pow2(sh.log_m), g = Fr::from_u64(5);
    half = Fr::from_u64(2).inverse(); big_inv = Fr::from_u64(big).inverse(); small_inv = Fr::from_u64(small).inverse();
    const size_t compr = big / small;
    if (w.reserve(big * sizeof(Fr)) || winv_half.reserve(small * sizeof(Fr)) || g_pow.reserve(m * sizeof(Fr)) || ginv_pow.reserve(m * sizeof(Fr)) ||
        zinv.reserve(compr * sizeof(Fr))) return ZKG_ERROR;
    if (powers_table(w.as<Fr>(), big, omega, Fr::one(), s) || powers_table(winv_half.as<Fr>(), small, omega.inverse(), half, s) ||
        powers_table(g_pow.as<Fr>(), m, g, Fr::one(), s) || powers_table(ginv_pow.as<Fr>(), m, g.inverse(), Fr::one(), s)) return ZKG_ERROR;
    // divide_by_Z_on_coset: Z(x) = (x^big - 1)(x^small - omega^small) at x = g wb^i (i < big) and x = g omega ws^i
    Fr Z0 = g.pow_u64(big) - Fr::one(), gs = g.pow_u64(small), os = omega.pow_u64(small), step = omega.pow_u64(2 * small), elt = Fr::one();
    std::vector<Fr> den(compr);
    for (size_t i = 0; i < compr; ++i) { den[i] = Z0 * (gs * elt - os); elt = elt * step; }
    Fr go = g * omega, z1 = (go.pow_u64(big) - Fr::one()) * (go.pow_u64(small) - os);
    if (batch_invert(den) || z1.is_zero()) { set_error("step domain: Z vanishes on the coset"); return ZKG_ERROR; }
    zinv_small = z1.inverse();
    if (!hip_ok(hipMemcpyAsync(zinv.p, den.data(), compr * sizeof(Fr), hipMemcpyHostToDevice, s), "H2D", __FILE__, __LINE__) ||
        !hip_ok(hipStreamSynchronize(s), "sync", __FILE__, __LINE__)) return ZKG_ERROR;
    return ZKG_OK;
}
void StepDomain::release() { w.release(); winv_half.release(); g_pow.release(); ginv_pow.release(); zinv.release(); }

StepDomain *step_domain(size_t m, hipStream_t s) {
    DomainShape sh;
    if (!domain_shape_of(m, sh) || !sh.step) { set_error("step_domain: not a step_radix2 size"); return nullptr; }
    {
        std::lock_guard<std::mutex> lk(g_dom_mu);
        auto it = g_step_domains.find(m);
        if (it != g_step_domains.end()) return it->second;
    }
    StepDomain *d = new StepDomain();                         // built outside the lock: init takes it through ntt_domain()
    if (d->init(sh, s) != ZKG_OK) { d->release(); delete d; return nullptr; }
    std::lock_guard<std::mutex> lk(g_dom_mu);
    auto it = g_step_domains.find(m);
    if (it != g_step_domains.end()) { d->release(); delete d; return it->second; }
    g_step_domains[m] = d;
    return d;
}

int step_ntt_run(StepDomain *d, Fr *a, bool inverse, bool coset, hipStream_t s, Fr *scratch, unsigned batch, size_t batch_stride) {
    const size_t big = d->shape.big, small = d->shape.small, stride = batch_stride ? batch_stride : d->shape.m;
    if (!scratch) { set_error("step ntt: scratch required"); return ZKG_ERROR; }
    StepArgs A;
    A.a = a; A.w = d->w.as<Fr>(); A.winv_half = d->winv_half.as<Fr>(); A.big = big; A.small = small; A.stride = stride;
    A.compr = (uint32_t)(big / small); A.half = d->half;
    dim3 grid((unsigned)((small + 255) / 256), batch);
    const bool chunked = A.compr > STEP_J;                     // compr is a power of two: a multiple of STEP_J then
    dim3 grid_ch((unsigned)((small * (A.compr / STEP_J) + 255) / 256), batch);
    Fr *part = scratch;                                        // big / STEP_J elements per vector: the transforms' scratch is idle around them
    if (!inverse) {
        A.g = coset ? d->g_pow.as<Fr>() : nullptr;
        if (chunked) { hipLaunchKernelGGL(k_step_fold_chunk, grid_ch, dim3(256), 0, s, A, part, stride); hipLaunchKernelGGL(k_step_fold_finish, grid, dim3(256), 0, s, A, part, stride); }
        else hipLaunchKernelGGL(k_step_fold, grid, dim3(256), 0, s, A);
        if (ntt_run_ex(d->dbig, a, false, nullptr, nullptr, nullptr, s, scratch, batch, stride)) return ZKG_ERROR;
        if (ntt_run_ex(d->dsmall, a + big, false, nullptr, nullptr, nullptr, s, scratch + big, batch, stride)) return ZKG_ERROR;
    } else {
        if (ntt_run_ex(d->dbig, a, true, nullptr, nullptr, &d->big_inv, s, scratch, batch, stride)) return ZKG_ERROR;
        if (ntt_run_ex(d->dsmall, a + big, true, nullptr, nullptr, &d->small_inv, s, scratch + big, batch, stride)) return ZKG_ERROR;
        A.g = coset ? d->ginv_pow.as<Fr>() : nullptr;
        if (chunked) { hipLaunchKernelGGL(k_step_unfold_chunk, grid_ch, dim3(256), 0, s, A, part, stride); hipLaunchKernelGGL(k_step_unfold_finish, grid, dim3(256), 0, s, A, part, stride); }
        else hipLaunchKernelGGL(k_step_unfold, grid, dim3(256), 0, s, A);
    }
    if (hipGetLastError() != hipSuccess) { set_error("step ntt launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}

}  // namespace zk
