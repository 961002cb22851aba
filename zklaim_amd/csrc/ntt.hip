// ntt.hip — radix-2 NTT over alt_bn128 Fr for gfx950.
//
// Replaces libfqfft basic_radix2_domain<Fr>::FFT / iFFT / cosetFFT / icosetFFT
// (_basic_serial_radix2_FFT, _multiply_by_coset), reached from r1cs_to_qap_witness_map inside
// r1cs_gg_ppzksnark_prover (/root/reference/zklaim/snark.cpp:126).
//
// Shape: the log2(N) butterfly stages are cut into passes of R <= 8 stages.  One workgroup
// stages a tile of 2^R rows x CW columns (2048 elements = 64 KiB) in LDS, runs its R stages there
// and writes the tile back, so a 2^20 transform touches HBM 3 times instead of 20.  The
// bit-reversal permutation is folded into the first pass's gather (reads stay CW*32 B contiguous),
// coset / 1/N scalings are folded into the first load / last store.  Twiddles come from a
// per-domain table omega^i (i < N/2) that stays L2/MALL resident.
// MFMA is not used: the work is 254-bit modular multiplication on the integer VALU.
#include "common.hpp"
#include "../../include/zkg.h"
#include <map>
#include <mutex>

namespace zk {

static constexpr int NTT_THREADS = 512;
static constexpr int NTT_TILE_LOG = 11;          // 2048 elements per workgroup tile
static constexpr int NTT_MAX_R = 8;

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
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    A.src += (size_t)blockIdx.y * A.src_batch_stride; A.dst += (size_t)blockIdx.y * A.dst_batch_stride;
    const uint32_t s1 = A.s0 + A.R;
    const uint32_t lo_mask = (1u << A.s0) - 1;
    auto lds_ld = [&](uint32_t row, uint32_t c) { Fr r; const U4 *p = &smem[row * stride + 2 * c]; *reinterpret_cast<U4 *>(&r.v[0]) = p[0]; *reinterpret_cast<U4 *>(&r.v[4]) = p[1]; return r; };
    auto lds_st = [&](uint32_t row, uint32_t c, const Fr &r) { U4 *p = &smem[row * stride + 2 * c]; p[0] = *reinterpret_cast<const U4 *>(&r.v[0]); p[1] = *reinterpret_cast<const U4 *>(&r.v[4]); };

    // ---- load tile (c fastest: CW*32 B contiguous per row)
    for (uint32_t e = tid; e < rows * CW; e += NTT_THREADS) {
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
        for (uint32_t bf = tid; bf < (rows >> 1) * CW; bf += NTT_THREADS) {
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
    for (uint32_t e = tid; e < rows * CW; e += NTT_THREADS) {
        uint32_t c, mid; size_t idx;
        if (A.first) { mid = e & (rows - 1); c = e >> A.R; idx = ((size_t)bitrev(tile * CW + c, A.n_log - A.R) << A.R) + mid; }
        else { c = e & (CW - 1); mid = e >> A.cw_log; uint32_t g = tile * CW + c; idx = ((size_t)(g >> A.s0) << s1) + ((size_t)mid << A.s0) + (g & lo_mask); }
        Fr v = lds_ld(mid, c);
        if (A.post) v = v * A.post[idx];
        else if (A.has_post_scalar) v = v * A.post_scalar;
        A.dst[idx] = v.normalized();
    }
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

int NttDomain::init(unsigned logn_, hipStream_t s) {
    logn = logn_;
    size_t N = (size_t)1 << logn;
    Fr omega = fr_root_of_unity();
    for (unsigned i = 28; i > logn; --i) omega = omega.sqr();          // libff::get_root_of_unity
    Fr g = Fr::from_u64(5);                                            // Fr::multiplicative_generator
    n_inv = Fr::from_u64(N).inverse();
    size_t half = N > 1 ? N / 2 : 1;
    if (tw_fwd.reserve(half * sizeof(Fr)) || tw_inv.reserve(half * sizeof(Fr)) || coset_pre.reserve(N * sizeof(Fr)) ||
        icoset_post.reserve(N * sizeof(Fr)) || scratch.reserve(N * sizeof(Fr))) return ZKG_ERROR;
    if (powers_table(tw_fwd.as<Fr>(), half, omega, Fr::one(), s)) return ZKG_ERROR;
    if (powers_table(tw_inv.as<Fr>(), half, omega.inverse(), Fr::one(), s)) return ZKG_ERROR;
    if (powers_table(coset_pre.as<Fr>(), N, g, Fr::one(), s)) return ZKG_ERROR;
    if (powers_table(icoset_post.as<Fr>(), N, g.inverse(), n_inv, s)) return ZKG_ERROR;
    return ZKG_OK;
}
void NttDomain::release() {
    tw_fwd.release(); tw_inv.release(); coset_pre.release(); icoset_post.release(); scratch.release();
    for (auto &kv : stream_scratch) kv.second.release();
    stream_scratch.clear();
}
// transforms of one size on different streams must not share the inter-pass scratch vector
Fr *NttDomain::scratch_for(hipStream_t s) {
    std::lock_guard<std::mutex> lk(mu);
    if (first_stream_set && s == first_stream) return scratch.as<Fr>();
    if (!first_stream_set) { first_stream_set = true; first_stream = s; return scratch.as<Fr>(); }
    DevBuf &b = stream_scratch[s];
    if (b.reserve(((size_t)1 << logn) * sizeof(Fr))) return nullptr;
    return b.as<Fr>();
}

static std::mutex g_dom_mu;
static std::map<unsigned, NttDomain *> g_domains;
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
    return hipFuncSetAttribute((const void *)k_ntt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess ? ZKG_OK : ZKG_ERROR;
}
void ntt_release_all() {
    std::lock_guard<std::mutex> lk(g_dom_mu);
    for (auto &kv : g_domains) { kv.second->release(); delete kv.second; }
    g_domains.clear();
}

int ntt_run_ex(NttDomain *d, Fr *d_a, bool inverse, const Fr *pre, const Fr *post, const Fr *post_scalar, hipStream_t s, Fr *scratch, unsigned batch) {
    const unsigned n = d->logn;
    if (n == 0) return ZKG_OK;                 // N = 1: every variant is the identity (g^0 = 1, 1/N = 1)
    const size_t N = (size_t)1 << n;
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
        A.src_batch_stride = N; A.dst_batch_stride = N;     // batched vectors (and their scratch) are laid out back to back
        unsigned cols_log = n - R;                                   // columns in total
        A.cw_log = cols_log < (unsigned)(NTT_TILE_LOG - R) ? cols_log : (unsigned)(NTT_TILE_LOG - R);
        if (npass == 1) A.cw_log = 0;
        size_t tiles = (N >> R) >> A.cw_log;
        size_t rows = (size_t)1 << R, CW = (size_t)1 << A.cw_log;
        size_t lds = rows * (2 * CW + 1) * 16;
        hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)tiles, batch), dim3(NTT_THREADS), lds, s, A);
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

}  // namespace zk
