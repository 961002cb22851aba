// msm.hip — Pippenger multi-scalar multiplication over alt_bn128 G1 / G2 for gfx950.
//
// Replaces libff::multi_exp<G, Fr, multi_exp_method_BDLO12>, multi_exp_with_mixed_addition and
// kc_multi_exp_with_mixed_addition (the A/B/H/L query multi-exponentiations inside
// r1cs_gg_ppzksnark_prover, /root/reference/zklaim/snark.cpp:126).
//
// libff walks the c-bit windows serially and mixed-adds every base into buckets[digit].  EC
// addition is not atomic, so on the GPU the scatter is removed structurally:
//   1. k_digits_count   signed c-bit digits of every scalar; per-(window,bucket) histogram
//   2. k_scan           exclusive scan of the histogram -> bucket offsets
//   3. k_scatter        point indices (sign in bit 0) grouped by (window,bucket)
//   4. k_bucket_accum   one lane per bucket: sequential XYZZ mixed adds over its index list
//                       (the dominant kernel: N*W mixed additions, bases gathered 64/128 B at a time)
//      buckets longer than HEAVY_T entries (skewed scalars; the short top window) are cut into
//      512-entry parts summed by one wavefront each (k_heavy_parts) and merged per bucket by an
//      LDS tree (k_heavy_merge), so no lane ever walks a long list alone
//   5. k_bucket_reduce  sum_b (b+1)*B_b per 2048-bucket chunk: per-lane running sums, then an
//                       LDS suffix scan + tree reduction across the workgroup
//   6. host             per-window chunk combine and the c-doublings Horner across windows
// Scalars equal to 0 / 1 take libff's multi_exp_with_mixed_addition shortcut: zeros are
// dropped, ones are summed by a strided + LDS tree reduction (zklaim witnesses are ~97 % bits).
#include "common.hpp"
#include "../../include/zkg.h"
#include <algorithm>
#include <mutex>

namespace zk {

static constexpr int SCALAR_BITS = 255;      // r < 2^254; one extra bit absorbs the signed-digit carry
static constexpr int RED_THREADS = 256;
static constexpr int RED_L = 8;              // buckets per lane in the running-sum step
static constexpr int RED_CHUNK = RED_THREADS * RED_L;
static constexpr int ONES_BLOCKS = 128;
static constexpr uint32_t HEAVY_T = 256;     // buckets with more entries than this leave the lane-per-bucket kernel
static constexpr uint32_t HEAVY_S = 512;     // entries per heavy part (one wavefront sums one part)
static constexpr int HEAVY_PART_BLOCKS = 1024, HEAVY_MERGE_BLOCKS = 256;

struct HeavyItem { uint32_t start, end; };                    // a part: range of the sorted index list
struct HeavyBucket { uint32_t gb, first_item, nparts; };

struct MsmGeom { uint32_t c, W, B; };        // window bits, windows, buckets per window (2^(c-1))

static MsmGeom pick_geom(size_t n) {
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = lg - 4; if (c < 4) c = 4; if (c > 20) c = 20;
    MsmGeom g; g.c = c; g.W = (SCALAR_BITS + c - 1) / c; g.B = 1u << (c - 1);
    return g;
}

// ---- scalar -> signed digits ---------------------------------------------------------------
ZK_D uint32_t bits_at(const uint32_t v[8], uint32_t off, uint32_t c) {
    uint32_t limb = off >> 5, sh = off & 31;
    if (limb >= 8) return 0;
    uint64_t x = v[limb];
    if (limb + 1 < 8) x |= (uint64_t)v[limb + 1] << 32;
    return (uint32_t)(x >> sh) & ((1u << c) - 1);
}

struct ScalarRead { uint32_t v[8]; int cls; };       // cls: 0 zero, 1 one, 2 general
ZK_D ScalarRead read_scalar(const uint32_t *scalars, size_t i, int mont, int filter01) {
    ScalarRead s; Fr f;
    const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * i);
    uint4 a = p[0], b = p[1];
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    if (mont) f = f.from_mont();
    uint32_t hi = 0;
    for (int k = 1; k < 8; ++k) hi |= f.v[k];
    for (int k = 0; k < 8; ++k) s.v[k] = f.v[k];
    s.cls = 2;
    if (hi == 0 && f.v[0] == 0) s.cls = 0;                     // zero scalars never contribute
    else if (filter01 && hi == 0 && f.v[0] == 1) s.cls = 1;
    return s;
}

// visits every non-zero signed digit: fn(window, bucket (|d|-1), negative)
template <class Fn> ZK_D void for_each_digit(const uint32_t v[8], MsmGeom g, Fn fn) {
    uint32_t carry = 0;
    for (uint32_t w = 0; w < g.W; ++w) {
        uint32_t raw = bits_at(v, w * g.c, g.c) + carry;
        if (raw > g.B) { uint32_t d = (1u << g.c) - raw; carry = 1; if (d) fn(w, d - 1, 1u); }
        else { carry = 0; if (raw) fn(w, raw - 1, 0u); }
    }
}

__global__ void k_digits_count(const uint32_t *scalars, size_t n, int mont, int filter01, MsmGeom g,
                               uint32_t *counts, uint32_t *ones_idx, uint32_t *n_ones) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ScalarRead s = read_scalar(scalars, i, mont, filter01);
    if (s.cls == 0) return;
    if (s.cls == 1) { ones_idx[atomicAdd(n_ones, 1u)] = (uint32_t)i; return; }
    for_each_digit(s.v, g, [&](uint32_t w, uint32_t b, uint32_t) { atomicAdd(&counts[(size_t)w * g.B + b], 1u); });
}

__global__ void k_scatter(const uint32_t *scalars, size_t n, int mont, int filter01, MsmGeom g,
                          uint32_t *cursor, uint32_t *sorted) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ScalarRead s = read_scalar(scalars, i, mont, filter01);
    if (s.cls != 2) return;
    for_each_digit(s.v, g, [&](uint32_t w, uint32_t b, uint32_t neg) {
        uint32_t pos = atomicAdd(&cursor[(size_t)w * g.B + b], 1u);
        sorted[pos] = ((uint32_t)i << 1) | neg;
    });
}

// single-workgroup exclusive scan: offsets[i] = sum counts[<i]; cursor = copy; offsets[total] = sum
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *counts, uint32_t *offsets, uint32_t *cursor, size_t total) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    size_t per = (total + 1023) / 1024, lo = t * per < total ? t * per : total, hi = lo + per < total ? lo + per : total;
    uint32_t s = 0;
    for (size_t i = lo; i < hi; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (size_t i = lo; i < hi; ++i) { uint32_t c = counts[i]; offsets[i] = run; cursor[i] = run; run += c; }
    if (t == 1023) offsets[total] = part[1023];
}

// ---- bucket accumulation (dominant kernel) ---------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void k_bucket_accum(const Affine<F> *bases, const uint32_t *sorted, const uint32_t *offsets,
                                                       size_t total_buckets, XYZZ<F> *buckets,
                                                       HeavyItem *items, HeavyBucket *heavy, uint32_t *counters /* [0] items, [1] heavy buckets */) {
    size_t gb = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gb >= total_buckets) return;
    uint32_t k = offsets[gb], end = offsets[gb + 1];
    if (end - k > HEAVY_T) {
        uint32_t nparts = (end - k + HEAVY_S - 1) / HEAVY_S;
        uint32_t first = atomicAdd(&counters[0], nparts);
        for (uint32_t p = 0; p < nparts; ++p) { uint32_t a = k + p * HEAVY_S; items[first + p] = {a, a + HEAVY_S < end ? a + HEAVY_S : end}; }
        heavy[atomicAdd(&counters[1], 1u)] = {(uint32_t)gb, first, nparts};
        return;
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    for (; k < end; ++k) {
        uint32_t e = sorted[k];
        Affine<F> p = bases[e >> 1];
        if (e & 1u) p.y = p.y.neg();
        acc.madd(p);
    }
    buckets[gb] = acc;
}

// one wavefront per heavy part: 64 lanes stride over <= HEAVY_S entries, then a 6-level LDS tree
template <class F>
__global__ __launch_bounds__(256) void k_heavy_parts(const Affine<F> *bases, const uint32_t *sorted, const HeavyItem *items,
                                                      const uint32_t *counters, XYZZ<F> *partials) {
    extern __shared__ unsigned char red_smem[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(red_smem);            // 256 points
    const uint32_t n_items = counters[0], t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (uint32_t base = blockIdx.x * 4; base < n_items; base += gridDim.x * 4) {      // uniform trip count per workgroup
        uint32_t it = base + wv;
        XYZZ<F> acc = XYZZ<F>::inf();
        if (it < n_items) {
            HeavyItem h = items[it];
            for (uint32_t k = h.start + lane; k < h.end; k += 64) {
                uint32_t e = sorted[k];
                Affine<F> p = bases[e >> 1];
                if (e & 1u) p.y = p.y.neg();
                acc.madd(p);
            }
        }
        sh[t] = acc;
        __syncthreads();
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            if (lane < d) { XYZZ<F> a = sh[t]; a.add(sh[t + d]); sh[t] = a; }
            __syncthreads();
        }
        if (lane == 0 && it < n_items) partials[it] = sh[t];
        __syncthreads();
    }
}

// one workgroup per heavy bucket: strided sum of its parts' partials, LDS tree, write the bucket
template <class F>
__global__ __launch_bounds__(256) void k_heavy_merge(const HeavyBucket *heavy, const uint32_t *counters, const XYZZ<F> *partials, XYZZ<F> *buckets) {
    extern __shared__ unsigned char red_smem[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(red_smem);
    const uint32_t n_heavy = counters[1], t = threadIdx.x;
    for (uint32_t hb = blockIdx.x; hb < n_heavy; hb += gridDim.x) {
        HeavyBucket h = heavy[hb];
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t p = t; p < h.nparts; p += 256) acc.add(partials[h.first_item + p]);
        sh[t] = acc;
        __syncthreads();
        for (uint32_t d = 128; d >= 1; d >>= 1) {
            if (t < d) { XYZZ<F> a = sh[t]; a.add(sh[t + d]); sh[t] = a; }
            __syncthreads();
        }
        if (t == 0) buckets[h.gb] = sh[0];
        __syncthreads();
    }
}

// ---- bucket reduction: per chunk of RED_CHUNK buckets emit P = sum X_i and U = sum i*X_i (i 0-based in chunk)
template <class F>
__global__ __launch_bounds__(RED_THREADS) void k_bucket_reduce(const XYZZ<F> *buckets, uint32_t B, uint32_t chunks_per_window, XYZZ<F> *out) {
    extern __shared__ unsigned char red_smem[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(red_smem);              // 2 * RED_THREADS points
    const uint32_t t = threadIdx.x, w = blockIdx.x / chunks_per_window, ch = blockIdx.x % chunks_per_window;
    const XYZZ<F> *X = buckets + (size_t)w * B;
    const uint32_t base = ch * RED_CHUNK + t * RED_L;
    // lane-local running sums: S = sum_j X_j, T0 = sum_j j*X_j
    XYZZ<F> run = XYZZ<F>::inf(), T0 = XYZZ<F>::inf();
    for (int j = RED_L - 1; j >= 1; --j) {
        if (base + j < B) run.add(X[base + j]);
        T0.add(run);
    }
    if (base < B) run.add(X[base]);
    // inclusive suffix scan of S over lanes (Hillis-Steele through LDS): Q_t = sum_{u>=t} S_u
    XYZZ<F> Q = run;
    for (uint32_t d = 1; d < RED_THREADS; d <<= 1) {
        sh[t] = Q;
        __syncthreads();
        if (t + d < RED_THREADS) Q.add(sh[t + d]);
        __syncthreads();
    }
    // sum_t t*S_t = sum_{t>=1} Q_t ; tree-reduce Q (t>=1) in sh[0..), T0 in sh[RED_THREADS..)
    XYZZ<F> P = Q;                                                  // lane 0: total of the chunk
    sh[t] = (t >= 1) ? Q : XYZZ<F>::inf();
    sh[RED_THREADS + t] = T0;
    __syncthreads();
    for (uint32_t d = RED_THREADS / 2; d >= 1; d >>= 1) {
        if (t < d) { XYZZ<F> a = sh[t]; a.add(sh[t + d]); sh[t] = a; }
        else if (t >= RED_THREADS / 2 && t < RED_THREADS / 2 + d) {
            uint32_t u = RED_THREADS + (t - RED_THREADS / 2);
            XYZZ<F> a = sh[u]; a.add(sh[u + d]); sh[u] = a;
        }
        __syncthreads();
    }
    if (t == 0) {
        XYZZ<F> E = sh[0];
        for (int i = 0; i < 3; ++i) E = E.dbl();                    // * RED_L (= 8)
        E.add(sh[RED_THREADS]);
        out[2 * (size_t)blockIdx.x] = P;
        out[2 * (size_t)blockIdx.x + 1] = E;
    }
}

// ---- scalars == 1: plain sum of the selected bases
template <class F>
__global__ __launch_bounds__(256) void k_sum_ones(const Affine<F> *bases, const uint32_t *ones_idx, const uint32_t *n_ones, XYZZ<F> *out) {
    extern __shared__ unsigned char red_smem[];
    XYZZ<F> *sh = reinterpret_cast<XYZZ<F> *>(red_smem);
    const uint32_t n = *n_ones, t = threadIdx.x, stride = gridDim.x * blockDim.x;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = blockIdx.x * blockDim.x + t; k < n; k += stride) acc.madd(bases[ones_idx[k]]);
    sh[t] = acc;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (t < d) { XYZZ<F> a = sh[t]; a.add(sh[t + d]); sh[t] = a; }
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = sh[0];
}

// ---- workspace ------------------------------------------------------------------------------
struct MsmWorkspace {
    DevBuf counts, offsets, cursor, sorted, ones_idx, n_ones, buckets, red_out, ones_out, heavy_items, heavy_buckets, heavy_counters, heavy_partials;
    std::vector<unsigned char> host_red, host_ones;
    std::mutex mu;
};
static MsmWorkspace g_ws;

template <class F>
static int accumulate_and_reduce(const Affine<F> *d_bases, MsmGeom g, size_t n_entries_max, size_t total_buckets, XYZZ<F> *out, hipStream_t s, bool time_it) {
    MsmWorkspace &ws = g_ws;
    size_t max_heavy = n_entries_max / HEAVY_T + 1, max_items = n_entries_max / HEAVY_S + max_heavy + 1;
    if (ws.heavy_items.reserve(max_items * sizeof(HeavyItem)) || ws.heavy_buckets.reserve(max_heavy * sizeof(HeavyBucket)) ||
        ws.heavy_counters.reserve(8) || ws.heavy_partials.reserve(max_items * sizeof(XYZZ<F>))) return ZKG_ERROR;
    ZK_HIP(hipMemsetAsync(ws.heavy_counters.p, 0, 8, s));
    uint32_t cpw = (g.B + RED_CHUNK - 1) / RED_CHUNK;
    size_t nred = (size_t)g.W * cpw;
    if (ws.buckets.reserve(total_buckets * sizeof(XYZZ<F>)) || ws.red_out.reserve(nred * 2 * sizeof(XYZZ<F>)) ||
        ws.ones_out.reserve(ONES_BLOCKS * sizeof(XYZZ<F>))) return ZKG_ERROR;
    XYZZ<F> *buckets = ws.buckets.as<XYZZ<F>>();
    if (time_it) g_dominant_timer.begin(s);
    hipLaunchKernelGGL(k_bucket_accum<F>, dim3((unsigned)((total_buckets + 255) / 256)), dim3(256), 0, s,
                       d_bases, ws.sorted.as<uint32_t>(), ws.offsets.as<uint32_t>(), total_buckets, buckets,
                       ws.heavy_items.as<HeavyItem>(), ws.heavy_buckets.as<HeavyBucket>(), ws.heavy_counters.as<uint32_t>());
    if (time_it) g_dominant_timer.end(s);
    hipLaunchKernelGGL(k_heavy_parts<F>, dim3(HEAVY_PART_BLOCKS), dim3(256), 256 * sizeof(XYZZ<F>), s,
                       d_bases, ws.sorted.as<uint32_t>(), ws.heavy_items.as<HeavyItem>(), ws.heavy_counters.as<uint32_t>(), ws.heavy_partials.as<XYZZ<F>>());
    hipLaunchKernelGGL(k_heavy_merge<F>, dim3(HEAVY_MERGE_BLOCKS), dim3(256), 256 * sizeof(XYZZ<F>), s,
                       ws.heavy_buckets.as<HeavyBucket>(), ws.heavy_counters.as<uint32_t>(), ws.heavy_partials.as<XYZZ<F>>(), buckets);
    hipLaunchKernelGGL(k_bucket_reduce<F>, dim3((unsigned)nred), dim3(RED_THREADS), 2 * RED_THREADS * sizeof(XYZZ<F>), s,
                       buckets, g.B, cpw, ws.red_out.as<XYZZ<F>>());
    hipLaunchKernelGGL(k_sum_ones<F>, dim3(ONES_BLOCKS), dim3(256), 256 * sizeof(XYZZ<F>), s,
                       d_bases, ws.ones_idx.as<uint32_t>(), ws.n_ones.as<uint32_t>(), ws.ones_out.as<XYZZ<F>>());
    if (hipGetLastError() != hipSuccess) { set_error("msm kernel launch failed"); return ZKG_ERROR; }
    ws.host_red.resize(nred * 2 * sizeof(XYZZ<F>)); ws.host_ones.resize(ONES_BLOCKS * sizeof(XYZZ<F>));
    ZK_HIP(hipMemcpyAsync(ws.host_red.data(), ws.red_out.p, ws.host_red.size(), hipMemcpyDeviceToHost, s));
    ZK_HIP(hipMemcpyAsync(ws.host_ones.data(), ws.ones_out.p, ws.host_ones.size(), hipMemcpyDeviceToHost, s));
    ZK_HIP(hipStreamSynchronize(s));
    // host: window value V_w = sum_b (b+1) X_b = U_w + P_w, with chunk ch contributing
    //   U_ch + (ch*RED_CHUNK) * P_ch  to U_w and P_ch to P_w; then Horner over windows.
    const XYZZ<F> *red = reinterpret_cast<const XYZZ<F> *>(ws.host_red.data());
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = (int)g.W - 1; w >= 0; --w) {
        for (uint32_t i = 0; i < g.c; ++i) acc = acc.dbl();
        XYZZ<F> Usum = XYZZ<F>::inf(), suffix = XYZZ<F>::inf(), weighted = XYZZ<F>::inf();
        for (int ch = (int)cpw - 1; ch >= 0; --ch) {
            const XYZZ<F> &P = red[2 * ((size_t)w * cpw + ch)], &U = red[2 * ((size_t)w * cpw + ch) + 1];
            Usum.add(U);
            if (ch >= 1) { suffix.add(P); weighted.add(suffix); }           // sum_ch ch * P_ch
            else suffix.add(P);                                             // suffix == P_w now
        }
        for (int i = 0; i < 11; ++i) weighted = weighted.dbl();            // * RED_CHUNK (2048)
        acc.add(Usum); acc.add(weighted); acc.add(suffix);
    }
    const XYZZ<F> *ones = reinterpret_cast<const XYZZ<F> *>(ws.host_ones.data());
    for (int i = 0; i < ONES_BLOCKS; ++i) acc.add(ones[i]);
    *out = acc;
    return ZKG_OK;
}
static_assert(RED_CHUNK == 2048 && RED_L == 8, "host combine assumes 2048-bucket chunks");

static int sort_digits(const uint32_t *d_scalars, size_t n, bool mont, bool filter01, MsmGeom g, hipStream_t s) {
    MsmWorkspace &ws = g_ws;
    size_t total = (size_t)g.W * g.B;
    if (ws.counts.reserve(total * 4) || ws.offsets.reserve((total + 1) * 4) || ws.cursor.reserve(total * 4) ||
        ws.sorted.reserve(std::max<size_t>(1, n * g.W) * 4) || ws.ones_idx.reserve(std::max<size_t>(1, n) * 4) || ws.n_ones.reserve(4)) return ZKG_ERROR;
    ZK_HIP(hipMemsetAsync(ws.counts.p, 0, total * 4, s));
    ZK_HIP(hipMemsetAsync(ws.n_ones.p, 0, 4, s));
    if (n) hipLaunchKernelGGL(k_digits_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_scalars, n, (int)mont, (int)filter01, g,
                              ws.counts.as<uint32_t>(), ws.ones_idx.as<uint32_t>(), ws.n_ones.as<uint32_t>());
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, ws.counts.as<uint32_t>(), ws.offsets.as<uint32_t>(), ws.cursor.as<uint32_t>(), total);
    if (n) hipLaunchKernelGGL(k_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_scalars, n, (int)mont, (int)filter01, g,
                              ws.cursor.as<uint32_t>(), ws.sorted.as<uint32_t>());
    if (hipGetLastError() != hipSuccess) { set_error("msm sort launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}

int msm_shared(const G1Affine *const *d_g1_bases, int n_g1, const G2Affine *d_g2_bases, const uint32_t *d_scalars, size_t n,
               bool scalars_mont, bool filter01, G1 *out_g1, G2 *out_g2, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_ws.mu);
    if (n >= ((size_t)1 << 31)) { set_error("msm: n too large"); return ZKG_ERROR; }
    MsmGeom g = pick_geom(n);
    size_t total = (size_t)g.W * g.B;
    if (sort_digits(d_scalars, n, scalars_mont, filter01, g, s)) return ZKG_ERROR;
    for (int i = 0; i < n_g1; ++i)
        if (accumulate_and_reduce<Fq>(d_g1_bases[i], g, n * g.W, total, &out_g1[i], s, true)) return ZKG_ERROR;
    if (d_g2_bases && accumulate_and_reduce<Fq2>(d_g2_bases, g, n * g.W, total, out_g2, s, n_g1 == 0)) return ZKG_ERROR;
    return ZKG_OK;
}

int msm_g1(const G1Affine *d_bases, const uint32_t *d_scalars, size_t n, bool mont, bool filter01, G1 *out, hipStream_t s) {
    return msm_shared(&d_bases, 1, nullptr, d_scalars, n, mont, filter01, out, nullptr, s);
}
int msm_g2(const G2Affine *d_bases, const uint32_t *d_scalars, size_t n, bool mont, bool filter01, G2 *out, hipStream_t s) {
    return msm_shared(nullptr, 0, d_bases, d_scalars, n, mont, filter01, nullptr, out, s);
}

// ---- fixed-base batch: out[i] = k_i * base, table of 2^j * base (j < 254) ----------------------
template <class F>
__global__ __launch_bounds__(256) void k_fixed_base(const Affine<F> *table, const uint32_t *scalars, size_t n, Affine<F> *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
    for (int j = 0; j < 8; ++j) k[j] = scalars[8 * i + j];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int b = 0; b < 254; ++b)
        if ((k[b >> 5] >> (b & 31)) & 1u) acc.madd(table[b]);
    out[i] = acc.to_affine();
}

template <class F>
static int fixed_base(const Affine<F> &base, const uint32_t *d_scalars, size_t n, Affine<F> *d_out, hipStream_t s) {
    std::vector<Affine<F>> table(254);
    XYZZ<F> cur = XYZZ<F>::from_affine(base);
    for (int b = 0; b < 254; ++b) { table[b] = cur.to_affine(); cur = cur.dbl(); }
    DevBuf d_table;
    if (d_table.reserve(table.size() * sizeof(Affine<F>))) return ZKG_ERROR;
    ZK_HIP(hipMemcpyAsync(d_table.p, table.data(), table.size() * sizeof(Affine<F>), hipMemcpyHostToDevice, s));
    if (n) hipLaunchKernelGGL(k_fixed_base<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_table.as<Affine<F>>(), d_scalars, n, d_out);
    hipError_t e = hipGetLastError();
    ZK_HIP(hipStreamSynchronize(s));
    d_table.release();
    if (e != hipSuccess) { set_error("fixed_base launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}
int fixed_base_g1(const G1Affine &base, const uint32_t *d_scalars, size_t n, G1Affine *d_out, hipStream_t s) { return fixed_base<Fq>(base, d_scalars, n, d_out, s); }
int fixed_base_g2(const G2Affine &base, const uint32_t *d_scalars, size_t n, G2Affine *d_out, hipStream_t s) { return fixed_base<Fq2>(base, d_scalars, n, d_out, s); }

int msm_configure() {
    bool ok = hipFuncSetAttribute((const void *)k_bucket_reduce<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RED_THREADS * (int)sizeof(G2)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_bucket_reduce<Fq>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RED_THREADS * (int)sizeof(G1)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_sum_ones<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(G2)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_heavy_parts<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(G2)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_heavy_merge<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(G2)) == hipSuccess;
    return ok ? ZKG_OK : ZKG_ERROR;
}
void msm_release_all() {
    MsmWorkspace &ws = g_ws;
    std::lock_guard<std::mutex> lk(ws.mu);
    for (DevBuf *b : {&ws.counts, &ws.offsets, &ws.cursor, &ws.sorted, &ws.ones_idx, &ws.n_ones, &ws.buckets, &ws.red_out, &ws.ones_out, &ws.heavy_items, &ws.heavy_buckets, &ws.heavy_counters, &ws.heavy_partials}) b->release();
}

}  // namespace zk
