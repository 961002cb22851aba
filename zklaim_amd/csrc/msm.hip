// msm.hip — Pippenger multi-scalar multiplication over alt_bn128 G1 / G2 for gfx950.
//
// Replaces libff::multi_exp<G, Fr, multi_exp_method_BDLO12>, multi_exp_with_mixed_addition and
// kc_multi_exp_with_mixed_addition (the A/B/H/L query multi-exponentiations inside
// r1cs_gg_ppzksnark_prover, /root/reference/zklaim/snark.cpp:126).
//
// libff walks the c-bit windows serially and mixed-adds every base into buckets[digit].  EC addition is
// not atomic, so on the GPU the scatter is removed structurally — a counting sort by (window, bucket)
// followed by conflict-free accumulation:
//   1. k_digits        signed c-bit digits of every scalar, written window-major (4 B per digit)
//   2-4. the sort, in one of two forms (sort_digits picks):
//        two-pass (uniform scalars, 4096 <= n < 2^23, c >= 12): k_rx_count / k_rx_scan count and lay out coarse bins of buckets per
//                      window; k_rx_scatter splits each 16K-point slice into those bins inside LDS and writes every bin's run
//                      contiguously; k_rx_fine sorts one bin by the remaining bucket bits inside LDS and writes the final list and
//                      the bucket counts / offsets in order.  No single-word scatter to global memory anywhere.
//        one-pass (small or huge n, small c, ZKG_SCALARS_MOSTLY_BITS): k_hist — one workgroup per (window, 64K-point slice),
//                      histogram in LDS (2^(c-1) counters, up to 128 KiB of the CU's 160 KiB); k_colscan + block scan -> bucket
//                      offsets; k_place — bucket cursors staged in LDS, point indices (sign in bit 0) stored at their final
//                      position.  No global atomics, so skewed digits (bit witnesses) cost nothing extra.
//   5. k_order         bucket ids ordered by descending length, so the 64 lanes of a wavefront walk
//                      lists of (nearly) equal length
//   6. k_bucket_accum  one lane per bucket: sequential XYZZ mixed adds over its index list, next base
//                      prefetched (the dominant kernel: N*W mixed additions, 64/128 B gathers).
//                      Buckets longer than heavy_t leave this kernel: they are cut into 512-entry parts
//                      summed by one wavefront each (k_heavy_parts) and merged per bucket by an LDS tree
//                      (k_heavy_merge), so no lane ever walks a long list alone
//   7. k_bucket_reduce sum_b (b+1)*B_b per chunk of 128 ... 2048 buckets: per-lane running sums, then an LDS suffix
//                      scan + tree reduction across the workgroup; every addition is shared by a DPP quad
//   8. host            per-window chunk combine and the c-doublings Horner across windows
// Zero scalars are dropped in step 1 and scalars equal to one simply land in bucket (window 0, digit 1),
// a "heavy" bucket: the effect of libff's multi_exp_with_mixed_addition prefilter without a special case.
//
// The prover's resident key adds two things (prover.hip drives them):
//   * per-window tables (window_table_build_*): level w holds 2^(c w) P_i, so a digit of any window weighs the same — step 6 gathers from
//     the digit's level, k_bucket_fold sums the W bucket sets bucket-wise (LDS tree over the windows), step 7 runs ONCE over 2^(c-1)
//     buckets and step 8 has no doubling left;
//   * the witness split (witness_classify, ones_sum_launch): multi_exp_with_mixed_addition as libff does it — zeros skipped, bases whose
//     scalar is one added flat (k_ones_sum, k_sum_partials), the remaining scalars through steps 1-8 as a gathered subset, several base
//     sets (A, B_g1, L) per launch (blockIdx.y).
// Also here: the generator's fixed-base batch (k_fixed_table, k_fixed_base: byte windows, in-lane batched normalisation).
#include "common.hpp"
#include "fq29.hip.hpp"
#include "../../include/zkg.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

// Register cap of one kernel.  gfx90a and later have one 512-entry file per lane for VGPRs and AGPRs, and LLVM doubles the attribute's
// value there before it compares it with the budget ("amdgpu-num-vgpr" counts the unified file): half the wanted number of VGPRs goes in.
#define ZK_VGPR_CAP(n) __attribute__((amdgpu_num_vgpr((n) / 2)))
static constexpr int ACC29_VGPRS = 192;

namespace zk {

static constexpr int SCALAR_BITS = 255;      // r < 2^254; one extra bit absorbs the signed-digit carry
// logical lanes per reduce workgroup, four physical lanes (one DPP quad) each.  G1: 128 (512 threads, 128-VGPR budget is enough for
// three 32-register points).  G2: 64 -> 256 threads, so that the compiler may use 256 VGPRs: the kernel holds three 64-register
// points and was spilling 1.4 KB per lane under the 128-register cap of a 512-thread workgroup.
template <class F> struct RedGeom { static constexpr int LANES_LOG = sizeof(F) == sizeof(Fq) ? 7 : 6, LANES = 1 << LANES_LOG, THREADS = 4 * LANES; };
// buckets per logical lane in the running-sum step, 2^L_LOG: the kernel is a dependency chain of 2(L-1) + 2 log2(128) + 2
// additions but does (2(L-1) + 14)/L additions per bucket, so small bucket sets (latency-bound: the prover's witness MSMs)
// take L = 4 and large ones (work-bound: 2^19 buckets at N = 2^20) take L = 16.
static constexpr int RED_L_LOG_TINY = 0, RED_L_LOG_SMALL = 2, RED_L_LOG_LARGE = 4;
// up to 2^16 buckets one bucket per logical lane still is a single round of workgroups (512 of them for G1): the chain is 2 log2(LANES) + 3
// additions, 17 instead of 23 for G1 — every table launch of the prover (one folded bucket set of <= 2^15 buckets) is in this class
static constexpr size_t RED_SMALL_BUCKETS = (size_t)1 << 16, RED_LARGE_BUCKETS = (size_t)1 << 18;
static constexpr int MAX_C = 16;             // LDS histogram: 2^(c-1) u32 counters <= 128 KiB
static constexpr uint32_t HEAVY_S = 512;     // entries per heavy part (one wavefront sums one part)
static constexpr uint32_t HEAVY_T_MAX = 1024;
static constexpr int HEAVY_PART_BLOCKS = 512, HEAVY_MERGE_BLOCKS = 128;
static constexpr int SORT_THREADS = 1024;
static constexpr int SCAN_ITEMS = 4;         // per thread in the block scan

// window bits, windows handled by this launch, buckets per window (2^(c-1)); the launch owns windows w0, w0 + ws, ... of the Wt
// windows of the scalar (w0 = 0, ws = 1: all of them; a window-sharded multi-GPU run gives rank g the set w0 = g, ws = G)
struct MsmGeom { uint32_t c, W, B, Wt, w0, ws; };
// How a sorted entry finds its base.  Plain set: bases[i].  Per-window table of a resident key (level w holds 2^(c w) P_i, built once by
// window_table_build): bases[w * level_stride + i], w = the window of the entry's bucket — every window's digit then weighs the same, so
// the W bucket sets are summed bucket-wise (k_bucket_fold) and reduced ONCE, and no doubling is left for the host.  `gather`: the scalars
// were a gathered subset (a witness' non-bit values); entry i stands for element gather[i] of the set.  `index_sub`: the set starts at
// that element (the L query starts behind the public inputs); entries below it belong to no base of this set.
// The base sets of one field in a launch are batched: every kernel below runs once for all of them, blockIdx.y = set, each set with its
// own accumulators, heavy lists and results laid out `set` strides apart (SetLayout), all walking the same sorted digit list.
struct SetLayout { size_t buckets, items, heavy, partials, folded, red_out; };     // elements per set in each per-set array
template <class F> struct BaseView {
    const Affine<F> *p; size_t level_stride; const uint32_t *gather; uint32_t index_sub, B; const uint32_t *remap;
    template <bool PLAIN = false> ZK_D Affine<F> load(uint32_t e, size_t gb) const {
        if constexpr (PLAIN) return p[e >> 1];            // a plain base set: nothing but the gather (the 2^20-point multi_exp's inner loop)
        uint32_t i = e >> 1;
        if (gather) i = gather[i];
        if (remap) i = remap[i];                          // a subset table: position of element i in it
        if (i < index_sub) return Affine<F>::inf();
        return p[(level_stride ? (gb / B) * level_stride : 0) + (i - index_sub)];
    }
};
template <class F> struct ViewSet { BaseView<F> v[MSM_MAX_SETS]; };
struct HeavyItem { uint32_t start, end, gb, single; };   // a part: range of the sorted index list, the bucket it belongs to, and whether it is that bucket's only part
struct HeavyBucket { uint32_t gb, first_item, nparts; };

// Window size.  Measured on MI355X with uniformly random scalars (tools/msm_c_sweep.py): the accumulation runs one lane per
// bucket, so it needs W * 2^(c-1) >> 64 K buckets to fill the chip, and that outweighs the extra bucket-reduction work down to
// 2^15 points: c = 16 there, 12 below, and small windows only for tiny inputs.  `window_hint` (the prover's witness
// multi-exponentiations, whose scalars are mostly 0/1 so that the bucket reduction dominates) overrides the rule.
static MsmGeom pick_geom(size_t n, int window_hint, uint32_t w0 = 0, uint32_t ws = 1) {
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = lg >= 15 ? 16 : lg >= 11 ? 12 : lg + 1;
    if (c < 4) c = 4;
    if (window_hint > 0) c = window_hint;
    static const char *force = getenv("ZKG_MSM_C");                          // tuning aid
    if (force && atoi(force) >= 2) c = atoi(force);
    if (c > MAX_C) c = MAX_C;
    MsmGeom g; g.c = c; g.Wt = (SCALAR_BITS + c - 1) / c; g.B = 1u << (c - 1);
    g.w0 = w0; g.ws = ws ? ws : 1; g.W = w0 < g.Wt ? (g.Wt - w0 + g.ws - 1) / g.ws : 0;
    return g;
}

// ---- 1. scalar -> signed digits -----------------------------------------------------------------
ZK_D uint32_t bits_at(const uint32_t v[8], uint32_t off, uint32_t c) {
    uint32_t limb = off >> 5, sh = off & 31;
    if (limb >= 8) return 0;
    uint64_t x = v[limb];
    if (limb + 1 < 8) x |= (uint64_t)v[limb + 1] << 32;
    return (uint32_t)(x >> sh) & ((1u << c) - 1);
}

// The digit sort's kernels raise their wavefronts' issue priority: in a piece-wise job they share SIMDs with the accumulation of the piece
// before (k_bucket_accum29 leaves them the registers, ACC29_VGPRS), whose older, multiply-bound wavefronts would otherwise win every
// arbitration — k_digits_c4 took 113 ... 183 us beside an accumulation, 10 us alone.  The sort is 5 % of the step's instructions and on its critical path.
ZK_D void sort_wave_priority() { __builtin_amdgcn_s_setprio(3); }


// digit code: 0 = no contribution; otherwise ((bucket + 1) << 1) | negative, bucket = |d| - 1
// The job's first kernel also clears the counters its later kernels accumulate into (coarse-bin counts, class histogram, heavy-bucket
// counters): three hipMemsetAsync fill launches per job were three more dependent launches on a proof's critical path.
struct ZeroList { uint32_t *p[4]; uint32_t words[4]; };
__global__ __launch_bounds__(256) void k_digits(const uint32_t *scalars, const uint32_t *gather, size_t n, int mont, MsmGeom g, uint32_t *digits, ZeroList zl) {
    sort_wave_priority();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) for (size_t z = i; z < zl.words[k]; z += lanes) zl.p[k][z] = 0;
    if (i >= n) return;
    Fr f;
    const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * (gather ? (size_t)gather[i] : i));
    uint4 a = p[0], b = p[1];
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    if (mont) f = f.from_mont();
    uint32_t carry = 0;
    for (uint32_t w = 0, own = g.w0, slot = 0; w < g.Wt; ++w) {                 // the carry chain runs over every window; owned ones are stored
        uint32_t raw = bits_at(f.v, w * g.c, g.c) + carry, code = 0;
        if (raw > g.B) { uint32_t d = (1u << g.c) - raw; carry = 1; if (d) code = (d << 1) | 1u; }
        else { carry = 0; if (raw) code = raw << 1; }
        if (w == own) { digits[(size_t)slot * n + i] = code; ++slot; own += g.ws; }
    }
}

// The same with the window size known at compile time (16 and 12: every launch of the shipped geometry): the window loop unrolls, every limb
// index and shift is a constant, and the scalar stays in registers — the generic kernel indexes its limbs by a run-time window offset, which
// the compiler serves from an LDS copy of the scalar with two dependent reads per window (k_digits took 55 us at 2^20 points: 2.5 x what its
// 96 MB of traffic cost).
template <int C>
__global__ __launch_bounds__(256) void k_digits_c(const uint32_t *scalars, const uint32_t *gather, size_t n, int mont, MsmGeom g, uint32_t *digits, ZeroList zl) {
    sort_wave_priority();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) for (size_t z = i; z < zl.words[k]; z += lanes) zl.p[k][z] = 0;
    if (i >= n) return;
    Fr f;
    const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * (gather ? (size_t)gather[i] : i));
    uint4 a = p[0], b = p[1];
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    if (mont) f = f.from_mont();
    constexpr int WT = (SCALAR_BITS + C - 1) / C;
    constexpr uint32_t B = 1u << (C - 1), MASK = (1u << C) - 1;
    uint32_t carry = 0, own = g.w0, slot = 0;
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        const int off = w * C, limb = off >> 5, sh = off & 31;
        uint32_t v = limb < 8 ? f.v[limb < 8 ? limb : 7] >> sh : 0u;
        if (sh + C > 32 && limb + 1 < 8) v |= f.v[limb + 1 < 8 ? limb + 1 : 7] << (32 - sh);
        const uint32_t raw = (v & MASK) + carry;
        const bool neg = raw > B;
        const uint32_t d = neg ? (1u << C) - raw : raw;                          // |digit|; 2^C - raw = 0 only for raw = 2^C (a pure carry)
        carry = neg ? 1u : 0u;
        const uint32_t code = d ? (d << 1) | (neg ? 1u : 0u) : 0u;
        if ((uint32_t)w == own) { digits[(size_t)slot * n + i] = code; ++slot; own += g.ws; }
    }
}

// ... and four consecutive scalars per thread (no gather list, n a multiple of 4, every window owned): a window's four digits leave as ONE 16-byte
// store per lane — 1 KiB per wavefront instead of four 256-byte stores (the kernel is bound by its store instructions: 16.7 M four-byte stores at 2^20)
template <int C>
__global__ __launch_bounds__(256) void k_digits_c4(const uint32_t *scalars, size_t n, int mont, MsmGeom g, uint32_t *digits, ZeroList zl) {
    sort_wave_priority();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) for (size_t z = i; z < zl.words[k]; z += lanes) zl.p[k][z] = 0;
    if (4 * i >= n) return;
    Fr f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * (4 * i + q));
        const uint4 a = p[0], b = p[1];
        f[q].v[0] = a.x; f[q].v[1] = a.y; f[q].v[2] = a.z; f[q].v[3] = a.w; f[q].v[4] = b.x; f[q].v[5] = b.y; f[q].v[6] = b.z; f[q].v[7] = b.w;
        if (mont) f[q] = f[q].from_mont();
    }
    constexpr int WT = (SCALAR_BITS + C - 1) / C;
    constexpr uint32_t B = 1u << (C - 1), MASK = (1u << C) - 1;
    uint32_t carry[4] = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        const int off = w * C, limb = off >> 5, sh = off & 31;
        uint32_t code[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t v = limb < 8 ? f[q].v[limb < 8 ? limb : 7] >> sh : 0u;
            if (sh + C > 32 && limb + 1 < 8) v |= f[q].v[limb + 1 < 8 ? limb + 1 : 7] << (32 - sh);
            const uint32_t raw = (v & MASK) + carry[q];
            const bool neg = raw > B;
            const uint32_t d = neg ? (1u << C) - raw : raw;
            carry[q] = neg ? 1u : 0u;
            code[q] = d ? (d << 1) | (neg ? 1u : 0u) : 0u;
        }
        *reinterpret_cast<uint4 *>(digits + (size_t)w * n + 4 * i) = make_uint4(code[0], code[1], code[2], code[3]);
    }
}

// ---- 2. per-(window, slice) histogram in LDS ---------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_hist(const uint32_t *digits, size_t n, uint32_t B, uint32_t S, uint32_t slice_len, uint32_t *hist) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t s = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    for (uint32_t b = t; b < B; b += SORT_THREADS) lds_u32[b] = 0;
    __syncthreads();
    size_t lo = (size_t)s * slice_len, hi = lo + slice_len < n ? lo + slice_len : n;
    const uint32_t *d = digits + (size_t)w * n;
    for (size_t i = lo + t; i < hi; i += SORT_THREADS) { uint32_t code = d[i]; if (code) atomicAdd(&lds_u32[(code >> 1) - 1], 1u); }
    __syncthreads();
    uint32_t *out = hist + ((size_t)w * S + s) * B;
    for (uint32_t b = t; b < B; b += SORT_THREADS) out[b] = lds_u32[b];
}

// ---- 3. per-bucket totals; hist becomes the slice-relative offset inside its bucket -----------------------
__global__ __launch_bounds__(256) void k_colscan(uint32_t *hist, uint32_t B, uint32_t S, size_t total, uint32_t *counts) {
    size_t gb = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gb >= total) return;
    uint32_t w = (uint32_t)(gb / B), b = (uint32_t)(gb % B), run = 0;
    uint32_t *col = hist + (size_t)w * S * B + b;
    for (uint32_t s = 0; s < S; ++s) { uint32_t v = col[(size_t)s * B]; col[(size_t)s * B] = run; run += v; }
    counts[gb] = run;
}

// block-wise exclusive scan (three launches): offsets[i] = sum counts[<i], offsets[total] = sum
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t *in, uint32_t *out, uint32_t *sums, size_t total) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    size_t base = ((size_t)blockIdx.x * 1024 + t) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = base + j < total ? in[base + j] : 0; s += v[j]; }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t x = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (int j = 0; j < SCAN_ITEMS; ++j) { if (base + j < total) out[base + j] = run; run += v[j]; }
    if (t == 1023) sums[blockIdx.x] = part[1023];
}
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *sums, uint32_t nblk, uint32_t *grand_total) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    uint32_t s = t < nblk ? sums[t] : 0;
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t x = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    if (t < nblk) sums[t] = part[t] - s;
    if (t == 1023) *grand_total = part[1023];
}
__global__ __launch_bounds__(1024) void k_scan_apply(uint32_t *out, const uint32_t *sums, size_t total) {
    size_t base = ((size_t)blockIdx.x * 1024 + threadIdx.x) * SCAN_ITEMS;
    uint32_t add = sums[blockIdx.x];
    for (int j = 0; j < SCAN_ITEMS; ++j) if (base + j < total) out[base + j] += add;
}

// ---- 4. placement: cursors of this (window, slice) staged in LDS -----------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_place(const uint32_t *digits, size_t n, uint32_t B, uint32_t S, uint32_t slice_len,
                                                         const uint32_t *hist, const uint32_t *offsets, uint32_t *sorted) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t s = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    const uint32_t *rel = hist + ((size_t)w * S + s) * B, *off = offsets + (size_t)w * B;
    for (uint32_t b = t; b < B; b += SORT_THREADS) lds_u32[b] = off[b] + rel[b];
    __syncthreads();
    size_t lo = (size_t)s * slice_len, hi = lo + slice_len < n ? lo + slice_len : n;
    const uint32_t *d = digits + (size_t)w * n;
    for (size_t i = lo + t; i < hi; i += SORT_THREADS) {
        uint32_t code = d[i];
        if (code) { uint32_t pos = atomicAdd(&lds_u32[(code >> 1) - 1], 1u); sorted[pos] = ((uint32_t)i << 1) | (code & 1u); }
    }
}


// ---- 4b. two-pass sort (c >= 12, n < 2^22).  k_place's 4-byte stores each leave the L2 as their own 32-byte write (measured: 538 MB of
//      WRITE_SIZE for a 64 MiB list), so here no pass scatters single words to global memory: pass 1 splits a 16K-point slice into the
//      64 coarse bins of its window inside LDS and writes each bin's run contiguously; pass 2 takes one (window, coarse bin), sorts its
//      ~n/64 entries by the remaining bucket bits inside LDS and writes the final list — and the bucket counts and offsets — in order.
static constexpr uint32_t RX_MAX_CBITS = 10, RX_MAX_CB = 1u << RX_MAX_CBITS, RX_SLICE = 16384, RX_PER_THREAD = 16, RX_FINE_MAX = 8192, RX_BIN_AVG = 4096;
// temporary entry between the passes: fine bucket (fbits) in the top bits | point index << 1 | sign below.  cbits: log2 of the coarse bins
// per window, chosen on the host so that a bin averages <= 4096 entries, half of what the second pass holds in LDS (measured: 1365 ...
// 8192 all within 2 %).  The top window of a 254-bit scalar populates only 38 % of its buckets, so its bins are 2.65x as full and go to
// the streamed path of k_rx_fine (rx_fine_big_bin).
__global__ __launch_bounds__(1024) void k_rx_count(const uint32_t *digits, size_t n, uint32_t fbits, uint32_t cbits, uint32_t *cnt) {
    sort_wave_priority();
    __shared__ uint32_t c[RX_MAX_CB];
    const uint32_t sl = blockIdx.x, w = blockIdx.y, t = threadIdx.x, CB = 1u << cbits;
    if (t < CB) c[t] = 0;
    __syncthreads();
    size_t lo = (size_t)sl * RX_SLICE, hi = lo + RX_SLICE < n ? lo + RX_SLICE : n;
    const uint32_t *d = digits + (size_t)w * n;
    if ((n & 3) == 0) {                                                   // rows 16-byte aligned: four digits per load
        for (size_t i = lo + 4 * t; i < hi; i += 4096) {
            const uint4 q = *reinterpret_cast<const uint4 *>(d + i);     // (hi - lo is a multiple of 4 with n)
            if (q.x) atomicAdd(&c[((q.x >> 1) - 1) >> fbits], 1u);
            if (q.y) atomicAdd(&c[((q.y >> 1) - 1) >> fbits], 1u);
            if (q.z) atomicAdd(&c[((q.z >> 1) - 1) >> fbits], 1u);
            if (q.w) atomicAdd(&c[((q.w >> 1) - 1) >> fbits], 1u);
        }
    } else
    for (size_t i = lo + t; i < hi; i += 1024) { uint32_t code = d[i]; if (code) atomicAdd(&c[((code >> 1) - 1) >> fbits], 1u); }
    __syncthreads();
    if (t < CB && c[t]) atomicAdd(&cnt[w * CB + t], c[t]);
}
// exclusive scan of the W * 2^cbits bin counts (<= 16384 of them, 16 per thread) -> bin start positions; base[nbins] = number of entries
__global__ __launch_bounds__(1024) void k_rx_scan(const uint32_t *cnt, uint32_t nbins, uint32_t *base, uint32_t *cursor, uint32_t *grand_total) {
    sort_wave_priority();
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (nbins + 1023) / 1024, lo = t * per, hi = lo + per < nbins ? lo + per : nbins;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += cnt[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t x = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) { base[i] = run; cursor[i] = 0; run += cnt[i]; }
    if (t == 1023) { base[nbins] = part[1023]; *grand_total = part[1023]; }
}
// T threads per workgroup take a slice of 16 T digits.  T = 1024 (16K-digit slices, 64 KiB of LDS) when the sort has the chip to itself; T = 512 for
// the sorts of a piece-wise job, which run BESIDE the accumulation of the piece before: two wavefronts per SIMD of 56 registers fit into what
// k_bucket_accum29 leaves free (ACC29_VGPRS), four do not.
template <int T>
__global__ __launch_bounds__(T) void k_rx_scatter(const uint32_t *digits, size_t n, uint32_t fbits, uint32_t cbits, const uint32_t *base, uint32_t *cursor,
                                                   uint32_t *tmp) {
    sort_wave_priority();
    extern __shared__ uint32_t lds_u32[];
    constexpr uint32_t SLICE = RX_PER_THREAD * T;
    const uint32_t CB = 1u << cbits;
    uint32_t *stage = lds_u32, *c = lds_u32 + SLICE, *off = c + RX_MAX_CB, *gb = off + RX_MAX_CB + 1, *cur = gb + RX_MAX_CB;
    const uint32_t sl = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    for (uint32_t b = t; b < CB; b += T) c[b] = 0;
    __syncthreads();
    size_t lo = (size_t)sl * SLICE, hi = lo + SLICE < n ? lo + SLICE : n;
    const uint32_t *d = digits + (size_t)w * n;
    uint32_t code[RX_PER_THREAD];                                    // this thread's 16 digits stay in registers between the two passes
    // which element of the slice is this thread's j-th: four consecutive ones per 16-byte load when the rows are aligned (n a multiple of 4)
    const bool vec = (n & 3) == 0;
    auto elem = [&](uint32_t j) -> size_t { return vec ? lo + 4 * (size_t)t + (size_t)(j >> 2) * (4 * T) + (j & 3) : lo + t + (size_t)j * T; };
    if (vec) {
#pragma unroll
        for (uint32_t j = 0; j < RX_PER_THREAD; j += 4) {
            const size_t i = elem(j);
            uint4 q = make_uint4(0, 0, 0, 0);
            if (i < hi) q = *reinterpret_cast<const uint4 *>(d + i);
            code[j] = q.x; code[j + 1] = q.y; code[j + 2] = q.z; code[j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < RX_PER_THREAD; ++j) { const size_t i = elem(j); code[j] = i < hi ? d[i] : 0; }
    }
#pragma unroll
    for (uint32_t j = 0; j < RX_PER_THREAD; ++j) if (code[j]) atomicAdd(&c[((code[j] >> 1) - 1) >> fbits], 1u);
    __syncthreads();
    // exclusive scan of the bin counts (Hillis-Steele; a thread owns bins t and t + T: CB <= 1024 <= 2 T), off[CB] = entries of the slice
    const uint32_t b0 = t, b1 = t + T;
    if (b0 < CB) off[b0] = c[b0];
    if (b1 < CB) off[b1] = c[b1];
    __syncthreads();
    for (uint32_t dd = 1; dd < CB; dd <<= 1) {
        const uint32_t x0 = (b0 < CB && b0 >= dd) ? off[b0 - dd] : 0, x1 = (b1 < CB && b1 >= dd) ? off[b1 - dd] : 0;
        __syncthreads();
        if (b0 < CB) off[b0] += x0;
        if (b1 < CB) off[b1] += x1;
        __syncthreads();
    }
    if (t == 0) off[CB] = off[CB - 1];
    __syncthreads();
    for (uint32_t b = t; b < CB; b += T) { off[b] -= c[b]; gb[b] = c[b] ? base[w * CB + b] + atomicAdd(&cursor[w * CB + b], c[b]) : 0; cur[b] = 0; }
    __syncthreads();
    const uint32_t fmask = (1u << fbits) - 1;
#pragma unroll
    for (uint32_t j = 0; j < RX_PER_THREAD; ++j) {
        if (!code[j]) continue;
        uint32_t i = (uint32_t)elem(j), b = (code[j] >> 1) - 1, k = b >> fbits;
        uint32_t pos = off[k] + atomicAdd(&cur[k], 1u);
        stage[pos] = ((b & fmask) << (32 - fbits)) | (i << 1) | (code[j] & 1u);
    }
    __syncthreads();
    // runs go out contiguously, one wavefront per bin at a time (consecutive lanes -> consecutive addresses)
    const uint32_t wave = t >> 6, lane = t & 63;
    for (uint32_t k = wave; k < CB; k += T / 64) {
        const uint32_t a = off[k], b = off[k + 1], g = gb[k];
        for (uint32_t p = a + lane; p < b; p += 64) tmp[g + (p - a)] = stage[p];
    }
}
// shared by the two second-pass kernels: exclusive scan of the FB <= 512 counters (Hillis-Steele in LDS), counts / offsets to global
ZK_D void rx_fine_offsets(uint32_t *cf, uint32_t *of, uint32_t FB, uint32_t t, uint32_t start, size_t gb0, uint32_t *counts, uint32_t *offsets) {
    const uint32_t T = blockDim.x;                                    // FB <= 512 <= T
    if (t < FB) of[t] = cf[t];
    __syncthreads();
    for (uint32_t d = 1; d < FB; d <<= 1) {
        uint32_t x = (t < FB && t >= d) ? of[t - d] : 0;
        __syncthreads();
        if (t < FB) of[t] += x;
        __syncthreads();
    }
    if (t < FB) of[t] -= cf[t];
    __syncthreads();
    for (uint32_t f = t; f < FB; f += T) { counts[gb0 + f] = cf[f]; offsets[gb0 + f] = start + of[f]; }
}
// A bin with more entries than LDS holds, i.e. many equal digits (a 0/1 witness without ZKG_SCALARS_MOSTLY_BITS, adversarial input, the
// partly filled top window): the workgroup streams the bin twice; lanes of a wavefront that hold the same fine bucket share one LDS atomic
// and get consecutive positions, so the stores of the dominant bucket are coalesced.  (Round 2 ran this as a kernel of its own: one
// more launch per sort whose workgroups all returned at once for uniform scalars.)
ZK_D void rx_fine_big_bin(const uint32_t *tmp, uint32_t fbits, uint32_t start, uint32_t cnt, size_t gb0, uint32_t *cf, uint32_t *of, uint32_t *cur, uint32_t *counts,
                          uint32_t *offsets, uint32_t *sorted) {
    const uint32_t FB = 1u << fbits, sh = 32 - fbits, low = (1u << sh) - 1, t = threadIdx.x, T = blockDim.x;
    const bool giant = cnt > 8 * RX_FINE_MAX;                         // many equal digits; below that (a dense top window) plain atomics are faster
    if (!giant) {
        for (uint32_t p = t; p < cnt; p += T) atomicAdd(&cf[tmp[start + p] >> sh], 1u);
    } else
    for (uint32_t p0 = 0; p0 < cnt; p0 += T) {
        uint32_t p = p0 + t; bool live = p < cnt; uint32_t f = live ? tmp[start + p] >> sh : 0;
        while (__any(live)) {
            uint32_t lf = __shfl(f, __ffsll((long long)__ballot(live)) - 1, 64);
            unsigned long long same = __ballot(live && f == lf);
            if (live && f == lf) { if ((same & ((1ull << (threadIdx.x & 63)) - 1)) == 0) atomicAdd(&cf[lf], (uint32_t)__popcll(same)); live = false; }
        }
    }
    __syncthreads();
    rx_fine_offsets(cf, of, FB, t, start, gb0, counts, offsets);
    if (!giant) {
        for (uint32_t p = t; p < cnt; p += T) { uint32_t e = tmp[start + p], f = e >> sh; sorted[start + of[f] + atomicAdd(&cur[f], 1u)] = e & low; }
        return;
    }
    for (uint32_t p0 = 0; p0 < cnt; p0 += T) {
        uint32_t p = p0 + t; bool live = p < cnt; uint32_t e = live ? tmp[start + p] : 0, f = e >> sh;
        while (__any(live)) {
            uint32_t lf = __shfl(f, __ffsll((long long)__ballot(live)) - 1, 64);
            unsigned long long same = __ballot(live && f == lf), below = same & ((1ull << (threadIdx.x & 63)) - 1);
            uint32_t basepos = 0;
            if (live && f == lf && below == 0) basepos = atomicAdd(&cur[lf], (uint32_t)__popcll(same));
            basepos = __shfl(basepos, __ffsll((long long)same) - 1, 64);
            if (live && f == lf) { sorted[start + of[lf] + basepos + (uint32_t)__popcll(below)] = e & low; live = false; }
        }
    }
}
// second pass, one workgroup per (window, coarse bin).  A bin whose entries fit LDS (every bin of uniformly random scalars): entries stay in
// registers between the count and the placement, the sorted bin is assembled in LDS and written out in order
// (T threads: 1024, or 512 beside a running accumulation — see k_rx_scatter; the bin size the workgroup holds is the same)
template <int T>
__global__ __launch_bounds__(T) void k_rx_fine(const uint32_t *tmp, uint32_t fbits, uint32_t cbits, uint32_t B, const uint32_t *base, uint32_t *counts, uint32_t *offsets,
                                                uint32_t *sorted) {
    sort_wave_priority();
    extern __shared__ uint32_t lds_u32[];
    const uint32_t FB = 1u << fbits, sh = 32 - fbits, low = (1u << sh) - 1;
    uint32_t *bufB = lds_u32, *cf = bufB + RX_FINE_MAX, *of = cf + FB, *cur = of + FB;       // 64 KiB + counters: two workgroups per CU
    const uint32_t cb = blockIdx.x, w = blockIdx.y, t = threadIdx.x, bin = (w << cbits) + cb;
    const uint32_t start = base[bin], cnt = base[bin + 1] - start;
    const size_t gb0 = (size_t)w * B + (size_t)cb * FB;
    for (uint32_t f = t; f < FB; f += T) { cf[f] = 0; cur[f] = 0; }
    __syncthreads();
    if (cnt > RX_FINE_MAX) { rx_fine_big_bin(tmp, fbits, start, cnt, gb0, cf, of, cur, counts, offsets, sorted); return; }     // (uniform across the workgroup)
    uint32_t ent[RX_FINE_MAX / T];
#pragma unroll
    for (uint32_t j = 0; j < RX_FINE_MAX / T; ++j) {
        uint32_t p = t + j * T;
        ent[j] = p < cnt ? tmp[start + p] : 0xffffffffu;
        if (p < cnt) atomicAdd(&cf[ent[j] >> sh], 1u);
    }
    __syncthreads();
    rx_fine_offsets(cf, of, FB, t, start, gb0, counts, offsets);
#pragma unroll
    for (uint32_t j = 0; j < RX_FINE_MAX / T; ++j) {
        if (t + j * T >= cnt) continue;
        uint32_t e = ent[j], f = e >> sh;
        bufB[of[f] + atomicAdd(&cur[f], 1u)] = e & low;
    }
    __syncthreads();
    for (uint32_t p = t; p < cnt; p += T) sorted[start + p] = bufB[p];
}

// Heavy threshold, computed on the device from the real list lengths: a lane walks its bucket's list alone (~7 us per G1
// addition at low occupancy), so the longest non-heavy list bounds the kernel's latency however little total work there is.
// Lists are cut at 4x the true average length + 32 (the partly filled top window of uniform scalars averages ~2.7x the other
// windows and stays on the lane-per-bucket path; a sparse bit-witness MSM gets a small threshold and a short tail).
ZK_D uint32_t heavy_threshold_dev(const uint32_t *offsets, size_t total_buckets, uint32_t divisor) {
    uint32_t entries = offsets[total_buckets];
    // the slack shrinks with the average: a sparse launch (a witness' few non-bit scalars over 2^19 buckets) is pure latency, and its
    // longest lane-walked list is its duration
    uint32_t avg6 = (uint32_t)(6 * (uint64_t)entries / total_buckets), slack = 8 + avg6 > 32 ? 32 : 8 + avg6;
    uint32_t t = (uint32_t)(4 * (uint64_t)entries / total_buckets) + slack;
    t = t > HEAVY_T_MAX ? HEAVY_T_MAX : t;
    t /= divisor;
    return t < 4 ? 4 : t;
}

// ---- 5. bucket order by descending length (classes 0..heavy_t, heavy_t+1 = heavy) ------------------------
__global__ __launch_bounds__(1024) void k_class_hist(const uint32_t *counts, const uint32_t *offsets, size_t total, uint32_t *class_hist) {
    sort_wave_priority();
    __shared__ uint32_t h[HEAVY_T_MAX + 2];
    const uint32_t heavy_t = heavy_threshold_dev(offsets, total, 1);
    const uint32_t t = threadIdx.x, nc = heavy_t + 2;
    for (uint32_t i = t; i < nc; i += 1024) h[i] = 0;
    __syncthreads();
    size_t gb = (size_t)blockIdx.x * 1024 + t;
    if (gb < total) { uint32_t c = counts[gb]; atomicAdd(&h[c > heavy_t ? heavy_t + 1 : c], 1u); }
    __syncthreads();
    for (uint32_t i = t; i < nc; i += 1024) if (h[i]) atomicAdd(&class_hist[i], h[i]);
}
// in place: start position of every class, longest first (one 1024-thread workgroup; class c is scanned at position nc - 1 - c)
__global__ __launch_bounds__(1024) void k_class_scan(uint32_t *class_hist, const uint32_t *offsets, size_t total) {
    sort_wave_priority();
    __shared__ uint32_t part[1024];
    const uint32_t heavy_t = heavy_threshold_dev(offsets, total, 1), nc = heavy_t + 2, t = threadIdx.x;
    // HEAVY_T_MAX + 2 = 1026 classes at most: thread t takes positions 2t and 2t + 1 of the reversed order
    uint32_t v0 = 2 * t < nc ? class_hist[nc - 1 - 2 * t] : 0, v1 = 2 * t + 1 < nc ? class_hist[nc - 2 - 2 * t] : 0;
    part[t] = v0 + v1;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t x = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    const uint32_t before = part[t] - v0 - v1;
    if (2 * t < nc) class_hist[nc - 1 - 2 * t] = before;
    if (2 * t + 1 < nc) class_hist[nc - 2 - 2 * t] = before + v0;
}
__global__ __launch_bounds__(1024) void k_order_place(const uint32_t *counts, const uint32_t *offsets, size_t total, uint32_t *class_cursor, uint32_t *order) {
    sort_wave_priority();
    __shared__ uint32_t h[HEAVY_T_MAX + 2];
    const uint32_t heavy_t = heavy_threshold_dev(offsets, total, 1);
    const uint32_t t = threadIdx.x, nc = heavy_t + 2;
    for (uint32_t i = t; i < nc; i += 1024) h[i] = 0;
    __syncthreads();
    size_t gb = (size_t)blockIdx.x * 1024 + t;
    uint32_t cls = 0, rank = 0;
    if (gb < total) { uint32_t c = counts[gb]; cls = c > heavy_t ? heavy_t + 1 : c; rank = atomicAdd(&h[cls], 1u); }
    __syncthreads();
    for (uint32_t i = t; i < nc; i += 1024) if (h[i]) h[i] = atomicAdd(&class_cursor[i], h[i]);      // reserve a range per class
    __syncthreads();
    if (gb < total) order[h[cls] + rank] = (uint32_t)gb;
}

// ---- 6. bucket accumulation (dominant kernel) -------------------------------------------------------------
template <class F, bool PLAIN>
__global__ __launch_bounds__(256) void k_bucket_accum(const ViewSet<F> views, const uint32_t *sorted, const uint32_t *offsets, const uint32_t *order,
                                                       size_t lanes /* <= total buckets: the first `lanes` entries of order */, XYZZ<F> *buckets, HeavyItem *items,
                                                       HeavyBucket *heavy, uint32_t *counters /* per set: [0] items, [1] heavy buckets */, SetLayout L) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= lanes) return;
    const size_t total_buckets = L.buckets;
    const uint32_t set = blockIdx.y;
    const BaseView<F> bases = views.v[set];
    buckets += set * L.buckets; items += set * L.items; heavy += set * L.heavy; counters += 2 * set;
    const uint32_t heavy_t = heavy_threshold_dev(offsets, total_buckets, sizeof(F) > sizeof(Fq) ? 3 : 1);   // a G2 addition costs ~3x a G1 addition
    const uint32_t gb = order[tid];
    uint32_t k = offsets[gb], end = offsets[gb + 1];
    if (end - k > heavy_t) {
        uint32_t nparts = (end - k + HEAVY_S - 1) / HEAVY_S;
        uint32_t first = atomicAdd(&counters[0], nparts);
        for (uint32_t p = 0; p < nparts; ++p) { uint32_t a = k + p * HEAVY_S; items[first + p] = {a, a + HEAVY_S < end ? a + HEAVY_S : end, gb, nparts == 1}; }
        if (nparts > 1) heavy[atomicAdd(&counters[1], 1u)] = {gb, first, nparts};                   // a single part writes its bucket itself
        return;
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    if (k < end) {
        // two loads deep (see k_bucket_accum29): the base of entry k + 1 and the index of entry k + 2 arrive under the addition of entry k
        uint32_t e = sorted[k], e_next = k + 1 < end ? sorted[k + 1] : 0;
        Affine<F> p = bases.template load<PLAIN>(e, gb);
        while (true) {
            Affine<F> cur = p; uint32_t ce = e;
            ++k;
            if (k < end) { e = e_next; p = bases.template load<PLAIN>(e, gb); if (k + 1 < end) e_next = sorted[k + 1]; }
            if (ce & 1u) cur.y = cur.y.neg();
            acc.madd(cur);
            if (k >= end) break;
        }
    }
    buckets[gb] = acc.normalized();
}

// ---- 6'. the same accumulation for a plain G1 base set on the 29-bit representation (fq29.hip.hpp): 10 products of 895 cycles instead of
//      1173 and carry-free additions per mixed addition.  The bases are converted once per launch into 64-byte packed records (Rec64, k_bases_to29: two
//      products per point against the 160 the point costs in the loop); an accumulator is converted back when its bucket is written, so
//      everything downstream (heavy parts, fold, reduction, host) sees the same XYZZ<Fq> buckets as before.
__global__ __launch_bounds__(256) void k_bases_to29(const Affine<Fq> *in, size_t n, Rec64 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<Fq> a = in[i];
    store_rec64(out + i, f29::to29(a.x), f29::to29(a.y), a.is_inf());
}
ZK_D Affine29 load29(const Rec64 *bases, uint32_t e) { return load_rec64(bases + (e >> 1)); }
ZK_D Rec64 load29_raw(const Rec64 *bases, uint32_t e) { return load_rec64_raw(bases + (e >> 1)); }
// WAVES: wavefronts per SIMD the register allocation aims for (2: no spill in the loop; 3: 168 registers and a 232-byte spill frame)
// ACC29_VGPRS: left to itself the allocator spends all 256 registers two wavefronts per SIMD allow (253 for OUT29: loop-invariant constants
// parked in registers) and nothing else fits on the SIMD beside them.  Capped at 192 the addition's instruction stream is the same (2 281
// instructions, the spills all on the exceptional doubling path), and 128 registers per SIMD stay free: the digit sort of the NEXT piece of a
// piece-wise job (msm_g1_host_scalars) runs on the same compute units at once instead of waiting for accumulation workgroups to retire.
// level_stride != 0: `bases` is a per-window table (level w = 2^(c w) P_i, level_stride records apart) and a bucket reads its window's level
// OUT29: the buckets are Bucket29 records (the plain G1 path: the reduction reads them without conversion) instead of canonical XYZZ<Fq>
// (table launches: the fold in between works on 32-bit limbs).  resume (OUT29 only): the buckets hold the sums of the job's earlier pieces
// (piece-wise multi-exponentiation, msm_g1_host_scalars): a lane continues its bucket's accumulator, an empty list leaves it alone, and a
// heavy bucket's old value joins its parts in k_heavy_merge.
template <int WAVES, bool OUT29>
__global__ __launch_bounds__(256, WAVES) ZK_VGPR_CAP(ACC29_VGPRS) void k_bucket_accum29(const Rec64 *bases, size_t level_stride, uint32_t B, const uint32_t *sorted, const uint32_t *offsets, const uint32_t *order,
                                                         size_t lanes, void *buckets_, HeavyItem *items, HeavyBucket *heavy, uint32_t *counters, SetLayout L, int flags /* 1: resume, 2: critical path */) {
    crit_wave_priority(flags & 2);
    const int resume = flags & 1;
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= lanes) return;
    const uint32_t heavy_t = heavy_threshold_dev(offsets, L.buckets, 1);
    const uint32_t gb = order[tid];
    if (level_stride) bases += (size_t)(gb / B) * level_stride;
    uint32_t k = offsets[gb], end = offsets[gb + 1];
    if (end - k > heavy_t) {                                                    // as k_bucket_accum: long lists go to k_heavy_parts / k_heavy_merge
        uint32_t nparts = (end - k + HEAVY_S - 1) / HEAVY_S;
        uint32_t first = atomicAdd(&counters[0], nparts);
        const bool single = nparts == 1 && !resume;                             // a lone part may write its bucket itself only when nothing is there yet
        for (uint32_t p = 0; p < nparts; ++p) { uint32_t a = k + p * HEAVY_S; items[first + p] = {a, a + HEAVY_S < end ? a + HEAVY_S : end, gb, single}; }
        if (!single) heavy[atomicAdd(&counters[1], 1u)] = {gb, first, nparts};
        return;
    }
    XYZZ29 acc; bool inf = true;
    acc.x = acc.y = acc.zz = acc.zzz = Fq29::zero();
    if constexpr (OUT29) {
        if (resume) {
            if (k >= end) return;                                               // nothing of this piece lands here: the bucket keeps its sum
            const Bucket29 old = load_bucket29_raw(reinterpret_cast<const Bucket29 *>(buckets_) + gb);
            uint32_t any = 0;
#pragma unroll
            for (int j = 0; j < 9; ++j) { acc.x.v[j] = old.x[j]; acc.y.v[j] = old.y[j]; acc.zz.v[j] = old.zz[j]; acc.zzz.v[j] = old.zzz[j]; any |= old.zz[j]; }
            inf = any == 0;
        }
    }
    if (k < end) {
        // two loads deep: under the addition of entry k the base of entry k + 1 arrives (its address came from an index loaded one
        // addition earlier) and the index of entry k + 2 — neither the index nor the base is ever waited for right after it was requested
        uint32_t e = sorted[k], e_next = k + 1 < end ? sorted[k + 1] : 0;
        Rec64 p = load29_raw(bases, e);                                         // (kept packed while in flight: unpacked when its addition starts)
        while (true) {
            const Rec64 curw = p; const uint32_t ce = e;
            ++k;
            if (k < end) { e = e_next; p = load29_raw(bases, e); if (k + 1 < end) e_next = sorted[k + 1]; }
            const Affine29 cur = unpack_rec64(curw);
            if (!cur.inf) {
                Fq29 bx, by;
#pragma unroll
                for (int j = 0; j < 9; ++j) { bx.v[j] = cur.x[j]; by.v[j] = cur.y[j]; }
                if (ce & 1u) by = f29::neg(f29::S2_1, by);
                if (__builtin_expect(!acc.madd(bx, by, inf), 0)) {
                    // b == +-accumulator (a duplicated base under equal digits): the doubling / cancellation cases on the 32-bit path
                    XYZZ<Fq> a32 = {f29::from29(acc.x), f29::from29(acc.y), f29::from29(acc.zz), f29::from29(acc.zzz)};
                    a32.madd(Affine<Fq>{f29::from29(bx), f29::from29(f29::norm(by))});
                    if (a32.is_inf()) inf = true;
                    else { acc.x = f29::to29(a32.x.normalized()); acc.y = f29::to29(a32.y.normalized()); acc.zz = f29::to29(a32.zz.normalized()); acc.zzz = f29::to29(a32.zzz.normalized()); }
                }
            }
            if (k >= end) break;
        }
    }
    if constexpr (OUT29) store_bucket29(reinterpret_cast<Bucket29 *>(buckets_) + gb, acc, inf);
    else {
        XYZZ<Fq> *buckets = reinterpret_cast<XYZZ<Fq> *>(buckets_);
        if (inf) buckets[gb] = XYZZ<Fq>::inf().normalized();
        else buckets[gb] = XYZZ<Fq>{f29::from29(acc.x), f29::from29(acc.y), f29::from29(acc.zz), f29::from29(acc.zzz)};
    }
}

#include "msm_ba.inc"

// A point in LDS, padded to 144 B (G1) / 272 B (G2): at the natural 128 / 256-byte stride consecutive points start on the same four banks.
// Measured effect: k_bucket_reduce 363 -> 354 us at 2^19 buckets — the kernels that use it are bound by the multiplications of their
// addition chains (two wavefronts per SIMD: 1.16 us per dependent product), not by the LDS port.
template <class F> struct alignas(16) LdsPoint {
    XYZZ<F> p; uint32_t pad[4];
    ZK_D LdsPoint &operator=(const XYZZ<F> &v) { p = v; return *this; }
    ZK_D operator const XYZZ<F> &() const { return p; }
    ZK_D XYZZ<F> normalized() const { return p.normalized(); }
};

// LDS tree reduction of seg[0..n) into seg[0] by `nthreads` threads (tid = 0..nthreads-1, n a power of two <= nthreads;
// every thread of the group must call it; barrier() synchronises the group).  Levels with at most nthreads/4 pairs share
// each addition across a DPP quad (nthreads is a multiple of 64, so quads never straddle groups).
template <class F, class Barrier>
ZK_D void lds_tree_reduce(LdsPoint<F> *seg, uint32_t n, uint32_t tid, uint32_t nthreads, Barrier barrier) {
    for (uint32_t d = n / 2; d >= 1; d >>= 1) {
        if (4 * d <= nthreads) {
            uint32_t t = tid >> 2, q = tid & 3;
            if (t < d) { XYZZ<F> a = seg[t]; xyzz_add_quad(a, seg[t + d].p, q); if (q == 0) seg[t] = a; }
        } else if (tid < d) { XYZZ<F> a = seg[tid]; a.add(seg[tid + d]); seg[tid] = a; }
        barrier();
    }
}

// one wavefront per heavy part: 64 lanes stride over <= HEAVY_S entries, then a 6-level LDS tree
// out29 (F = Fq only): the buckets are Bucket29 records (see k_bucket_accum29)
template <class F> ZK_D void put_bucket(XYZZ<F> *buckets, uint32_t gb, const XYZZ<F> &v, int out29) {
    if constexpr (sizeof(F) == sizeof(Fq)) { if (out29) { store_bucket29(reinterpret_cast<Bucket29 *>(buckets) + gb, v); return; } }
    buckets[gb] = v;
}
template <class F>
__global__ __launch_bounds__(256) void k_heavy_parts(const ViewSet<F> views, const uint32_t *sorted, const HeavyItem *items,
                                                      const uint32_t *counters, XYZZ<F> *partials, XYZZ<F> *buckets, SetLayout L, int out29) {
    extern __shared__ unsigned char red_smem[];
    LdsPoint<F> *sh = reinterpret_cast<LdsPoint<F> *>(red_smem);            // 256 points
    const uint32_t set = blockIdx.y;
    const BaseView<F> bases = views.v[set];
    items += set * L.items; partials += set * L.partials; buckets += set * L.buckets;          // (out29: one set per launch, no offset)
    const uint32_t n_items = counters[2 * set], t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (uint32_t base = blockIdx.x * 4; base < n_items; base += gridDim.x * 4) {      // uniform trip count per workgroup
        uint32_t it = base + wv;
        XYZZ<F> acc = XYZZ<F>::inf();
        if (it < n_items) {
            HeavyItem h = items[it];
            for (uint32_t k = h.start + lane; k < h.end; k += 64) {
                uint32_t e = sorted[k];
                Affine<F> p = bases.load(e, h.gb);
                if (e & 1u) p.y = p.y.neg();
                acc.madd(p);
            }
        }
        sh[t] = acc;
        __syncthreads();
        lds_tree_reduce<F>(sh + wv * 64, 64, lane, 64, [] { __syncthreads(); });      // the four wavefronts run their trees in step
        if (lane == 0 && it < n_items) { if (items[it].single) put_bucket<F>(buckets, items[it].gb, sh[t].normalized(), out29); else partials[it] = sh[t].normalized(); }
        __syncthreads();
    }
}

// one workgroup per heavy bucket: strided sum of its parts' partials, LDS tree, write the bucket
template <class F>
__global__ __launch_bounds__(256) void k_heavy_merge(const HeavyBucket *heavy, const uint32_t *counters, const XYZZ<F> *partials, XYZZ<F> *buckets, SetLayout L, int out29, int resume) {
    extern __shared__ unsigned char red_smem[];
    LdsPoint<F> *sh = reinterpret_cast<LdsPoint<F> *>(red_smem);
    const uint32_t set = blockIdx.y;
    heavy += set * L.heavy; partials += set * L.partials; buckets += set * L.buckets;
    const uint32_t n_heavy = counters[2 * set + 1], t = threadIdx.x;
    for (uint32_t hb = blockIdx.x; hb < n_heavy; hb += gridDim.x) {
        HeavyBucket h = heavy[hb];
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t p = t; p < h.nparts; p += 256) acc.add(partials[h.first_item + p]);
        if constexpr (sizeof(F) == sizeof(Fq)) {
            if (resume && out29 && t == 255) acc.add(bucket29_to_xyzz(load_bucket29_raw(reinterpret_cast<const Bucket29 *>(buckets) + h.gb)));   // the earlier pieces' sum
        }
        sh[t] = acc;
        __syncthreads();
        lds_tree_reduce<F>(sh, 256, t, 256, [] { __syncthreads(); });
        if (t == 0) put_bucket<F>(buckets, h.gb, sh[0].normalized(), out29);
        __syncthreads();
    }
}

// ---- 7. bucket reduction: per chunk of RED_CHUNK buckets emit P = sum X_i and U = sum i*X_i (i 0-based in chunk).
//      RedGeom<F>::LANES logical lanes per workgroup, each played by the four lanes of a DPP quad (xyzz_add_quad): the kernel is a
//      dependency chain of 2(L-1) + 2 log2(LANES) + 2 additions, and the quad turns each addition's 14 dependent
//      multiplications into 4 rounds.
template <class F, int RED_L_LOG>
__global__ __launch_bounds__(RedGeom<F>::THREADS) void k_bucket_reduce(const XYZZ<F> *buckets, uint32_t B, uint32_t chunks_per_window, XYZZ<F> *out,
                                                                       size_t in_set_stride, size_t out_set_stride) {
    constexpr int RED_LANES = RedGeom<F>::LANES, RED_L = 1 << RED_L_LOG, RED_CHUNK = RED_LANES * RED_L;
    extern __shared__ unsigned char red_smem[];
    LdsPoint<F> *sh = reinterpret_cast<LdsPoint<F> *>(red_smem);              // 2 * RED_LANES points
    const uint32_t t = threadIdx.x >> 2, q = threadIdx.x & 3;          // logical lane, position in its quad
    const uint32_t w = blockIdx.x / chunks_per_window, ch = blockIdx.x % chunks_per_window;
    const XYZZ<F> *X = buckets + blockIdx.y * in_set_stride + (size_t)w * B;
    out += blockIdx.y * out_set_stride;
    const uint32_t base = ch * RED_CHUNK + t * RED_L;
    // lane-local running sums: S = sum_j X_j, T0 = sum_j j*X_j
    XYZZ<F> run = XYZZ<F>::inf(), T0 = XYZZ<F>::inf();
    XYZZ<F> cur = (base + RED_L - 1 < B) ? X[base + RED_L - 1] : XYZZ<F>::inf();
    for (int j = RED_L - 1; j >= 1; --j) {
        XYZZ<F> nxt = (base + j - 1 < B) ? X[base + j - 1] : XYZZ<F>::inf();      // next bucket arrives under the two additions
        xyzz_add_quad(run, cur, q);
        xyzz_add_quad(T0, run, q);
        cur = nxt;
    }
    xyzz_add_quad(run, cur, q);
    // inclusive suffix scan of S over logical lanes (Hillis-Steele through LDS): Q_t = sum_{u>=t} S_u
    XYZZ<F> Q = run;
    for (uint32_t d = 1; d < RED_LANES; d <<= 1) {
        if (q == 0) sh[t] = Q;
        __syncthreads();
        if (t + d < RED_LANES) xyzz_add_quad(Q, sh[t + d].p, q);
        __syncthreads();
    }
    // sum_t t*S_t = sum_{t>=1} Q_t ; tree-reduce Q (t>=1) in sh[0..), T0 in sh[RED_LANES..)
    XYZZ<F> P = Q;                                                  // logical lane 0: total of the chunk
    if (q == 0) { sh[t] = (t >= 1) ? Q : XYZZ<F>::inf(); sh[RED_LANES + t] = T0; }
    __syncthreads();
    for (uint32_t d = RED_LANES / 2; d >= 1; d >>= 1) {
        if (t < d) { XYZZ<F> a = sh[t]; xyzz_add_quad(a, sh[t + d].p, q); if (q == 0) sh[t] = a; }
        else if (t >= RED_LANES / 2 && t < RED_LANES / 2 + d) {
            uint32_t u = RED_LANES + (t - RED_LANES / 2);
            XYZZ<F> a = sh[u]; xyzz_add_quad(a, sh[u + d].p, q); if (q == 0) sh[u] = a;
        }
        __syncthreads();
    }
    if (t == 0) {
        XYZZ<F> E = sh[0];
        for (int i = 0; i < RED_L_LOG; ++i) E = E.dbl();            // * RED_L
        xyzz_add_quad(E, sh[RED_LANES].p, q);
        if (q == 0) { out[2 * (size_t)blockIdx.x] = P.normalized(); out[2 * (size_t)blockIdx.x + 1] = E.normalized(); }
    }
}

// ---- 7'. the same reduction for G1 on the 29-bit representation (fq29.hip.hpp, xyzz29_add_quad): the kernel is a chain of dependent
//      products at two wavefronts per SIMD, where a 29-bit product takes 930 cycles against 1293.  Buckets arrive as the accumulation wrote
//      them (canonical XYZZ<Fq>) and are converted on load, one coordinate per lane of the quad; the two chunk results leave as XYZZ<Fq>.
struct alignas(16) LdsPoint29 { XYZZ29q p; uint32_t pad[4]; };
template <bool IN29> ZK_D XYZZ29q load_bucket29(const void *X_, uint32_t i, uint32_t B, uint32_t q) {
    if (i >= B) return XYZZ29q::inf();
    if constexpr (IN29) {                                                        // as the accumulation left it: no conversion, every lane of the quad reads the record
        const Bucket29 b = load_bucket29_raw(reinterpret_cast<const Bucket29 *>(X_) + i);
        XYZZ29q r;
#pragma unroll
        for (int j = 0; j < 9; ++j) { r.x.v[j] = b.x[j]; r.y.v[j] = b.y[j]; r.zz.v[j] = b.zz[j]; r.zzz.v[j] = b.zzz[j]; }
        return r;
    } else {
        const XYZZ<Fq> b = reinterpret_cast<const XYZZ<Fq> *>(X_)[i];
        if (b.is_inf()) return XYZZ29q::inf();
        return quad_load29(b.x, b.y, b.zz, b.zzz, q);
    }
}
ZK_D XYZZ<Fq> store_point29(const XYZZ29q &p) {
    if (p.is_inf()) return XYZZ<Fq>::inf().normalized();
    return {f29::from29(p.x), f29::from29(p.y), f29::from29(p.zz), f29::from29(p.zzz)};
}
template <int RED_L_LOG, bool IN29>
__global__ __launch_bounds__(RedGeom<Fq>::THREADS) void k_bucket_reduce29(const void *buckets, uint32_t B, uint32_t chunks_per_window, XYZZ<Fq> *out,
                                                                          size_t in_set_stride, size_t out_set_stride, int critical) {
    crit_wave_priority(critical);
    constexpr int RED_LANES = RedGeom<Fq>::LANES, RED_L = 1 << RED_L_LOG, RED_CHUNK = RED_LANES * RED_L;
    extern __shared__ unsigned char red_smem[];
    LdsPoint29 *sh = reinterpret_cast<LdsPoint29 *>(red_smem);             // 2 * RED_LANES points
    const uint32_t t = threadIdx.x >> 2, q = threadIdx.x & 3;
    const uint32_t w = blockIdx.x / chunks_per_window, ch = blockIdx.x % chunks_per_window;
    const size_t elem = IN29 ? sizeof(Bucket29) : sizeof(XYZZ<Fq>);
    const void *X = reinterpret_cast<const char *>(buckets) + (blockIdx.y * in_set_stride + (size_t)w * B) * elem;
    out += blockIdx.y * out_set_stride;
    const uint32_t base = ch * RED_CHUNK + t * RED_L;
    XYZZ29q run = XYZZ29q::inf(), T0 = XYZZ29q::inf();
    XYZZ29q cur = load_bucket29<IN29>(X, base + RED_L - 1, B, q);
    for (int j = RED_L - 1; j >= 1; --j) {
        const XYZZ29q nxt = load_bucket29<IN29>(X, base + j - 1, B, q);
        xyzz29_add_quad(run, cur, q);
        xyzz29_add_quad(T0, run, q);
        cur = nxt;
    }
    xyzz29_add_quad(run, cur, q);
    XYZZ29q Q = run;                                                     // inclusive suffix scan of the lane sums over logical lanes
    for (uint32_t d = 1; d < RED_LANES; d <<= 1) {
        if (q == 0) sh[t].p = Q;
        __syncthreads();
        if (t + d < RED_LANES) xyzz29_add_quad(Q, sh[t + d].p, q);
        __syncthreads();
    }
    const XYZZ29q P = Q;                                                 // logical lane 0: total of the chunk
    if (q == 0) { sh[t].p = (t >= 1) ? Q : XYZZ29q::inf(); sh[RED_LANES + t].p = T0; }
    __syncthreads();
    for (uint32_t d = RED_LANES / 2; d >= 1; d >>= 1) {
        if (t < d) { XYZZ29q a = sh[t].p; xyzz29_add_quad(a, sh[t + d].p, q); if (q == 0) sh[t].p = a; }
        else if (t >= RED_LANES / 2 && t < RED_LANES / 2 + d) {
            const uint32_t u = RED_LANES + (t - RED_LANES / 2);
            XYZZ29q a = sh[u].p; xyzz29_add_quad(a, sh[u + d].p, q); if (q == 0) sh[u].p = a;
        }
        __syncthreads();
    }
    if (t == 0) {
        XYZZ<Fq> E = store_point29(sh[0].p);
        for (int i = 0; i < RED_L_LOG; ++i) E = E.dbl();                 // * RED_L (RED_L_LOG doublings, once per workgroup: the 32-bit path)
        E.add(store_point29(sh[RED_LANES].p));
        if (q == 0) { out[2 * (size_t)blockIdx.x] = store_point29(P); out[2 * (size_t)blockIdx.x + 1] = E.normalized(); }
    }
}

// ---- 7''. the same reduction with every addition shared by a PAIR of lanes (fq29.hip.hpp xyzz29_add_pair): the quad kernel above is bound by
//      the quad addition's issue (2.8 us per wavefront-addition: 16 lane-products and as many operand moves again for the 14 products an addition
//      needs), a pair spends about half of that per addition at the same depth.  128 logical lanes = 256 threads, one wavefront per SIMD; same
//      chunks, same chain (running sums, suffix scan, two trees), same two results per chunk as k_bucket_reduce29.
ZK_D Half29 lds_load_half29(const LdsPoint29 *p, uint32_t r) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p) + 9 * r;
    Half29 h;
#pragma unroll
    for (int i = 0; i < 9; ++i) { h.c0.v[i] = w[i]; h.c1.v[i] = w[18 + i]; }
    return h;
}
ZK_D void lds_store_half29(LdsPoint29 *p, const Half29 &h, uint32_t r) {
    uint32_t *w = reinterpret_cast<uint32_t *>(p) + 9 * r;
#pragma unroll
    for (int i = 0; i < 9; ++i) { w[i] = h.c0.v[i]; w[18 + i] = h.c1.v[i]; }
}
template <bool IN29> ZK_D Half29 load_half29(const void *X_, uint32_t i, uint32_t B, uint32_t r) {
    if (i >= B) return Half29::inf();
    if constexpr (IN29) {                                                        // this lane's two coordinates of the 29-bit record
        const uint32_t *w = reinterpret_cast<const uint32_t *>(reinterpret_cast<const Bucket29 *>(X_) + i) + 9 * r;
        Half29 h;
#pragma unroll
        for (int j = 0; j < 9; ++j) { h.c0.v[j] = w[j]; h.c1.v[j] = w[18 + j]; }
        return h;
    } else {                                                                     // canonical XYZZ<Fq> (a table launch's folded buckets): two conversions per lane
        const Fq *f = reinterpret_cast<const Fq *>(reinterpret_cast<const XYZZ<Fq> *>(X_) + i);
        const Fq c0 = f[r], c1 = f[2 + r];
        if (c1.is_zero()) return Half29::inf();
        Fq29 to; for (int j = 0; j < 9; ++j) to.v[j] = f29::TO[j];
        Half29 h; f29::mul2(h.c0, h.c1, f29::unpack(c0), to, f29::unpack(c1), to);
        return h;
    }
}
template <int RED_L_LOG, bool IN29>
__global__ __launch_bounds__(2 * RedGeom<Fq>::LANES) void k_bucket_reduce29p(const void *buckets, uint32_t B, uint32_t chunks_per_window, XYZZ<Fq> *out,
                                                                             size_t in_set_stride, size_t out_set_stride, int critical) {
    crit_wave_priority(critical);
    constexpr int RED_LANES = RedGeom<Fq>::LANES, RED_L = 1 << RED_L_LOG, RED_CHUNK = RED_LANES * RED_L;
    extern __shared__ unsigned char red_smem[];
    LdsPoint29 *sh = reinterpret_cast<LdsPoint29 *>(red_smem);             // 2 * RED_LANES points
    const uint32_t t = threadIdx.x >> 1, r = threadIdx.x & 1;             // logical lane, place in its pair
    const uint32_t w = blockIdx.x / chunks_per_window, ch = blockIdx.x % chunks_per_window;
    const size_t elem = IN29 ? sizeof(Bucket29) : sizeof(XYZZ<Fq>);
    const void *X = reinterpret_cast<const char *>(buckets) + (blockIdx.y * in_set_stride + (size_t)w * B) * elem;
    out += blockIdx.y * out_set_stride;
    const uint32_t base = ch * RED_CHUNK + t * RED_L;
    Half29 run = Half29::inf(), T0 = Half29::inf();
    Half29 cur = load_half29<IN29>(X, base + RED_L - 1, B, r);
    for (int j = RED_L - 1; j >= 1; --j) {
        const Half29 nxt = load_half29<IN29>(X, base + j - 1, B, r);     // the next bucket arrives under the two additions
        xyzz29_add_pair(run, cur, r);
        xyzz29_add_pair(T0, run, r);
        cur = nxt;
    }
    xyzz29_add_pair(run, cur, r);
    Half29 Q = run;                                                      // inclusive suffix scan of the lane sums over logical lanes
    for (uint32_t d = 1; d < RED_LANES; d <<= 1) {
        lds_store_half29(sh + t, Q, r);
        __syncthreads();
        if (t + d < RED_LANES) xyzz29_add_pair(Q, lds_load_half29(sh + t + d, r), r);
        __syncthreads();
    }
    const Half29 P = Q;                                                  // logical lane 0: total of the chunk
    lds_store_half29(sh + t, t >= 1 ? Q : Half29::inf(), r); lds_store_half29(sh + RED_LANES + t, T0, r);
    __syncthreads();
    for (uint32_t d = RED_LANES / 2; d >= 1; d >>= 1) {
        if (t < d) { Half29 a = lds_load_half29(sh + t, r); xyzz29_add_pair(a, lds_load_half29(sh + t + d, r), r); lds_store_half29(sh + t, a, r); }
        else if (t >= RED_LANES / 2 && t < RED_LANES / 2 + d) {
            const uint32_t u = RED_LANES + (t - RED_LANES / 2);
            Half29 a = lds_load_half29(sh + u, r); xyzz29_add_pair(a, lds_load_half29(sh + u + d, r), r); lds_store_half29(sh + u, a, r);
        }
        __syncthreads();
    }
    if (t == 0) lds_store_half29(sh + 1, P, r);                          // (slot 1 is free once the trees are through)
    __syncthreads();
    if (threadIdx.x == 0) {
        XYZZ<Fq> E = store_point29(sh[0].p);
        for (int i = 0; i < RED_L_LOG; ++i) E = E.dbl();                 // * RED_L (RED_L_LOG doublings, once per workgroup: the 32-bit path)
        E.add(store_point29(sh[RED_LANES].p));
        out[2 * (size_t)blockIdx.x] = store_point29(sh[1].p); out[2 * (size_t)blockIdx.x + 1] = E.normalized();
    }
}

// ---- 7''. the large reduction with its first phase on single lanes.  k_bucket_reduce29p walks a logical lane's 16 buckets as 31 pair
//      additions in sequence — two thirds of its chain, at 126 ns of a SIMD per addition.  Here each of the workgroup's 256 THREADS takes 8
//      buckets alone (xyzz29_add_lane: 85 ns of a SIMD per addition, 64 additions per wavefront-step): 15 steps of 5.6 us instead of 31 of
//      4.7.  The 256 (sum, weighted sum) results go through LDS to the 128 pairs, two entries each, for the same scan and trees:
//          sum_b b B_b (b local, 8 i + j)  =  8 sum_i i R_i + sum_i T_i,      sum_i i R_i = sum_{i >= 1} (suffix sum of R at i),
//      a pair owning entries 2t, 2t + 1 adds four pair additions to the old chain (their sum for the scan, the odd entry's suffix, the
//      pair's two suffixes, the pair's two weighted sums).  Same chunks as the pair kernel; three results per chunk (see the end).
//      Bucket29 input only (the plain G1 path); the other shapes keep the pair / quad kernels.
__global__ __launch_bounds__(2 * RedGeom<Fq>::LANES) void k_bucket_reduce29l(const Bucket29 *buckets, uint32_t B, uint32_t chunks_per_window, XYZZ<Fq> *out,
                                                                             size_t in_set_stride, size_t out_set_stride, int critical) {
    crit_wave_priority(critical);
    constexpr int RED_LANES = RedGeom<Fq>::LANES, NT = 2 * RED_LANES, PER = 8, RED_CHUNK = NT * PER;      // = RED_LANES << RED_L_LOG_LARGE
    static_assert(RED_CHUNK == (RED_LANES << RED_L_LOG_LARGE), "the lane-form reduction keeps the large reduction's chunks");
    extern __shared__ unsigned char red_smem[];
    LdsPoint29 *sh = reinterpret_cast<LdsPoint29 *>(red_smem);             // 2 * NT points
    const uint32_t i = threadIdx.x;
    const uint32_t w = blockIdx.x / chunks_per_window, ch = blockIdx.x % chunks_per_window;
    const Bucket29 *X = buckets + blockIdx.y * in_set_stride + (size_t)w * B;
    out += blockIdx.y * out_set_stride;
    {   // phase 1: this thread's 8 buckets — run = their sum, T0 = sum_j j B_(base + j)
        const uint32_t base = ch * RED_CHUNK + i * PER;
        XYZZ29q run = XYZZ29q::inf(), T0 = XYZZ29q::inf();
        XYZZ29q cur = load_bucket29<true>(X, base + PER - 1, B, 0);
        for (int j = PER - 1; j >= 1; --j) {
            const XYZZ29q nxt = load_bucket29<true>(X, base + j - 1, B, 0);  // the next bucket arrives under the two additions
            xyzz29_add_lane(run, cur);
            xyzz29_add_lane(T0, run);
            cur = nxt;
        }
        xyzz29_add_lane(run, cur);
        sh[i].p = run; sh[NT + i].p = T0;
    }
    __syncthreads();
    // phase 2: pairs.  Pair t owns entries 2t and 2t + 1.
    const uint32_t t = i >> 1, r = i & 1;
    const Half29 R1 = lds_load_half29(sh + 2 * t + 1, r);
    Half29 Q = lds_load_half29(sh + 2 * t, r);
    xyzz29_add_pair(Q, R1, r);                                           // the pair's sum
    Half29 Tp = lds_load_half29(sh + NT + 2 * t, r);
    xyzz29_add_pair(Tp, lds_load_half29(sh + NT + 2 * t + 1, r), r);     // the pair's weighted sums
    __syncthreads();                                                     // (every entry has been read: the scan reuses the first RED_LANES slots)
    for (uint32_t d = 1; d < RED_LANES; d <<= 1) {                       // inclusive suffix scan of the pair sums
        lds_store_half29(sh + t, Q, r);
        __syncthreads();
        if (t + d < RED_LANES) xyzz29_add_pair(Q, lds_load_half29(sh + t + d, r), r);
        __syncthreads();
    }
    const Half29 P = Q;                                                  // pair 0: total of the chunk
    lds_store_half29(sh + t, Q, r);
    __syncthreads();
    Half29 V = t + 1 < RED_LANES ? lds_load_half29(sh + t + 1, r) : Half29::inf();     // the sum of everything above this pair
    xyzz29_add_pair(V, R1, r);                                           // suffix at entry 2t + 1
    if (t >= 1) xyzz29_add_pair(V, Q, r);                                // + suffix at entry 2t (entry 0 weighs nothing)
    __syncthreads();
    lds_store_half29(sh + t, V, r); lds_store_half29(sh + RED_LANES + t, Tp, r);
    __syncthreads();
    for (uint32_t d = RED_LANES / 2; d >= 1; d >>= 1) {                  // two trees side by side, as in the pair kernel
        if (t < d) { Half29 a = lds_load_half29(sh + t, r); xyzz29_add_pair(a, lds_load_half29(sh + t + d, r), r); lds_store_half29(sh + t, a, r); }
        else if (t >= RED_LANES / 2 && t < RED_LANES / 2 + d) {
            const uint32_t u = RED_LANES + (t - RED_LANES / 2);
            Half29 a = lds_load_half29(sh + u, r); xyzz29_add_pair(a, lds_load_half29(sh + u + d, r), r); lds_store_half29(sh + u, a, r);
        }
        __syncthreads();
    }
    if (t == 0) lds_store_half29(sh + 1, P, r);
    __syncthreads();
    // three results per chunk: the sum, sum_i T_i, sum_i i R_i.  The host adds the windows' chunks up anyway and multiplies the third by 8 (buckets
    // per thread) once per window: three doublings and an addition on ONE lane's 32-bit path here were 25 us at the end of every workgroup's chain.
    if (threadIdx.x < 3) out[3 * (size_t)blockIdx.x + threadIdx.x] = store_point29(threadIdx.x == 0 ? sh[1].p : threadIdx.x == 1 ? sh[RED_LANES].p : sh[0].p);
}


// ---- 7b'. the fold of a table launch's bucket rows on the 29-bit records: a pair of lanes per bucket adds its W rows in sequence (xyzz29_add_pair; rows
//      without entries were never written and are skipped by their length), Bucket29 in, Bucket29 out — the accumulation stores its accumulators as they
//      are and the reduction loads the folded set as it is: no conversion anywhere between the gathers and the chunk results.
__global__ __launch_bounds__(256) void k_bucket_fold29(const Bucket29 *buckets, const uint32_t *counts, uint32_t W, uint32_t B, Bucket29 *out, int critical) {
    crit_wave_priority(critical);
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 1, r = threadIdx.x & 1;
    if (b >= B) return;                                                         // (B is even: pairs stay whole)
    Half29 acc = Half29::inf();
    uint32_t w = 0;
    while (w < W && !counts[(size_t)w * B + b]) ++w;
    Half29 cur = w < W ? load_half29<true>(buckets + (size_t)w * B, b, B, r) : Half29::inf();
    while (w < W) {
        uint32_t nw = w + 1;
        while (nw < W && !counts[(size_t)nw * B + b]) ++nw;
        const Half29 nxt = nw < W ? load_half29<true>(buckets + (size_t)nw * B, b, B, r) : Half29::inf();   // the next row arrives under the addition
        xyzz29_add_pair(acc, cur, r);
        cur = nxt; w = nw;
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(out + b) + 9 * r;
#pragma unroll
    for (int i = 0; i < 9; ++i) { o[i] = acc.c0.v[i]; o[18 + i] = acc.c1.v[i]; }
}

// ---- 7b. per-window tables: all W windows weigh the same, so their bucket sets are summed bucket-wise before the one reduction.
//      A workgroup takes FOLD_B buckets: its 256 threads load one accumulator each (window-major, so a wavefront reads FOLD_B
//      consecutive buckets of one window), then an LDS tree over the windows — ceil(log2 W) levels instead of a W-long chain, the upper
//      levels shared by DPP quads.  Empty accumulators (a sparse witness) cost nothing: additions with infinity return at once.
static constexpr uint32_t FOLD_THREADS = 256;
template <class F>
__global__ __launch_bounds__(FOLD_THREADS) void k_bucket_fold(const XYZZ<F> *buckets, const uint32_t *counts, uint32_t W, uint32_t B, uint32_t fold_b_log, XYZZ<F> *out, SetLayout L) {
    extern __shared__ unsigned char red_smem[];
    LdsPoint<F> *sh = reinterpret_cast<LdsPoint<F> *>(red_smem);            // FOLD_THREADS points: [window slot][bucket]
    const uint32_t t = threadIdx.x, FB = 1u << fold_b_log, slots = FOLD_THREADS >> fold_b_log;     // window slots held at once (a power of two)
    const uint32_t bl = t & (FB - 1), slot = t >> fold_b_log, b = blockIdx.x * FB + bl;
    buckets += blockIdx.y * L.buckets; out += blockIdx.y * L.folded;
    XYZZ<F> acc = XYZZ<F>::inf();
    // buckets without entries were never written (the accumulation is launched over the non-empty ones only): their length says so
    if (b < B) for (uint32_t w = slot; w < W; w += slots) if (counts[(size_t)w * B + b]) acc.add(buckets[(size_t)w * B + b]);   // W <= slots for the shipped window sizes: no addition
    sh[t] = acc;
    __syncthreads();
    for (uint32_t d = slots / 2; d >= 1; d >>= 1) {                    // tree over the window slots; pairs (slot, slot + d) of the same bucket
        const uint32_t pairs = d << fold_b_log;
        if (4 * pairs <= FOLD_THREADS) {
            const uint32_t u = t >> 2, q = t & 3;
            if (u < pairs) { XYZZ<F> a = sh[u]; xyzz_add_quad(a, sh[u + pairs].p, q); if (q == 0) sh[u] = a; }
        } else if (t < pairs) { XYZZ<F> a = sh[t]; a.add(sh[t + pairs]); sh[t] = a; }
        __syncthreads();
    }
    if (t < FB && b < B) out[b] = sh[t].normalized();
}

// 1 / x for the in-lane batched normalisations below.  Fq: through the 29-bit safegcd (f29::inverse: ~70 products' worth against the ~380 of the
// Fermat chain behind Fq::inverse(); two conversions); Fq2: the same on its norm.
template <class F> ZK_D F inverse_fast(const F &x) {
    if constexpr (std::is_same<F, Fq>::value) return f29::from29(f29::inverse(f29::to29(x.normalized())));
    else { const Fq d = inverse_fast<Fq>(x.c0.sqr() + x.c1.sqr()); return F{x.c0 * d, (x.c1 * d).neg()}; }      // (c0 - c1 u) / (c0^2 + c1^2), as Fq2::inverse()
}
// levels j+1 .. j+K of a window table from level j: level j+l = 2^(c l) * level j.  One lane walks a point through its K x c doublings in
// XYZZ and normalises the K results with ONE inversion (Montgomery's trick over their ZZZ): a level per launch spent 80 % of its
// time in the 380-product Fermat inversion of each point.  out points at level j+1; levels are n entries apart.
template <class F, int K>
__global__ __launch_bounds__(256) void k_table_levels(const Affine<F> *in, Affine<F> *out, size_t n, uint32_t c, int levels) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> a = in[i];
    if (a.is_inf()) { for (int l = 0; l < levels; ++l) out[(size_t)l * n + i] = a; return; }
    F x[K], y[K], zz[K], zzz[K];
    XYZZ<F> p = XYZZ<F>::dbl_affine_inl(a);
#pragma unroll
    for (int l = 0; l < K; ++l) {
        if (l < levels) {
            for (uint32_t k = (l == 0 ? 1u : 0u); k < c; ++k) p = p.dbl();
            x[l] = p.x; y[l] = p.y; zz[l] = p.zz; zzz[l] = p.zzz;
        } else { x[l] = F::zero(); y[l] = F::zero(); zz[l] = F::zero(); zzz[l] = F::zero(); }
    }
    // (a point of a prime-order group never doubles to infinity; a hostile key's small-order G2 point can: ZZZ = 0 is skipped and restored)
    F pre[K], run = F::one();
#pragma unroll
    for (int l = 0; l < K; ++l) { pre[l] = run; if (!zzz[l].is_zero()) run = run * zzz[l]; }
    F inv = inverse_fast(run);
#pragma unroll
    for (int l = K - 1; l >= 0; --l) {
        if (l >= levels) continue;
        if (zzz[l].is_zero()) { out[(size_t)l * n + i] = Affine<F>::inf(); continue; }
        F zi = inv * pre[l];                                                    // 1 / ZZZ_l
        inv = inv * zzz[l];
        F t = zi * zz[l];                                                       // ZZ / ZZZ = 1 / Z
        F zi2 = t.sqr();                                                        // 1 / ZZ
        out[(size_t)l * n + i] = Affine<F>{x[l] * zi2, y[l] * zi}.normalized();
    }
}

// ---- witness split (libff multi_exp_with_mixed_addition, reached from snark.cpp:126 for the A / B / L queries): a scalar that is 0 is
//      skipped, a scalar that is 1 adds its base directly, anything else is left to the bucket method.  k_classify tags every element
//      of z = [1 | w] (0 zero, 1 one, 2 other) and lists the indices of the others; k_ones_sum adds the bases tagged 1 (a flat sum: lanes
//      stride over the tags, LDS tree per workgroup, k_sum_partials finishes); the listed ones go through the digit sort as a gathered
//      subset.  Both parts are exact group sums, so their total is the same point the reference's loop produces.
__global__ __launch_bounds__(256) void k_classify(const Fr *z, size_t n1, uint8_t *tags, uint32_t *listed, uint32_t *count, const uint32_t *subset_pos) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tag = 0;
    if (i < n1) {
        Fr v = z[i];
        uint32_t any = 0, diff = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { any |= v.v[j]; diff |= v.v[j] ^ FrParams::ONE[j]; }
        tag = any == 0 ? 0u : (diff == 0 ? 1u : 2u);
        tags[i] = (uint8_t)tag;
    }
    const unsigned long long mask = __ballot(tag == 2);
    if (!mask) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((long long)mask) - 1)) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1, 64);
    if (tag == 2) {
        listed[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = (uint32_t)i;
        if (subset_pos && subset_pos[i] == SUBSET_NONE) count[1] = 1;        // the witness tables do not cover this element (yet)
    }
}
template <class F>
__global__ __launch_bounds__(256) void k_gather_points(const Affine<F> *src, const uint32_t *idx, size_t count, uint32_t index_sub, Affine<F> *out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint32_t i = idx[j];
    out[j] = i >= index_sub ? src[i - index_sub] : Affine<F>::inf();
}
template <class F>
__global__ __launch_bounds__(256) void k_scatter_points(const Affine<F> *src, const uint32_t *idx, size_t count, Affine<F> *out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) out[idx[j]] = src[j];
}
// A workgroup owns `span` consecutive tags.  Its four wavefronts first list the positions tagged 1 in LDS (ballot + prefix count, a segment
// per wavefront: the order is the index order, nothing depends on timing), then all 256 lanes stride over the list — every lane of every
// addition is a real one.  (Round 2 strode over the tags themselves: with half the witness bits zero half of each wavefront's additions were
// masked off, the kernel's time was twice its work.)
template <class F>
__global__ __launch_bounds__(256) void k_ones_sum(const ViewSet<F> views, const uint8_t *tags, size_t n1, uint32_t span, XYZZ<F> *partials) {
    extern __shared__ unsigned char red_smem[];
    LdsPoint<F> *sh = reinterpret_cast<LdsPoint<F> *>(red_smem);            // 256 points
    uint32_t *list = reinterpret_cast<uint32_t *>(red_smem + 256 * sizeof(LdsPoint<F>));        // span positions, span / 4 per wavefront
    __shared__ uint32_t seg_count[4];
    const Affine<F> *bases = views.v[blockIdx.y].p; const uint32_t index_sub = views.v[blockIdx.y].index_sub;
    partials += (size_t)blockIdx.y * (gridDim.x + 1);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, cap = span >> 2;
    const size_t first = (size_t)blockIdx.x * span + (size_t)wv * cap;
    uint32_t count = 0;                                                       // uniform across the wavefront
    for (uint32_t o = 0; o < cap; o += 64) {
        const size_t i = first + o + lane;
        const bool one = i < n1 && i >= index_sub && tags[i] == 1;
        const unsigned long long mask = __ballot(one);
        if (one) list[wv * cap + count + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = (uint32_t)(i - index_sub);
        count += (uint32_t)__popcll(mask);
    }
    if (lane == 0) seg_count[wv] = count;
    __syncthreads();
    const uint32_t c0 = seg_count[0], c1 = c0 + seg_count[1], c2 = c1 + seg_count[2], total = c2 + seg_count[3];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = tid; k < total; k += 256) {
        const uint32_t seg = (k >= c0) + (k >= c1) + (k >= c2), off = seg == 0 ? 0u : (seg == 1 ? c0 : (seg == 2 ? c1 : c2));
        acc.madd(bases[list[seg * cap + (k - off)]]);
    }
    sh[tid] = acc;
    __syncthreads();
    lds_tree_reduce<F>(sh, 256, tid, 256, [] { __syncthreads(); });
    if (tid == 0) partials[blockIdx.x] = sh[0].normalized();
}
template <class F>
__global__ __launch_bounds__(256) void k_sum_partials(XYZZ<F> *partials_all, uint32_t count) {        // per set: partials[0..count) -> partials[count]
    extern __shared__ unsigned char red_smem[];
    LdsPoint<F> *sh = reinterpret_cast<LdsPoint<F> *>(red_smem);
    XYZZ<F> *partials = partials_all + (size_t)blockIdx.x * (count + 1), *out = partials + count;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t i = threadIdx.x; i < count; i += 256) acc.add(partials[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    lds_tree_reduce<F>(sh, 256, threadIdx.x, 256, [] { __syncthreads(); });
    if (threadIdx.x == 0) out[0] = sh[0].normalized();
}

// ---- jobs: one MSM (or several base sets over one scalar vector) in flight on one stream ---------------------
struct MsmGroup {                   // the base sets of one field in a launch: accumulators and the host landing zone of their chunk results
    DevBuf buckets, folded, red_out, heavy_items, heavy_buckets, heavy_counters, heavy_partials;
    DevBuf bases29;                                    // a plain G1 set's bases as 64-byte packed 29-bit records (Rec64, k_bases_to29), rebuilt per launch
    DevBuf ba_start, ba_pre, ba_planes;                // batched-affine levels (msm_ba.inc): bucket-start bits, prefix-product scratch, the levels' coordinate planes
    void *host_red = nullptr; size_t host_cap = 0; bool g2 = false, table = false; int nsets = 0;
    int out_index[MSM_MAX_SETS] = {0};                 // position of each set among the launch's sets of this field
    int red_slots = 2; size_t red_stride = 0;                                // results per chunk (k_bucket_reduce29l: 3, see there), elements per set in red_out
    uint32_t cpw = 0, red_windows = 0; size_t nred = 0; int chunk_log = 0;   // reduce geometry per set: chunks per window, windows reduced, chunk results, log2(buckets per chunk)
    int host_reserve(size_t bytes) {
        if (bytes <= host_cap) return 0;
        if (host_red) (void)hipHostFree(host_red);
        host_red = nullptr; host_cap = 0;
        if (!hip_ok(hipHostMalloc(&host_red, bytes + 4096, hipHostMallocDefault), "hipHostMalloc", __FILE__, __LINE__)) return 1;
        host_cap = bytes + 4096;
        return 0;
    }
    void release() {
        for (DevBuf *b : {&buckets, &folded, &red_out, &heavy_items, &heavy_buckets, &heavy_counters, &heavy_partials, &bases29, &ba_start, &ba_pre, &ba_planes}) b->release();
        if (host_red) (void)hipHostFree(host_red);
        host_red = nullptr; host_cap = 0;
    }
};
struct MsmJob {
    hipStream_t stream = nullptr; bool own_stream = false;
    DevBuf digits, hist, counts, offsets, scan_sums, class_hist, order, sorted, rx_tmp, rx_meta;
    MsmGroup group[2];                 // [0] the G1 sets, [1] the G2 sets of the launch
    MsmGeom g{}; size_t n = 0;
    int window_hint = 0;               // 0: pick_geom's rule
    // Table launches: every window's digit weighs the same, so f consecutive windows can share one row of buckets — a row is then a window
    // over the f n entries (level, point) of its levels, which lie back to back in the table.  Same sorted list (rows of the digit array
    // are simply read in pairs), W / f x B accumulators instead of W x B: the fold's additions shrink with it and chains get f times longer.
    uint32_t merge_hint = 1, merge = 1;
    // a plain G1 set's conversion to 29-bit records (k_bases_to29) runs on a stream of its own beside the digit sort, whose small launches
    // leave most of the chip idle: fork after everything queued before (the previous launch's accumulation still reads the records), join before the accumulation
    hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool converted_aside = false;
    bool one_pass_sort = false;        // the caller knows the digits are skewed (a prover's 0/1 witness): skip the two-pass sort's attempt
    // piece-wise jobs (msm_g1_host_scalars): `resume` — this launch's accumulation continues the buckets of the launch before it;
    // `defer_reduce` — more pieces follow: no reduction, nothing copied back; `c_fixed` — every piece uses the whole job's window size
    bool resume = false, defer_reduce = false; bool last_out29 = false;
    hipEvent_t ev_tail = nullptr; bool tail_recorded = false;      // recorded behind the job's last accumulation (see msm_job_finish)
    bool critical = false;                        // msm_job_set_critical: the accumulate / fold / reduce kernels raise their wavefronts' issue priority
    // a second job that alternates with this one over the pieces of one multi-exponentiation accumulates into THIS job's buckets
    MsmJob *bucket_owner = nullptr;
    // the digit sort may run on a stream of its own (high priority: its small kernels get the compute units an accumulation frees first);
    // ev_sorted orders the accumulation behind it, ev_acc_done the next sort of this job's buffers behind the accumulation that reads them
    hipStream_t sort_stream = nullptr; hipEvent_t ev_sorted = nullptr, ev_acc_done = nullptr;
    uint32_t w0 = 0, ws = 1;           // window subset of the next launches (window-sharded multi-GPU runs)
    bool empty = false;
    // bases resident, scalars in host memory (msm_g1_host_scalars): device copy of the scalars, the copy stream and one event per piece
    DevBuf hs_scalars; hipStream_t copy = nullptr; hipEvent_t ev_piece[8] = {nullptr};
    std::mutex mu;
};

template <class F>
static int launch_accumulate(MsmJob *job, MsmGroup &gr, const MsmBases *sets, const uint32_t *d_gather, bool time_it) {
    const MsmGeom g = job->g; const size_t n = job->n, total_buckets = (size_t)g.W * g.B; const unsigned ns = (unsigned)gr.nsets;
    hipStream_t s = job->stream;
    size_t n_entries_max = n * g.W;
    // worst case of the device-side threshold: a heavy bucket holds more than 8 entries (G1; 4 for G2), so there are fewer than
    // entries / 8 (entries / 4) of them and never more than there are buckets; each is cut into ceil(len / 512) parts
    const size_t max_heavy = std::min(total_buckets, n_entries_max / (sizeof(F) != sizeof(Fq) ? 4 : 8)) + 1,
                 max_items = max_heavy + n_entries_max / HEAVY_S + 1;
    typedef RedGeom<F> RG;
    gr.table = sets[0].level_stride != 0;
    gr.red_windows = gr.table ? 1 : g.W;                             // a table's windows are folded into one bucket set first
    const size_t red_buckets = (size_t)gr.red_windows * g.B;
    static const char *force_l = getenv("ZKG_RED_L_LOG");                                                  // tuning aid: 0, 2 or 4
    const int red_l_log = force_l ? atoi(force_l) : red_buckets >= RED_LARGE_BUCKETS ? RED_L_LOG_LARGE : red_buckets > RED_SMALL_BUCKETS ? RED_L_LOG_SMALL : RED_L_LOG_TINY;
    gr.chunk_log = RG::LANES_LOG + red_l_log;
    gr.cpw = (g.B + (1u << gr.chunk_log) - 1) >> gr.chunk_log; gr.nred = (size_t)gr.red_windows * gr.cpw;
    DevBuf &bucket_buf = job->bucket_owner ? job->bucket_owner->group[sizeof(F) != sizeof(Fq) ? 1 : 0].buckets : gr.buckets;
    gr.red_slots = 2;
    SetLayout L; L.buckets = total_buckets; L.items = max_items; L.heavy = max_heavy; L.partials = max_items; L.folded = g.B; L.red_out = gr.nred * 3;      // two results per chunk, three from k_bucket_reduce29l (gr.red_slots)
    if (gr.heavy_items.reserve(ns * max_items * sizeof(HeavyItem)) || gr.heavy_buckets.reserve(ns * max_heavy * sizeof(HeavyBucket)) ||
        gr.heavy_counters.reserve(8 * MSM_MAX_SETS) || gr.heavy_partials.reserve(ns * max_items * sizeof(XYZZ<F>)) ||
        bucket_buf.reserve(ns * total_buckets * std::max(sizeof(XYZZ<F>), sizeof(Bucket29))) || gr.red_out.reserve(ns * L.red_out * sizeof(XYZZ<F>)) || ((gr.red_stride = L.red_out), false) ||
        (gr.table && gr.folded.reserve(ns * (size_t)g.B * std::max(sizeof(XYZZ<F>), sizeof(Bucket29)))) || gr.host_reserve(ns * L.red_out * sizeof(XYZZ<F>))) return ZKG_ERROR;
    // (gr.heavy_counters: cleared by the job's k_digits)
    XYZZ<F> *buckets = bucket_buf.as<XYZZ<F>>();
    ViewSet<F> views;
    for (unsigned i = 0; i < (unsigned)MSM_MAX_SETS; ++i) {
        const MsmBases &b = sets[i < ns ? i : 0];
        views.v[i] = BaseView<F>{reinterpret_cast<const Affine<F> *>(b.p), b.level_stride, d_gather, b.index_sub, g.B, b.remap};
    }
    // `order` lists the buckets by descending length, the empty ones last.  A table launch folds through the bucket lengths and never reads
    // an empty bucket, so its accumulation covers at most one lane per entry: a witness' ~10^5 entries over 2^19 buckets would otherwise
    // dispatch 8 K wavefronts that find nothing to do — on a chip they share with the H multi-exponentiation.
    const size_t lanes = gr.table ? std::min(total_buckets, n_entries_max) : total_buckets;
    bool plain = !d_gather;
    for (unsigned i = 0; i < ns; ++i) plain = plain && sets[i].level_stride == 0 && sets[i].index_sub == 0 && !sets[i].remap;
    // a plain G1 set takes the 29-bit accumulation (ZKG_ACCUM_32: the 8 x 32-bit kernel, kept for A/B runs and for several sets per launch)
    static const bool accum32 = getenv("ZKG_ACCUM_32") != nullptr;
    bool use29 = false; const Rec64 *rec29 = nullptr; size_t stride29 = 0;
    if constexpr (sizeof(F) == sizeof(Fq)) {
        if (ns == 1 && !accum32 && plain) {
            use29 = true;
            if (sets[0].p29) rec29 = reinterpret_cast<const Rec64 *>(sets[0].p29);       // the caller made the records (a piece of a piece-wise job)
            else {
                if (job->converted_aside) ZK_HIP(hipStreamWaitEvent(s, job->ev_join, 0));   // converted beside the sort (msm_job_launch)
                else {
                    if (gr.bases29.reserve(n * sizeof(Rec64))) return ZKG_ERROR;
                    hipLaunchKernelGGL(k_bases_to29, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const Affine<Fq> *>(sets[0].p), n, gr.bases29.as<Rec64>());
                }
                rec29 = gr.bases29.as<Rec64>();
            }
        } else if (ns == 1 && !accum32 && !d_gather && sets[0].p29 && !sets[0].remap && sets[0].index_sub == 0) {
            use29 = true; rec29 = reinterpret_cast<const Rec64 *>(sets[0].p29); stride29 = sets[0].level_stride;      // a resident table's records
        }
    }
    static const bool red32_env = getenv("ZKG_REDUCE_32") != nullptr;
    // plain G1 set on the 29-bit kernels: the buckets stay 29-bit records from the accumulation to the reduction
    static const bool fold32_env = getenv("ZKG_FOLD_32") != nullptr;                                       // A/B switch: table launches keep canonical XYZZ buckets and the 32-bit fold
    // ... and so do a lone table set's (the prover's H query: accumulate -> k_bucket_fold29 -> reduce, all on the records)
    // — where the rows are few (8 after the row merge of the large tables: a bucket's rows are added in sequence, 4 us each; a one-payload key's 22
    // rows of 2048 buckets keep the 32-bit fold, a tree over the rows: 0.75 against 0.83 ms per proof)
    const bool out29 = use29 && !red32_env && (stride29 == 0 ? !gr.table : (gr.table && ns == 1 && g.W <= 8 && !fold32_env));
    const int resume = job->resume ? 1 : 0;
    MsmJob *owner = job->bucket_owner ? job->bucket_owner : job;                // (whose buckets these are)
    if (resume && !(out29 && owner->last_out29)) { set_error("msm: a piece can only continue 29-bit buckets"); return ZKG_ERROR; }
    owner->last_out29 = out29;
    if (time_it) g_dominant_timer.begin(s);
    bool done_ba = false;
    if constexpr (sizeof(F) == sizeof(Fq)) {
        // batched-affine levels in front of the accumulation (msm_ba.inc): ZKG_ACCUM_BA = number of levels (1..4), ZKG_BA_K = nodes per lane
        // (read per launch, not once: the parity tests switch it inside one process)
        const char *e_ba = getenv("ZKG_ACCUM_BA"), *e_k = getenv("ZKG_BA_K");
        const int ba_levels = e_ba ? std::max(0, std::min((int)ba::MAX_LEVELS, atoi(e_ba))) : 0;
        const int ba_k = e_k ? std::max(2, std::min(32, atoi(e_k) & ~1)) : 16;
        if (use29 && out29 && stride29 == 0 && !resume && ba_levels > 0 && n_entries_max >= 4096) {
            const int R = ba_levels, K = ba_k;
            const size_t per_wg = (size_t)ba::THREADS * K, words_start = n_entries_max / 32 + 2;
            size_t stride[ba::MAX_LEVELS + 1] = {0}, plane_words = 0, grid1 = 0;
            for (int l = 1; l <= R; ++l) {
                const size_t nodes = (n_entries_max >> l) + 1, grid = (nodes + per_wg - 1) / per_wg;
                stride[l] = grid * per_wg;                                     // a multiple of 512: every lane's node index stays inside the plane
                plane_words += 18 * stride[l];
                if (l == 1) grid1 = grid;
            }
            if (gr.ba_start.reserve(words_start * 4) || gr.ba_pre.reserve(grid1 * K * 9 * ba::THREADS * 4) || gr.ba_planes.reserve(plane_words * 4)) return ZKG_ERROR;
            uint32_t *start = gr.ba_start.as<uint32_t>(), *planes = gr.ba_planes.as<uint32_t>();
            ZK_HIP(hipMemsetAsync(start, 0, words_start * 4, s));
            hipLaunchKernelGGL(k_ba_starts, dim3((unsigned)((total_buckets + 255) / 256)), dim3(256), 0, s, job->counts.as<uint32_t>(), job->offsets.as<uint32_t>(), total_buckets, start);
            BaPlanes pl{}; size_t at = 0;
            for (int l = 1; l <= R; ++l) { pl.X[l] = planes + at; pl.Y[l] = planes + at + 9 * stride[l]; pl.stride[l] = stride[l]; at += 18 * stride[l]; }
            const uint32_t *n_entries = job->offsets.as<uint32_t>() + total_buckets;
            for (int l = 1; l <= R; ++l) {
                const unsigned grid = (unsigned)(stride[l] / per_wg);
                uint32_t *oX = const_cast<uint32_t *>(pl.X[l]), *oY = const_cast<uint32_t *>(pl.Y[l]);
                if (l == 1) {
                    ba::Src src{rec29, job->sorted.as<uint32_t>(), nullptr, nullptr, 0};
                    hipLaunchKernelGGL(k_ba_level<true>, dim3(grid), dim3(ba::THREADS), 0, s, src, start, n_entries, (uint32_t)l, K, oX, oY, stride[l], gr.ba_pre.as<uint32_t>());
                } else {
                    ba::Src src{nullptr, nullptr, pl.X[l - 1], pl.Y[l - 1], stride[l - 1]};
                    hipLaunchKernelGGL(k_ba_level<false>, dim3(grid), dim3(ba::THREADS), 0, s, src, start, n_entries, (uint32_t)l, K, oX, oY, stride[l], gr.ba_pre.as<uint32_t>());
                }
            }
            hipLaunchKernelGGL(k_bucket_accum_ba, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, rec29, job->sorted.as<uint32_t>(), pl, R, job->offsets.as<uint32_t>(),
                               job->order.as<uint32_t>(), lanes, reinterpret_cast<Bucket29 *>(buckets), gr.heavy_items.as<HeavyItem>(), gr.heavy_buckets.as<HeavyBucket>(),
                               gr.heavy_counters.as<uint32_t>(), L);
            done_ba = true;
        }
    }
    if (done_ba) {}
    else if constexpr (sizeof(F) == sizeof(Fq)) {
        // wavefronts per SIMD: three (168 registers; the loop has no spill either way) when the launch has the chip to itself — a plain set sorted on
        // the job's own stream: the resident call, 1.17 -> 1.14 ms alone — and two capped at ACC29_VGPRS where something is meant to run beside it: the
        // next piece's sort of a piece-wise job, the witness multi-exponentiations of a proof (table launches)
        static const int waves_env = getenv("ZKG_ACC29_WAVES") ? atoi(getenv("ZKG_ACC29_WAVES")) : 0;               // tuning aid: 2 or 3 everywhere
        const int waves29 = waves_env ? waves_env : (job->sort_stream || gr.table || job->bucket_owner) ? 2 : 3;
        auto launch29 = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, rec29, stride29, g.B, job->sorted.as<uint32_t>(),
                               job->offsets.as<uint32_t>(), job->order.as<uint32_t>(), lanes, (void *)buckets, gr.heavy_items.as<HeavyItem>(),
                               gr.heavy_buckets.as<HeavyBucket>(), gr.heavy_counters.as<uint32_t>(), L, resume | (job->critical ? 2 : 0));
        };
        if (use29 && waves29 == 3) { if (out29) launch29(k_bucket_accum29<3, true>); else launch29(k_bucket_accum29<3, false>); }
        else if (use29) { if (out29) launch29(k_bucket_accum29<2, true>); else launch29(k_bucket_accum29<2, false>); }
    }
    if (use29 || done_ba) {}
    else if (plain)
        hipLaunchKernelGGL((k_bucket_accum<F, true>), dim3((unsigned)((lanes + 255) / 256), ns), dim3(256), 0, s,
                           views, job->sorted.as<uint32_t>(), job->offsets.as<uint32_t>(), job->order.as<uint32_t>(), lanes, buckets,
                           gr.heavy_items.as<HeavyItem>(), gr.heavy_buckets.as<HeavyBucket>(), gr.heavy_counters.as<uint32_t>(), L);
    else
        hipLaunchKernelGGL((k_bucket_accum<F, false>), dim3((unsigned)((lanes + 255) / 256), ns), dim3(256), 0, s,
                           views, job->sorted.as<uint32_t>(), job->offsets.as<uint32_t>(), job->order.as<uint32_t>(), lanes, buckets,
                           gr.heavy_items.as<HeavyItem>(), gr.heavy_buckets.as<HeavyBucket>(), gr.heavy_counters.as<uint32_t>(), L);
    if (time_it) g_dominant_timer.end(s);
    if (!job->defer_reduce && host_pool_prewake_enabled()) {                    // (opt-in) msm_job_finish wakes the host pool when the stream gets here: what follows is 0.1 - 0.3 ms
        if (!job->ev_tail && hipEventCreateWithFlags(&job->ev_tail, hipEventDisableTiming) != hipSuccess) job->ev_tail = nullptr;
        job->tail_recorded = job->ev_tail && hipEventRecord(job->ev_tail, s) == hipSuccess;
    }
    hipLaunchKernelGGL(k_heavy_parts<F>, dim3(HEAVY_PART_BLOCKS, ns), dim3(256), 256 * sizeof(LdsPoint<F>), s,
                       views, job->sorted.as<uint32_t>(), gr.heavy_items.as<HeavyItem>(), gr.heavy_counters.as<uint32_t>(), gr.heavy_partials.as<XYZZ<F>>(), buckets, L, (int)out29);
    hipLaunchKernelGGL(k_heavy_merge<F>, dim3(HEAVY_MERGE_BLOCKS, ns), dim3(256), 256 * sizeof(LdsPoint<F>), s,
                       gr.heavy_buckets.as<HeavyBucket>(), gr.heavy_counters.as<uint32_t>(), gr.heavy_partials.as<XYZZ<F>>(), buckets, L, (int)out29, resume);
    if (job->defer_reduce) {                                                    // a piece of a larger job: its buckets wait for the next piece
        if (hipGetLastError() != hipSuccess) { set_error("msm kernel launch failed"); return ZKG_ERROR; }
        return ZKG_OK;
    }
    const XYZZ<F> *red_in = buckets; size_t in_stride = L.buckets;
    bool folded29 = false;
    if constexpr (sizeof(F) == sizeof(Fq)) {
        if (gr.table && out29) {
            hipLaunchKernelGGL(k_bucket_fold29, dim3((2 * g.B + 255) / 256), dim3(256), 0, s, reinterpret_cast<const Bucket29 *>(buckets), job->counts.as<uint32_t>(), g.W, g.B,
                               gr.folded.as<Bucket29>(), job->critical ? 1 : 0);
            red_in = gr.folded.as<XYZZ<F>>(); in_stride = L.folded; folded29 = true;
        }
    }
    if (gr.table && !folded29) {
        uint32_t slots = 1; while (slots < g.W && slots < 32) slots <<= 1;                // window slots per workgroup: W rounded up to a power of two (<= 32)
        uint32_t fold_b_log = 0; while ((FOLD_THREADS >> (fold_b_log + 1)) >= slots) ++fold_b_log;
        hipLaunchKernelGGL(k_bucket_fold<F>, dim3((g.B + (1u << fold_b_log) - 1) >> fold_b_log, ns), dim3(FOLD_THREADS), FOLD_THREADS * sizeof(LdsPoint<F>), s,
                           buckets, job->counts.as<uint32_t>(), g.W, g.B, fold_b_log, gr.folded.as<XYZZ<F>>(), L);
        red_in = gr.folded.as<XYZZ<F>>(); in_stride = L.folded;
    }
    bool reduce29 = false;
    if constexpr (sizeof(F) == sizeof(Fq)) {
        reduce29 = !accum32 && !red32_env;                                                                 // (ZKG_REDUCE_32: A/B switch)
        if (reduce29) {
            const size_t lds = 2 * RG::LANES * sizeof(LdsPoint29);
            const void *rin = red_in; XYZZ<Fq> *rout = reinterpret_cast<XYZZ<Fq> *>(gr.red_out.p);
            static const bool red_quad = getenv("ZKG_REDUCE_QUAD") != nullptr;                            // A/B switch: round 3's quad kernel
            auto launch_red = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)gr.nred, ns), dim3(RG::THREADS), lds, s, rin, g.B, gr.cpw, rout, in_stride, L.red_out, job->critical ? 1 : 0); };
            auto launch_redp = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)gr.nred, ns), dim3(2 * RG::LANES), lds, s, rin, g.B, gr.cpw, rout, in_stride, L.red_out, job->critical ? 1 : 0); };
            // one bucket per logical lane (a table launch's folded set: the prover): the chain is 17 steps either way and the quad kernel's two
            // wavefronts per SIMD hide its LDS rounds a little better (90 against 94 us); the pair form is for the long chains
            static const bool red_pair = getenv("ZKG_REDUCE_PAIR") != nullptr;                            // A/B switch: the pair kernel for the large reduction too
            if (!red_quad && !red_pair && out29 && red_l_log == RED_L_LOG_LARGE) {
                gr.red_slots = 3;
                hipLaunchKernelGGL(k_bucket_reduce29l, dim3((unsigned)gr.nred, ns), dim3(2 * RG::LANES), 4 * RG::LANES * sizeof(LdsPoint29), s,
                                   reinterpret_cast<const Bucket29 *>(rin), g.B, gr.cpw, rout, in_stride, L.red_out, job->critical ? 1 : 0);
            } else if (!red_quad && red_l_log != RED_L_LOG_TINY) {
                if (out29) {
                    if (red_l_log == RED_L_LOG_LARGE) launch_redp(k_bucket_reduce29p<RED_L_LOG_LARGE, true>);
                    else if (red_l_log == RED_L_LOG_SMALL) launch_redp(k_bucket_reduce29p<RED_L_LOG_SMALL, true>);
                    else launch_redp(k_bucket_reduce29p<RED_L_LOG_TINY, true>);
                } else {
                    if (red_l_log == RED_L_LOG_LARGE) launch_redp(k_bucket_reduce29p<RED_L_LOG_LARGE, false>);
                    else if (red_l_log == RED_L_LOG_SMALL) launch_redp(k_bucket_reduce29p<RED_L_LOG_SMALL, false>);
                    else launch_redp(k_bucket_reduce29p<RED_L_LOG_TINY, false>);
                }
            } else if (out29) {
                if (red_l_log == RED_L_LOG_LARGE) launch_red(k_bucket_reduce29<RED_L_LOG_LARGE, true>);
                else if (red_l_log == RED_L_LOG_SMALL) launch_red(k_bucket_reduce29<RED_L_LOG_SMALL, true>);
                else launch_red(k_bucket_reduce29<RED_L_LOG_TINY, true>);
            } else {
                if (red_l_log == RED_L_LOG_LARGE) launch_red(k_bucket_reduce29<RED_L_LOG_LARGE, false>);
                else if (red_l_log == RED_L_LOG_SMALL) launch_red(k_bucket_reduce29<RED_L_LOG_SMALL, false>);
                else launch_red(k_bucket_reduce29<RED_L_LOG_TINY, false>);
            }
        }
    }
    if (reduce29) {}
    else if (red_l_log == RED_L_LOG_LARGE)
        hipLaunchKernelGGL((k_bucket_reduce<F, RED_L_LOG_LARGE>), dim3((unsigned)gr.nred, ns), dim3(RG::THREADS), 2 * RG::LANES * sizeof(LdsPoint<F>), s,
                           red_in, g.B, gr.cpw, gr.red_out.as<XYZZ<F>>(), in_stride, L.red_out);
    else if (red_l_log == RED_L_LOG_SMALL)
        hipLaunchKernelGGL((k_bucket_reduce<F, RED_L_LOG_SMALL>), dim3((unsigned)gr.nred, ns), dim3(RG::THREADS), 2 * RG::LANES * sizeof(LdsPoint<F>), s,
                           red_in, g.B, gr.cpw, gr.red_out.as<XYZZ<F>>(), in_stride, L.red_out);
    else
        hipLaunchKernelGGL((k_bucket_reduce<F, RED_L_LOG_TINY>), dim3((unsigned)gr.nred, ns), dim3(RG::THREADS), 2 * RG::LANES * sizeof(LdsPoint<F>), s,
                           red_in, g.B, gr.cpw, gr.red_out.as<XYZZ<F>>(), in_stride, L.red_out);
    if (hipGetLastError() != hipSuccess) { set_error("msm kernel launch failed"); return ZKG_ERROR; }
    ZK_HIP(hipMemcpyAsync(gr.host_red, gr.red_out.p, ns * L.red_out * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, s));
    return ZKG_OK;
}

// host: window value V_w = sum_b (b+1) X_b = U_w + P_w, with chunk ch contributing U_ch + (ch*RED_CHUNK) * P_ch to U_w
// and P_ch to P_w; then Horner over windows (c doublings each).  A table set arrives as ONE window of weight 1: no doublings at all.
template <class F>
static XYZZ<F> host_combine(const MsmJob *job, const MsmGroup &gr, int set) {
    const MsmGeom g = job->g; const uint32_t cpw = gr.cpw, W = gr.red_windows;
    const XYZZ<F> *red = reinterpret_cast<const XYZZ<F> *>(gr.host_red) + (size_t)set * gr.red_stride;
    const size_t sl = (size_t)gr.red_slots;                                     // 2: (P, U) per chunk; 3: (P, T, A) with U = T + 8 A (k_bucket_reduce29l)
    std::vector<XYZZ<F>> V(W);
    auto window = [&](int w) {
        XYZZ<F> Usum = XYZZ<F>::inf(), suffix = XYZZ<F>::inf(), weighted = XYZZ<F>::inf(), Asum = XYZZ<F>::inf();
        for (int ch = (int)cpw - 1; ch >= 0; --ch) {
            const XYZZ<F> &P = red[sl * ((size_t)w * cpw + ch)], &U = red[sl * ((size_t)w * cpw + ch) + 1];
            Usum.add(U);
            if (sl == 3) Asum.add(red[sl * ((size_t)w * cpw + ch) + 2]);
            if (ch >= 1) { suffix.add(P); weighted.add(suffix); }           // sum_ch ch * P_ch
            else suffix.add(P);                                             // suffix == P_w now
        }
        if (sl == 3 && !Asum.is_inf()) { for (int i = 0; i < 3; ++i) Asum = Asum.dbl(); Usum.add(Asum); }
        if (!weighted.is_inf()) for (int i = 0; i < gr.chunk_log; ++i) weighted = weighted.dbl();         // * buckets per chunk
        Usum.add(weighted); Usum.add(suffix);
        V[w] = Usum;
    };
    if (gr.table) {                                                             // one window of weight 1; its chunks in segments on the host pool
        const int SEG = 16, nseg = ((int)cpw + SEG - 1) / SEG;
        if (nseg <= 1) { window(0); return V[0]; }
        // segment s = chunks [16 s, 16 s + 16): Usum_s = sum U, Psum_s = sum P, Wt_s = sum (ch - 16 s) P_ch; then
        // sum_ch ch P_ch = sum_s Wt_s + 16 * sum_s s Psum_s
        std::vector<XYZZ<F>> Us(nseg), Ps(nseg), Wt(nseg);
        host_parallel_for(nseg, [&](int sg) {
            XYZZ<F> Usum = XYZZ<F>::inf(), suffix = XYZZ<F>::inf(), weighted = XYZZ<F>::inf();
            const int lo = sg * SEG, hi = std::min<int>((int)cpw, lo + SEG);
            for (int ch = hi - 1; ch >= lo; --ch) {
                Usum.add(red[2 * (size_t)ch + 1]);
                suffix.add(red[2 * (size_t)ch]);
                if (ch > lo) weighted.add(suffix);
            }
            Us[sg] = Usum; Ps[sg] = suffix; Wt[sg] = weighted;
        });
        XYZZ<F> Usum = XYZZ<F>::inf(), suffix = XYZZ<F>::inf(), seg_weighted = XYZZ<F>::inf(), weighted = XYZZ<F>::inf();
        for (int sg = nseg - 1; sg >= 0; --sg) { Usum.add(Us[sg]); weighted.add(Wt[sg]); suffix.add(Ps[sg]); if (sg >= 1) seg_weighted.add(suffix); }
        if (!seg_weighted.is_inf()) for (int i = 0; i < 4; ++i) seg_weighted = seg_weighted.dbl();        // * 16 chunks per segment
        weighted.add(seg_weighted);
        if (!weighted.is_inf()) for (int i = 0; i < gr.chunk_log; ++i) weighted = weighted.dbl();         // * buckets per chunk
        Usum.add(weighted); Usum.add(suffix);
        return Usum;
    }
    host_parallel_for((int)W, window);                                          // the windows are independent
    // sum_j 2^(c (w0 + j ws)) V_j: Horner with c*ws doublings per owned window, then the shift of the lowest one
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = (int)W - 1; w >= 0; --w) {
        if (!acc.is_inf()) for (uint32_t i = 0; i < g.c * g.ws; ++i) acc = acc.dbl();
        acc.add(V[w]);
    }
    if (!acc.is_inf()) for (uint32_t i = 0; i < g.c * g.w0; ++i) acc = acc.dbl();
    return acc;
}

static int sort_digits(MsmJob *job, const uint32_t *d_scalars, bool mont, const uint32_t *d_gather) {
    const MsmGeom g0 = job->g; const size_t n0 = job->n;                        // as the scalars see them (k_digits)
    if (job->merge > 1) { job->g.W /= job->merge; job->g.Wt = job->g.W; job->n *= job->merge; }     // rows of the digit array read `merge` at a time from here on
    const MsmGeom g = job->g; const size_t n = job->n;
    hipStream_t s = job->sort_stream ? job->sort_stream : job->stream;
    const size_t total = (size_t)g.W * g.B;
    uint32_t slice_len = (uint32_t)std::min<size_t>(65536, std::max<size_t>(4096, ((n + 15) / 16 + 1023) / 1024 * 1024));
    uint32_t S = (uint32_t)std::max<size_t>(1, (n + slice_len - 1) / slice_len);
    size_t nblk = (total + 1024 * SCAN_ITEMS - 1) / (1024 * SCAN_ITEMS);
    if (nblk > 1024) { set_error("msm: too many buckets for the block scan"); return ZKG_ERROR; }
    if (job->digits.reserve(std::max<size_t>(1, n * g.W) * 4) || job->hist.reserve((size_t)g.W * S * g.B * 4) || job->counts.reserve(total * 4) ||
        job->offsets.reserve((total + 1) * 4) || job->scan_sums.reserve(1024 * 4) || job->class_hist.reserve((HEAVY_T_MAX + 2) * 4) ||
        job->order.reserve(total * 4) || job->sorted.reserve(std::max<size_t>(1, n * g.W) * 4)) return ZKG_ERROR;
    uint32_t *digits = job->digits.as<uint32_t>(), *hist = job->hist.as<uint32_t>(), *counts = job->counts.as<uint32_t>(),
             *offsets = job->offsets.as<uint32_t>(), *sums = job->scan_sums.as<uint32_t>(), *chist = job->class_hist.as<uint32_t>();
    static const bool radix_off = getenv("ZKG_SORT_ONE_PASS") != nullptr;                       // A/B switch
    uint32_t cbits = 6;
    static const uint32_t bin_avg = getenv("ZKG_RX_AVG") ? (uint32_t)atoi(getenv("ZKG_RX_AVG")) : RX_BIN_AVG;     // tuning aid
    while (cbits < RX_MAX_CBITS && cbits + 1 < g.c - 1 && (n >> cbits) > bin_avg) ++cbits;  // bins average <= RX_BIN_AVG entries where possible, fbits >= 2
    // (below ~2^15.5 points the one-pass sort's six small launches beat the two-pass sort's eleven: a one-payload proof's H query, 2^15 - 1
    //  points, 0.657 -> 0.603 ms; equal at 2^16, the two-pass sort 5 % ahead at 2^17)
    const bool two_pass = !radix_off && !job->one_pass_sort && g.c >= 12 && n >= 49152 && (n >> cbits) <= RX_FINE_MAX && n <= ((size_t)1 << (31 - (g.c - 1 - cbits)));
    {
        ZeroList zl{};
        int k = 0;
        zl.p[k] = chist; zl.words[k++] = HEAVY_T_MAX + 2;
        for (MsmGroup &gr : job->group) if (gr.nsets) {
            if (gr.heavy_counters.reserve(8 * MSM_MAX_SETS)) return ZKG_ERROR;
            zl.p[k] = gr.heavy_counters.as<uint32_t>(); zl.words[k++] = 2 * MSM_MAX_SETS;
        }
        if (two_pass) {
            const uint32_t nbins = g.W << cbits;
            if (job->rx_meta.reserve((3 * (size_t)nbins + 8) * 4)) return ZKG_ERROR;
            zl.p[k] = job->rx_meta.as<uint32_t>(); zl.words[k++] = 3 * nbins + 8;
        }
        static const bool generic_digits = getenv("ZKG_DIGITS_GENERIC") != nullptr;                     // A/B switch
        const dim3 dg((unsigned)((n0 + 255) / 256));                                                     // n >= 1 (msm_job_launch)
        const bool four = !d_gather && (n0 & 3) == 0 && g0.w0 == 0 && g0.ws == 1 && g0.W == g0.Wt;      // every window stored, rows 16-byte aligned
        const dim3 dg4((unsigned)((n0 / 4 + 255) / 256));
        if (!generic_digits && four && g0.c == 16) hipLaunchKernelGGL(k_digits_c4<16>, dg4, dim3(256), 0, s, d_scalars, n0, (int)mont, g0, digits, zl);
        else if (!generic_digits && four && g0.c == 12) hipLaunchKernelGGL(k_digits_c4<12>, dg4, dim3(256), 0, s, d_scalars, n0, (int)mont, g0, digits, zl);
        else if (!generic_digits && g0.c == 16) hipLaunchKernelGGL(k_digits_c<16>, dg, dim3(256), 0, s, d_scalars, d_gather, n0, (int)mont, g0, digits, zl);
        else if (!generic_digits && g0.c == 12) hipLaunchKernelGGL(k_digits_c<12>, dg, dim3(256), 0, s, d_scalars, d_gather, n0, (int)mont, g0, digits, zl);
        else hipLaunchKernelGGL(k_digits, dg, dim3(256), 0, s, d_scalars, d_gather, n0, (int)mont, g0, digits, zl);
    }
    if (two_pass) {
        // two-pass sort: 2^cbits coarse bins per window, then the remaining fbits inside LDS
        const uint32_t fbits = g.c - 1 - cbits, CB = 1u << cbits, nbins = g.W * CB, S1 = (uint32_t)((n + RX_SLICE - 1) / RX_SLICE);
        if (job->rx_tmp.reserve(n * g.W * 4) || job->rx_meta.reserve((3 * (size_t)nbins + 8) * 4)) return ZKG_ERROR;
        uint32_t *cnt = job->rx_meta.as<uint32_t>(), *base = cnt + nbins, *cursor = base + nbins + 1, *tmp = job->rx_tmp.as<uint32_t>();
        hipLaunchKernelGGL(k_rx_count, dim3(S1, g.W), dim3(1024), 0, s, digits, n, fbits, cbits, cnt);
        hipLaunchKernelGGL(k_rx_scan, dim3(1), dim3(1024), 0, s, cnt, nbins, base, cursor, offsets + total);
        // (a sort on a stream of its own runs beside an accumulation: half-size workgroups fit into the registers that one leaves free)
        static const int wg_env = getenv("ZKG_SORT_WG") ? atoi(getenv("ZKG_SORT_WG")) : 0;                 // A/B switch: 512 / 1024 everywhere
        const bool half = (wg_env ? wg_env == 512 : job->sort_stream != nullptr) && fbits <= 9;           // (k_rx_fine's scan: one fine bucket per thread)
        if (half) {
            const uint32_t S2 = (uint32_t)((n + 512 * RX_PER_THREAD - 1) / (512 * RX_PER_THREAD));
            hipLaunchKernelGGL(k_rx_scatter<512>, dim3(S2, g.W), dim3(512), (512 * RX_PER_THREAD + 4 * RX_MAX_CB + 8) * 4, s, digits, n, fbits, cbits, base, cursor, tmp);
            hipLaunchKernelGGL(k_rx_fine<512>, dim3(CB, g.W), dim3(512), (RX_FINE_MAX + 3 * (1u << fbits)) * 4, s, tmp, fbits, cbits, g.B, base, counts, offsets,
                               job->sorted.as<uint32_t>());
        } else {
            hipLaunchKernelGGL(k_rx_scatter<1024>, dim3(S1, g.W), dim3(1024), (RX_SLICE + 4 * RX_MAX_CB + 8) * 4, s, digits, n, fbits, cbits, base, cursor, tmp);
            hipLaunchKernelGGL(k_rx_fine<1024>, dim3(CB, g.W), dim3(1024), (RX_FINE_MAX + 3 * (1u << fbits)) * 4, s, tmp, fbits, cbits, g.B, base, counts, offsets,
                               job->sorted.as<uint32_t>());
        }
    } else {                                                                                    // one pass: histogram, scans, k_place
    hipLaunchKernelGGL(k_hist, dim3(S, g.W), dim3(SORT_THREADS), g.B * 4, s, digits, n, g.B, S, slice_len, hist);
    hipLaunchKernelGGL(k_colscan, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hist, g.B, S, total, counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nblk), dim3(1024), 0, s, counts, offsets, sums, total);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums, (uint32_t)nblk, offsets + total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblk), dim3(1024), 0, s, offsets, sums, total);
    hipLaunchKernelGGL(k_place, dim3(S, g.W), dim3(SORT_THREADS), g.B * 4, s, digits, n, g.B, S, slice_len, hist, offsets, job->sorted.as<uint32_t>());
    }
    hipLaunchKernelGGL(k_class_hist, dim3((unsigned)((total + 1023) / 1024)), dim3(1024), 0, s, counts, offsets, total, chist);
    hipLaunchKernelGGL(k_class_scan, dim3(1), dim3(1024), 0, s, chist, offsets, total);
    hipLaunchKernelGGL(k_order_place, dim3((unsigned)((total + 1023) / 1024)), dim3(1024), 0, s, counts, offsets, total, chist, job->order.as<uint32_t>());
    if (hipGetLastError() != hipSuccess) { set_error("msm sort launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}

MsmJob *msm_job_create(hipStream_t s, bool own_stream, bool high_priority) {
    MsmJob *j = new MsmJob();
    j->stream = s;
    if (own_stream) {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (!hip_ok(hipStreamCreateWithPriority(&j->stream, hipStreamNonBlocking, high_priority ? prio_hi : prio_lo), "hipStreamCreate", __FILE__, __LINE__)) { delete j; return nullptr; }
        j->own_stream = true;
    }
    return j;
}
void msm_job_set_window(MsmJob *j, int c) { if (j) j->window_hint = c; }
bool crit_priority_enabled() { static const bool on = !(getenv("ZKG_CRIT_PRIO") && atoi(getenv("ZKG_CRIT_PRIO")) == 0); return on; }
void msm_job_set_critical(MsmJob *j, bool critical) { if (j) j->critical = critical && crit_priority_enabled(); }
void msm_job_set_row_merge(MsmJob *j, uint32_t f) { if (j) j->merge_hint = f ? f : 1; }
void msm_job_set_skewed(MsmJob *j, bool skewed) { if (j) j->one_pass_sort = skewed; }
void msm_job_set_window_subset(MsmJob *j, uint32_t w0, uint32_t ws) { if (j) { j->w0 = w0; j->ws = ws ? ws : 1; } }
hipStream_t msm_job_stream(MsmJob *j) { return j->stream; }
void msm_job_destroy(MsmJob *j) {
    if (!j) return;
    for (DevBuf *b : {&j->digits, &j->hist, &j->counts, &j->offsets, &j->scan_sums, &j->class_hist, &j->order, &j->sorted, &j->rx_tmp, &j->rx_meta}) b->release();
    for (auto &gr : j->group) gr.release();
    if (j->aux) { (void)hipStreamSynchronize(j->aux); (void)hipStreamDestroy(j->aux); }
    if (j->ev_fork) (void)hipEventDestroy(j->ev_fork);
    if (j->ev_tail) (void)hipEventDestroy(j->ev_tail);
    if (j->ev_join) (void)hipEventDestroy(j->ev_join);
    if (j->own_stream) (void)hipStreamDestroy(j->stream);
    delete j;
}

static MsmJob g_default_job;          // the synchronous entry points share one job (serialised by its mutex); only it feeds the kernel timer
static MsmJob g_piece_job;            // msm_g1_host_scalars: the second set of sort buffers its pieces alternate over (used under g_default_job's mutex)
static hipStream_t g_sort_hi = nullptr;   // and the high-priority stream their digit sorts run on

// enqueue: one digit sort of the scalars (element d_gather[i] of d_scalars when a gather list is given), then one accumulate + reduce
// per base set
int msm_job_launch(MsmJob *job, const MsmBases *sets, int nsets, const uint32_t *d_scalars, size_t n, bool scalars_mont, const uint32_t *d_gather) {
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) { if (dbg) fprintf(stderr, "[zkg]     %-18s %8.3f ms\n", w, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count()); };
    if (n >= ((size_t)1 << 31) || nsets < 0 || nsets > MSM_MAX_SETS) { set_error("msm: bad size"); return ZKG_ERROR; }
    bool any_table = false;
    for (int i = 0; i < nsets; ++i) any_table = any_table || sets[i].level_stride != 0;
    for (int i = 0; i < nsets; ++i) if ((sets[i].level_stride != 0) != any_table) { set_error("msm: table and plain base sets cannot share a launch"); return ZKG_ERROR; }
    if (any_table && (job->w0 != 0 || job->ws != 1 || job->window_hint <= 0)) { set_error("msm: a table launch covers all windows at the table's window size"); return ZKG_ERROR; }
    job->g = pick_geom(n, job->window_hint, job->w0, job->ws); job->n = n;
    job->empty = false;
    MsmBases by_field[2][MSM_MAX_SETS];
    for (MsmGroup &gr : job->group) gr.nsets = 0;
    job->group[0].g2 = false; job->group[1].g2 = true;
    for (int i = 0; i < nsets; ++i) { MsmGroup &gr = job->group[sets[i].g2 ? 1 : 0]; by_field[sets[i].g2 ? 1 : 0][gr.nsets] = sets[i]; gr.out_index[gr.nsets] = gr.nsets; ++gr.nsets; }
    if (job->g.W == 0 || n == 0) { job->empty = true; return ZKG_OK; }        // no point, or this rank owns no window: the identity
    if ((uint64_t)n * job->g.W >= ((uint64_t)1 << 32)) { set_error("msm: n * windows exceeds the 32-bit index space of the sorted list (n < 2^28)"); return ZKG_ERROR; }
    job->merge = 1;
    if (any_table && job->merge_hint > 1 && !d_gather && job->g.W % job->merge_hint == 0 && (uint64_t)n * job->merge_hint < ((uint64_t)1 << 31)) {
        bool ok = true;
        for (int i = 0; i < nsets; ++i) ok = ok && sets[i].level_stride == n && sets[i].index_sub == 0 && !sets[i].remap;      // levels back to back, entry i = point i
        if (ok) job->merge = job->merge_hint;
    }
    job->converted_aside = false;
    {
        static const bool accum32 = getenv("ZKG_ACCUM_32") != nullptr, inline29 = getenv("ZKG_TO29_INLINE") != nullptr;       // A/B switches
        MsmGroup &g1 = job->group[0];
        if (!accum32 && !inline29 && !any_table && !d_gather && g1.nsets == 1 && by_field[0][0].index_sub == 0 && !by_field[0][0].remap && !by_field[0][0].p29) {
            if (!job->aux && (hipStreamCreateWithFlags(&job->aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&job->ev_fork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&job->ev_join, hipEventDisableTiming) != hipSuccess)) { set_error("msm: side stream"); return ZKG_ERROR; }
            if (g1.bases29.reserve(n * sizeof(Rec64))) return ZKG_ERROR;
            ZK_HIP(hipEventRecord(job->ev_fork, job->stream));
            ZK_HIP(hipStreamWaitEvent(job->aux, job->ev_fork, 0));
            hipLaunchKernelGGL(k_bases_to29, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, job->aux, reinterpret_cast<const Affine<Fq> *>(by_field[0][0].p), n, g1.bases29.as<Rec64>());
            ZK_HIP(hipEventRecord(job->ev_join, job->aux));
            job->converted_aside = true;
        }
    }
    // an error return after the fork must not leave the side stream reading the caller's bases (a synchronous caller may free them next)
    auto fail = [&] { if (job->converted_aside) { (void)hipStreamSynchronize(job->aux); job->converted_aside = false; } return ZKG_ERROR; };
    if (sort_digits(job, d_scalars, scalars_mont, d_gather)) return fail();
    if (job->sort_stream && job->sort_stream != job->stream) {                 // the accumulation (job->stream) behind the sort
        if (hipEventRecord(job->ev_sorted, job->sort_stream) != hipSuccess || hipStreamWaitEvent(job->stream, job->ev_sorted, 0) != hipSuccess) return fail();
    }
    if (job->merge > 1) {
        for (int k = 0; k < 2; ++k) for (int i = 0; i < job->group[k].nsets; ++i) by_field[k][i].level_stride *= job->merge;
    }
    lap("sort enqueued");
    const bool timed_field_g2 = job->group[0].nsets == 0;                       // the kernel timer follows the first G1 launch (G2 when there is no G1 set)
    if (job->group[1].nsets && launch_accumulate<Fq2>(job, job->group[1], by_field[1], d_gather, job == &g_default_job && timed_field_g2)) return fail();   // G2 first: the longer chains
    if (job->group[0].nsets && launch_accumulate<Fq>(job, job->group[0], by_field[0], d_gather, job == &g_default_job || job == &g_piece_job)) return fail();
    lap("accum enqueued");
    return ZKG_OK;
}
// wait for the job's stream and finish on the host; outputs in launch order: out_g1[k] for the k-th G1 set, out_g2[k] for the k-th G2 set
int msm_job_finish(MsmJob *job, G1 *out_g1, G2 *out_g2) {
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // The host's tail (host_combine) hands sixteen windows' chunk sums to the host pool the moment the stream has drained; its workers have been
    // asleep for the whole call, and waking them through their condition variable costs that work ~12 us.  With ZKG_POOL_PREWAKE=1: wait for the last
    // accumulation first, wake the pool then — it polls for the work while the fold / reduction run (0.1 - 0.3 ms) — and only then wait for the
    // stream.  Off by default: see host_pool_prewake.
    if (job->tail_recorded) { job->tail_recorded = false; if (hipEventSynchronize(job->ev_tail) == hipSuccess) host_pool_prewake(600); }
    ZK_HIP(hipStreamSynchronize(job->stream));
    const auto t1 = std::chrono::steady_clock::now();
    struct Lap { bool on; std::chrono::steady_clock::time_point a, b; ~Lap() { if (on) fprintf(stderr, "[zkg]     job finish: waited %.3f ms, host combine %.3f ms\n", std::chrono::duration<float, std::milli>(b - a).count(), std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - b).count()); } } lap_{dbg, t0, t1};
    for (int k = 0; k < job->group[0].nsets; ++k) out_g1[k] = job->empty ? G1::inf() : host_combine<Fq>(job, job->group[0], k);
    for (int k = 0; k < job->group[1].nsets; ++k) out_g2[k] = job->empty ? G2::inf() : host_combine<Fq2>(job, job->group[1], k);
    return ZKG_OK;
}


int msm_shared(const G1Affine *const *d_g1_bases, int n_g1, const G2Affine *d_g2_bases, const uint32_t *d_scalars, size_t n,
               bool scalars_mont, G1 *out_g1, G2 *out_g2, hipStream_t s, uint32_t w0, uint32_t ws, bool mostly_bits) {
    if (n_g1 < 0 || n_g1 + (d_g2_bases ? 1 : 0) > MSM_MAX_SETS) { set_error("msm: too many base sets"); return ZKG_ERROR; }
    std::lock_guard<std::mutex> lk(g_default_job.mu);
    g_dominant_timer.new_call();
    g_default_job.stream = s; g_default_job.w0 = w0; g_default_job.ws = ws ? ws : 1; g_default_job.one_pass_sort = mostly_bits;
    MsmBases sets[MSM_MAX_SETS]; int nsets = 0;
    for (int i = 0; i < n_g1; ++i) { sets[nsets] = MsmBases(); sets[nsets].p = d_g1_bases[i]; ++nsets; }
    if (d_g2_bases) { sets[nsets] = MsmBases(); sets[nsets].p = d_g2_bases; sets[nsets].g2 = true; ++nsets; }
    // Above 2^23 points the job is cut into 2^23-point pieces run one after the other and summed on the host: the sort's 32-bit entry
    // space, its LDS bin sizes and the workspace (40 B per point and window) are sized for that, and per point a 2^26-point launch was
    // measured 2.2x slower than eight 2^23-point ones (275 ms against 8 x 15.3 ms).  Window-sharded launches keep their single pass.
    if (n >= ((size_t)1 << 28)) { set_error("msm: at most 2^28 - 1 points per call"); return ZKG_ERROR; }      // refused before a byte is touched
    const size_t PIECE = (size_t)1 << 23;
    if (n <= PIECE || w0 != 0 || g_default_job.ws != 1) {
        if (msm_job_launch(&g_default_job, sets, nsets, d_scalars, n, scalars_mont, nullptr)) return ZKG_ERROR;
        return msm_job_finish(&g_default_job, out_g1, out_g2);
    }
    G1 acc1[MSM_MAX_SETS]; G2 acc2[MSM_MAX_SETS];
    for (int i = 0; i < MSM_MAX_SETS; ++i) { acc1[i] = G1::inf(); acc2[i] = G2::inf(); }
    for (size_t lo = 0; lo < n; lo += PIECE) {
        const size_t cnt = std::min(PIECE, n - lo);
        MsmBases piece[MSM_MAX_SETS];
        for (int i = 0; i < nsets; ++i) { piece[i] = sets[i]; piece[i].p = (const char *)sets[i].p + lo * (sets[i].g2 ? sizeof(G2Affine) : sizeof(G1Affine)); }
        G1 p1[MSM_MAX_SETS]; G2 p2[MSM_MAX_SETS];
        if (msm_job_launch(&g_default_job, piece, nsets, d_scalars + 8 * lo, cnt, scalars_mont, nullptr) || msm_job_finish(&g_default_job, p1, p2)) return ZKG_ERROR;
        for (int i = 0; i < n_g1; ++i) acc1[i].add(p1[i]);
        if (d_g2_bases) acc2[0].add(p2[0]);
    }
    for (int i = 0; i < n_g1; ++i) out_g1[i] = acc1[i];
    if (d_g2_bases) out_g2[0] = acc2[0];
    return ZKG_OK;
}

// Bases resident, scalars in HOST memory: SURVEY.md section 8(d)'s step ("wall clock around the call incl. H2D of scalars; bases resident") as
// one entry point.  The upload is 32 B x n over PCIe (0.6 ms at 2^20 points) against 1.9 ms of work that cannot start on a scalar it has not
// seen — so the job is cut by POINTS into pieces (1/4, 1/4, 1/2 of them: a small first piece starts the chip early, and every later
// piece's upload is shorter than the work of the piece before it), each piece sorted and accumulated as it lands, all pieces into ONE set of
// buckets (k_bucket_accum29's `resume`: a lane picks its bucket's 29-bit accumulator up where the last piece left it), and one reduction at
// the end.  Same point as msm_g1 on the uploaded vector: the buckets hold the same sums, whatever the order of the additions.
int msm_g1_host_scalars(const G1Affine *d_bases, const uint32_t *h_scalars, size_t n, bool mont, G1 *out, hipStream_t s) {
    if (n >= ((size_t)1 << 28)) { set_error("msm: at most 2^28 - 1 points per call"); return ZKG_ERROR; }
    if (!n) { *out = G1::inf(); return ZKG_OK; }
    static const int max_pieces = getenv("ZKG_MSM_PIECES") ? std::max(1, std::min(8, atoi(getenv("ZKG_MSM_PIECES")))) : 4;        // tuning aid (1: one upload, one launch); 2^20 points, sorts beside the accumulation (ACC29_VGPRS, sort_wave_priority): 3 / 4 / 5 pieces 1.99 / 1.94 / 2.02 ms (before: 2.19 / 2.27 / 2.41)
    static const bool no29 = getenv("ZKG_ACCUM_32") != nullptr || getenv("ZKG_REDUCE_32") != nullptr;
    const bool pieces = max_pieces > 1 && !no29 && n >= ((size_t)1 << 19) && n <= ((size_t)1 << 23);
    if (n > ((size_t)1 << 23)) {                                                // huge: one upload, then the resident path's own 2^23-point pieces
        ScopedDevBuf tmp;
        if (tmp.reserve(n * 32)) return ZKG_ERROR;
        ZK_HIP(hipMemcpyAsync(tmp.p, h_scalars, n * 32, hipMemcpyHostToDevice, s));
        return msm_g1(d_bases, tmp.as<uint32_t>(), n, mont, out, s, false);     // (returns after the stream has drained: tmp may go)
    }
    MsmJob &J = g_default_job;
    std::lock_guard<std::mutex> lk(J.mu);
    g_dominant_timer.new_call();
    if (J.hs_scalars.reserve(n * 32)) return ZKG_ERROR;
    if (!J.copy) {
        if (hipStreamCreateWithFlags(&J.copy, hipStreamNonBlocking) != hipSuccess) { J.copy = nullptr; set_error("msm: copy stream"); return ZKG_ERROR; }
        for (auto &e : J.ev_piece) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error("msm: event"); return ZKG_ERROR; }
    }
    uint32_t *d_sc = J.hs_scalars.as<uint32_t>();
    J.stream = s; J.w0 = 0; J.ws = 1; J.one_pass_sort = false;
    MsmBases set; set.p = d_bases;
    if (!pieces) {                                                              // small or switched off: one upload in stream order, one launch
        ZK_HIP(hipMemcpyAsync(d_sc, h_scalars, n * 32, hipMemcpyHostToDevice, s));
        if (msm_job_launch(&J, &set, 1, d_sc, n, mont, nullptr)) return ZKG_ERROR;
        return msm_job_finish(&J, out, nullptr);
    }
    // piece boundaries: halves from the top, the lowest one split once more -> n/8, n/8, n/4, n/2 for four pieces
    size_t cut[9]; int P = max_pieces;
    cut[P] = n;
    for (int k = P - 1; k >= 1; --k) cut[k] = (cut[k + 1] / 2) & ~(size_t)255;
    cut[0] = 0;
    if (const char *e = getenv("ZKG_MSM_CUTS")) {                              // tuning aid: the piece boundaries in 64ths of n, e.g. "8,32" -> n/8, 3n/8, n/2
        int k = 1; const char *q = e;
        while (*q && k < 8) { const long v = strtol(q, const_cast<char **>(&q), 10); if (v <= 0 || v >= 64) break; cut[k++] = (n * (size_t)v / 64) & ~(size_t)255; if (*q == ',') ++q; }
        if (k > 1) { P = k; cut[P] = n; for (int i = 1; i < P; ++i) if (cut[i] < cut[i - 1]) cut[i] = cut[i - 1]; }
    }
    for (int k = 0; k < P; ++k) {
        ZK_HIP(hipMemcpyAsync(d_sc + 8 * cut[k], h_scalars + 8 * cut[k], (cut[k + 1] - cut[k]) * 32, hipMemcpyHostToDevice, J.copy));
        ZK_HIP(hipEventRecord(J.ev_piece[k], J.copy));
    }
    // Pieces alternate between two jobs — two sets of sort buffers, ONE set of buckets, all accumulations in order on the caller's stream —
    // and every piece's digit sort runs on a high-priority stream of its own beside the accumulation of the piece before it: the sort's
    // small dependent launches and single-workgroup scans stand a third of a step when alone and cost the accumulation little beside it
    // (without the priority they crawl: k_rx_count 134 us instead of 11; and they only run BESIDE it because the accumulation leaves a
    // quarter of each SIMD's registers free and the sort's wavefronts raise their own issue priority — ACC29_VGPRS, sort_wave_priority).  The bases' 29-bit records are made once for all pieces, on
    // the side stream, under the first upload.
    static const bool two_jobs = getenv("ZKG_MSM_PIECES_ONE_STREAM") == nullptr;                          // A/B switch
    MsmJob &K = g_piece_job;
    if (two_jobs && !g_sort_hi) {
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // (confining this stream to 16 ... 128 compute units with a CU mask instead was measured: 2.11 -> 4.25 ... 2.70 ms — the sort needs the chip's width)
        bool ok = hipStreamCreateWithPriority(&g_sort_hi, hipStreamNonBlocking, hi) == hipSuccess;
        // (these events without their system-scope fence — hipEventDisableSystemFence — were measured: 1.994 against 1.998 ms, nothing; an A/B on one box)
        for (MsmJob *w : {&J, &K}) ok = ok && hipEventCreateWithFlags(&w->ev_sorted, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&w->ev_acc_done, hipEventDisableTiming) == hipSuccess;
        if (!ok) { g_sort_hi = nullptr; set_error("msm: sort stream"); return ZKG_ERROR; }
    }
    hipStream_t s_hi = two_jobs ? g_sort_hi : nullptr;
    // all bases as 29-bit records, once (k_bases_to29 on the job's side stream: it runs under the first piece's upload)
    if (!J.aux && (hipStreamCreateWithFlags(&J.aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&J.ev_fork, hipEventDisableTiming) != hipSuccess ||
                   hipEventCreateWithFlags(&J.ev_join, hipEventDisableTiming) != hipSuccess)) { set_error("msm: side stream"); return ZKG_ERROR; }
    if (J.group[0].bases29.reserve(n * sizeof(Rec64))) return ZKG_ERROR;
    ZK_HIP(hipEventRecord(J.ev_fork, s));                                     // behind whatever the caller queued (the work that made the bases), and the last call's accumulation
    ZK_HIP(hipStreamWaitEvent(J.aux, J.ev_fork, 0));
    hipLaunchKernelGGL(k_bases_to29, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, J.aux, reinterpret_cast<const Affine<Fq> *>(d_bases), n, J.group[0].bases29.as<Rec64>());
    ZK_HIP(hipEventRecord(J.ev_join, J.aux));
    ZK_HIP(hipStreamWaitEvent(s, J.ev_join, 0));
    if (s_hi) ZK_HIP(hipStreamWaitEvent(s_hi, J.ev_fork, 0));
    const int c = (int)pick_geom(n, J.window_hint).c;                         // every piece under the whole job's window size
    const int saved_hint = J.window_hint, saved_hint_k = K.window_hint;
    J.window_hint = c; K.window_hint = c; K.bucket_owner = &J; K.w0 = 0; K.ws = 1; K.one_pass_sort = false; K.stream = s;
    int rc = ZKG_OK; bool started = false; bool used[2] = {false, false};
    for (int k = 0; k < P && rc == ZKG_OK; ++k) {
        if (cut[k + 1] == cut[k] && k + 1 < P) continue;                      // (an empty piece; the last one always runs: it carries the reduction)
        const int which = (two_jobs && ((P - 1 - k) & 1)) ? 1 : 0;            // the last piece is the default job's: it carries the reduction and the result
        MsmJob &W = which ? K : J;
        MsmBases piece = set; piece.p = reinterpret_cast<const char *>(d_bases) + cut[k] * sizeof(G1Affine);
        piece.p29 = J.group[0].bases29.as<Rec64>() + cut[k];
        W.resume = started; W.defer_reduce = k + 1 < P; started = true;
        W.sort_stream = s_hi;
        hipStream_t ss = s_hi ? s_hi : s;
        // this piece's sort: after its scalars have landed, and after the accumulation that last read this job's sort buffers
        if (hipStreamWaitEvent(ss, J.ev_piece[k], 0) != hipSuccess) { rc = ZKG_ERROR; break; }
        if (s_hi && used[which] && hipStreamWaitEvent(s_hi, W.ev_acc_done, 0) != hipSuccess) { rc = ZKG_ERROR; break; }
        rc = msm_job_launch(&W, &piece, 1, d_sc + 8 * cut[k], cut[k + 1] - cut[k], mont, nullptr);
        if (rc == ZKG_OK && s_hi && hipEventRecord(W.ev_acc_done, s) != hipSuccess) rc = ZKG_ERROR;
        used[which] = true;
    }
    J.resume = false; J.defer_reduce = false; J.window_hint = saved_hint; J.sort_stream = nullptr;      // (the other entry points sort on the job's own stream)
    K.resume = false; K.defer_reduce = false; K.window_hint = saved_hint_k; K.bucket_owner = nullptr; K.sort_stream = nullptr; K.stream = nullptr;
    if (rc != ZKG_OK) { (void)hipStreamSynchronize(J.copy); if (s_hi) (void)hipStreamSynchronize(s_hi); (void)hipStreamSynchronize(s); return ZKG_ERROR; }
    return msm_job_finish(&J, out, nullptr);
}

int msm_g1(const G1Affine *d_bases, const uint32_t *d_scalars, size_t n, bool mont, G1 *out, hipStream_t s, bool mostly_bits) {
    return msm_shared(&d_bases, 1, nullptr, d_scalars, n, mont, out, nullptr, s, 0, 1, mostly_bits);
}
int msm_g2(const G2Affine *d_bases, const uint32_t *d_scalars, size_t n, bool mont, G2 *out, hipStream_t s, bool mostly_bits) {
    return msm_shared(nullptr, 0, d_bases, d_scalars, n, mont, nullptr, out, s, 0, 1, mostly_bits);
}

// ---- per-window tables of a resident base set (built once per key) ------------------------------------------------
template <class F>
static int window_table_build_t(WindowTable &t, const Affine<F> *d_bases, size_t n, int c, hipStream_t s) {
    t.n = n; t.c = c; t.W = (SCALAR_BITS + c - 1) / c; t.g2 = sizeof(F) != sizeof(Fq);
    if (t.buf.reserve(std::max<size_t>(1, n) * t.W * sizeof(Affine<F>))) return ZKG_ERROR;
    if (!n) return ZKG_OK;
    Affine<F> *lv = t.buf.as<Affine<F>>();
    ZK_HIP(hipMemcpyAsync(lv, d_bases, n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, s));
    constexpr int K = sizeof(F) != sizeof(Fq) ? 2 : 4;                          // levels per launch (the K results stay in registers)
    for (int w = 1; w < t.W; w += K)
        hipLaunchKernelGGL((k_table_levels<F, K>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lv + (size_t)(w - 1) * n, lv + (size_t)w * n, n, (uint32_t)c,
                           std::min(K, t.W - w));
    if (hipGetLastError() != hipSuccess) { set_error("window table launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}
int window_table_build_g1(WindowTable &t, const G1Affine *d_bases, size_t n, int c, hipStream_t s) { return window_table_build_t<Fq>(t, d_bases, n, c, s); }
int window_table_records29(WindowTable &t, hipStream_t s) {
    if (t.g2) { set_error("window_table_records29: G1 tables only"); return ZKG_ERROR; }
    const size_t total = t.n * (size_t)t.W;
    if (t.rec29.reserve(std::max<size_t>(1, total) * sizeof(Rec64))) return ZKG_ERROR;
    if (total) hipLaunchKernelGGL(k_bases_to29, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t.buf.as<Affine<Fq>>(), total, t.rec29.as<Rec64>());
    if (hipGetLastError() != hipSuccess) { set_error("window table records launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}
int window_table_build_g2(WindowTable &t, const G2Affine *d_bases, size_t n, int c, hipStream_t s) { return window_table_build_t<Fq2>(t, d_bases, n, c, s); }

// ---- witness split: classification and the flat sum of the bases whose scalar is one ----------------------------------
int witness_classify(const Fr *d_z, size_t n1, uint8_t *d_tags, uint32_t *d_listed, uint32_t *d_count, hipStream_t s, const uint32_t *d_subset_pos) {
    ZK_HIP(hipMemsetAsync(d_count, 0, 8, s));
    if (n1) hipLaunchKernelGGL(k_classify, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, d_z, n1, d_tags, d_listed, d_count, d_subset_pos);
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}
template <class F> static int gather_points_t(const Affine<F> *d_src, const uint32_t *d_idx, size_t count, uint32_t index_sub, Affine<F> *d_out, hipStream_t s) {
    if (count) hipLaunchKernelGGL(k_gather_points<F>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, d_src, d_idx, count, index_sub, d_out);
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}
int gather_points_g1(const G1Affine *d_src, const uint32_t *d_idx, size_t count, uint32_t index_sub, G1Affine *d_out, hipStream_t s) { return gather_points_t<Fq>(d_src, d_idx, count, index_sub, d_out, s); }
int gather_points_g2(const G2Affine *d_src, const uint32_t *d_idx, size_t count, uint32_t index_sub, G2Affine *d_out, hipStream_t s) { return gather_points_t<Fq2>(d_src, d_idx, count, index_sub, d_out, s); }
template <class F> static int scatter_points_t(const Affine<F> *d_src, const uint32_t *d_idx, size_t count, Affine<F> *d_out, hipStream_t s) {
    if (count) hipLaunchKernelGGL(k_scatter_points<F>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, d_src, d_idx, count, d_out);
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}
int scatter_points_g1(const G1Affine *d_src, const uint32_t *d_idx, size_t count, G1Affine *d_out, hipStream_t s) { return scatter_points_t<Fq>(d_src, d_idx, count, d_out, s); }
int scatter_points_g2(const G2Affine *d_src, const uint32_t *d_idx, size_t count, G2Affine *d_out, hipStream_t s) { return scatter_points_t<Fq2>(d_src, d_idx, count, d_out, s); }
template <class F>
static int ones_sum_launch_t(OnesSum &o, const MsmBases *sets, int nsets, const uint8_t *d_tags, size_t n1, hipStream_t s) {
    // per workgroup: list the positions tagged 1, lanes stride over the list, an 8-level LDS tree; one more workgroup per set sums the partials
    if (nsets < 1 || nsets > MSM_MAX_SETS) { set_error("ones-sum: bad set count"); return ZKG_ERROR; }
    // tags per workgroup: a 256th of the vector, at least 1024 (a short list per lane leaves only the trees) and at most 8192 (32 KiB of LDS list)
    const uint32_t span = (uint32_t)std::min<size_t>(8192, std::max<size_t>(1024, ((n1 + 255) / 256 + 255) / 256 * 256));
    const unsigned blocks = (unsigned)std::max<size_t>(1, (n1 + span - 1) / span);
    o.g2 = sizeof(F) != sizeof(Fq); o.nsets = nsets;
    if (o.partials.reserve((size_t)nsets * (blocks + 1) * sizeof(XYZZ<F>))) return ZKG_ERROR;
    if (!o.host) { if (!hip_ok(hipHostMalloc(&o.host, MSM_MAX_SETS * sizeof(G2), hipHostMallocDefault), "hipHostMalloc", __FILE__, __LINE__)) return ZKG_ERROR; }
    XYZZ<F> *part = o.partials.as<XYZZ<F>>();
    ViewSet<F> views;
    for (int i = 0; i < MSM_MAX_SETS; ++i) { const MsmBases &b = sets[i < nsets ? i : 0]; views.v[i] = BaseView<F>{reinterpret_cast<const Affine<F> *>(b.p), 0, nullptr, b.index_sub, 1, nullptr}; }
    hipLaunchKernelGGL(k_ones_sum<F>, dim3(blocks, (unsigned)nsets), dim3(256), 256 * sizeof(LdsPoint<F>) + (size_t)span * 4, s, views, d_tags, n1, span, part);
    hipLaunchKernelGGL(k_sum_partials<F>, dim3((unsigned)nsets), dim3(256), 256 * sizeof(LdsPoint<F>), s, part, blocks);
    if (hipGetLastError() != hipSuccess) { set_error("ones-sum launch failed"); return ZKG_ERROR; }
    for (int i = 0; i < nsets; ++i)
        ZK_HIP(hipMemcpyAsync((char *)o.host + (size_t)i * sizeof(XYZZ<F>), part + (size_t)i * (blocks + 1) + blocks, sizeof(XYZZ<F>), hipMemcpyDeviceToHost, s));
    return ZKG_OK;
}
int ones_sum_launch(OnesSum &o, const MsmBases *sets, int nsets, const uint8_t *d_tags, size_t n1, hipStream_t s) {
    for (int i = 1; i < nsets; ++i) if (sets[i].g2 != sets[0].g2) { set_error("ones-sum: one field per launch"); return ZKG_ERROR; }
    return sets[0].g2 ? ones_sum_launch_t<Fq2>(o, sets, nsets, d_tags, n1, s) : ones_sum_launch_t<Fq>(o, sets, nsets, d_tags, n1, s);
}
void OnesSum::release() { partials.release(); if (host) (void)hipHostFree(host); host = nullptr; }

// ---- fixed-base batch: out[i] = k_i * base (libff batch_exp with a window table; libsnark's generator, snark.cpp:91) --------------
//      Table: entry [j][d-1] = d * 256^j * base (j < 32, d = 1..255), affine, 522 KB for G1: a scalar is 32 byte digits, i.e. at most
//      32 mixed additions and no doubling (the first build added bit by bit: 254 conditional additions).  A lane takes FB_K scalars and
//      keeps their results in registers, so that ONE field inversion (Montgomery's trick over ZZZ) normalises all of them: the kernel
//      writes each affine point once and nothing else (the first build's out-of-line to_affine spilt to scratch: 123x the output bytes).
static constexpr int FB_WINDOWS = 32, FB_DIGITS = 255, FB_K = 4;
template <class F>
__global__ __launch_bounds__(256) void k_fixed_table(const Affine<F> *rows /* 256^j * base */, Affine<F> *table) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * FB_DIGITS) return;
    const uint32_t j = t / FB_DIGITS, d = t % FB_DIGITS + 1;
    const Affine<F> r = rows[j];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 7; bit >= 0; --bit) { acc = acc.dbl(); if ((d >> bit) & 1u) acc.madd(r); }
    XYZZ<F> fin = acc;
    table[t] = fin.to_affine().normalized();
}
template <class F>
__global__ __launch_bounds__(256) void k_fixed_base(const Affine<F> *table, const uint32_t *scalars, size_t n, int mont, Affine<F> *out) {
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = lane * FB_K;
    if (i0 >= n) return;
    F x[FB_K], y[FB_K], zz[FB_K], zzz[FB_K];
#pragma unroll
    for (int k = 0; k < FB_K; ++k) {
        XYZZ<F> acc = XYZZ<F>::inf();
        if (i0 + k < n) {
            Fr f;
            for (int j = 0; j < 8; ++j) f.v[j] = scalars[8 * (i0 + k) + j];
            if (mont) f = f.from_mont();                                        // the generator hands over Montgomery Fr; the ABI canonical limbs
            else f = f.normalized();
            for (int j = 0; j < FB_WINDOWS; ++j) {
                const uint32_t d = (f.v[j >> 2] >> (8 * (j & 3))) & 255u;
                if (d) acc.madd(table[j * FB_DIGITS + d - 1]);
            }
        }
        x[k] = acc.x; y[k] = acc.y; zz[k] = acc.zz; zzz[k] = acc.zzz;
    }
    // Montgomery's trick over the FB_K values of ZZZ (infinity: ZZZ = 0, replaced by one and restored below)
    F pre[FB_K], run = F::one();
#pragma unroll
    for (int k = 0; k < FB_K; ++k) { pre[k] = run; if (!zzz[k].is_zero()) run = run * zzz[k]; }
    F inv = inverse_fast(run);
#pragma unroll
    for (int k = FB_K - 1; k >= 0; --k) {
        if (i0 + k >= n) continue;
        if (zzz[k].is_zero()) { out[i0 + k] = Affine<F>::inf(); continue; }
        F zi = inv * pre[k];                                                    // 1 / ZZZ_k
        inv = inv * zzz[k];
        F t = zi * zz[k];                                                       // ZZ / ZZZ = 1 / Z
        F zi2 = t.sqr();                                                        // 1 / ZZ
        out[i0 + k] = Affine<F>{x[k] * zi2, y[k] * zi}.normalized();
    }
}

// table per base point, built once per process (the generator always uses the curve's generators)
template <class F> struct FixedTableCache { std::mutex mu; std::vector<std::pair<Affine<F>, DevBuf>> entries; };
template <class F> static FixedTableCache<F> &fixed_cache() { static FixedTableCache<F> c; return c; }
template <class F>
static const Affine<F> *fixed_table_for(const Affine<F> &base, hipStream_t s) {
    FixedTableCache<F> &c = fixed_cache<F>();
    std::lock_guard<std::mutex> lk(c.mu);
    for (auto &e : c.entries) if (memcmp(&e.first, &base, sizeof(base)) == 0) return e.second.template as<Affine<F>>();
    std::vector<Affine<F>> rows(FB_WINDOWS);
    XYZZ<F> cur = XYZZ<F>::from_affine(base);
    for (int j = 0; j < FB_WINDOWS; ++j) { rows[j] = cur.to_affine(); for (int k = 0; k < 8; ++k) cur = cur.dbl(); }
    DevBuf d_rows, d_table;
    if (d_rows.reserve(rows.size() * sizeof(Affine<F>)) || d_table.reserve((size_t)FB_WINDOWS * FB_DIGITS * sizeof(Affine<F>))) return nullptr;
    if (!hip_ok(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * sizeof(Affine<F>), hipMemcpyHostToDevice, s), "H2D", __FILE__, __LINE__)) return nullptr;
    hipLaunchKernelGGL(k_fixed_table<F>, dim3((FB_WINDOWS * FB_DIGITS + 255) / 256), dim3(256), 0, s, d_rows.as<Affine<F>>(), d_table.as<Affine<F>>());
    const bool ok = hipGetLastError() == hipSuccess && hip_ok(hipStreamSynchronize(s), "sync", __FILE__, __LINE__);
    d_rows.release();
    if (!ok) { d_table.release(); set_error("fixed-base table build failed"); return nullptr; }
    c.entries.emplace_back(base, d_table);
    return c.entries.back().second.template as<Affine<F>>();
}
template <class F>
static int fixed_base(const Affine<F> &base, const uint32_t *d_scalars, size_t n, Affine<F> *d_out, hipStream_t s, bool mont) {
    const Affine<F> *table = fixed_table_for<F>(base, s);
    if (!table) return ZKG_ERROR;
    const size_t lanes = (n + FB_K - 1) / FB_K;
    if (n) hipLaunchKernelGGL(k_fixed_base<F>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, table, d_scalars, n, (int)mont, d_out);
    hipError_t e = hipGetLastError();
    ZK_HIP(hipStreamSynchronize(s));
    if (e != hipSuccess) { set_error("fixed_base launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}
int fixed_base_g1(const G1Affine &base, const uint32_t *d_scalars, size_t n, G1Affine *d_out, hipStream_t s, bool mont) { return fixed_base<Fq>(base, d_scalars, n, d_out, s, mont); }
int fixed_base_g2(const G2Affine &base, const uint32_t *d_scalars, size_t n, G2Affine *d_out, hipStream_t s, bool mont) { return fixed_base<Fq2>(base, d_scalars, n, d_out, s, mont); }
static void fixed_tables_release() {
    { auto &c = fixed_cache<Fq>(); std::lock_guard<std::mutex> lk(c.mu); for (auto &e : c.entries) e.second.release(); c.entries.clear(); }
    { auto &c = fixed_cache<Fq2>(); std::lock_guard<std::mutex> lk(c.mu); for (auto &e : c.entries) e.second.release(); c.entries.clear(); }
}

int msm_configure() {
    bool ok = hipFuncSetAttribute((const void *)k_bucket_reduce<Fq2, RED_L_LOG_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RedGeom<Fq2>::LANES * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_bucket_reduce<Fq2, RED_L_LOG_TINY>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RedGeom<Fq2>::LANES * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_bucket_reduce<Fq2, RED_L_LOG_LARGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RedGeom<Fq2>::LANES * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_heavy_parts<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_heavy_merge<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_bucket_fold<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FOLD_THREADS * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_ones_sum<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(LdsPoint<Fq2>) + 8192 * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_ones_sum<Fq>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(LdsPoint<Fq>) + 8192 * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_sum_partials<Fq2>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (int)sizeof(LdsPoint<Fq2>)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (1 << (MAX_C - 1)) * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_place, hipFuncAttributeMaxDynamicSharedMemorySize, (1 << (MAX_C - 1)) * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_bucket_reduce29l, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * RedGeom<Fq>::LANES * (int)sizeof(LdsPoint29)) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_rx_scatter<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (RX_SLICE + 4 * RX_MAX_CB + 8) * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_rx_scatter<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (512 * RX_PER_THREAD + 4 * RX_MAX_CB + 8) * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_rx_fine<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (RX_FINE_MAX + 3 * 1024) * 4) == hipSuccess;
    ok = ok && hipFuncSetAttribute((const void *)k_rx_fine<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (RX_FINE_MAX + 3 * 1024) * 4) == hipSuccess;
    return ok ? ZKG_OK : ZKG_ERROR;
}
void msm_release_all() {
    fixed_tables_release();
    MsmJob &j = g_default_job;
    std::lock_guard<std::mutex> lk(j.mu);
    for (DevBuf *b : {&j.digits, &j.hist, &j.counts, &j.offsets, &j.scan_sums, &j.class_hist, &j.order, &j.sorted, &j.rx_tmp, &j.rx_meta}) b->release();
    for (auto &gr : j.group) gr.release();
    if (j.aux) { (void)hipStreamSynchronize(j.aux); (void)hipStreamDestroy(j.aux); j.aux = nullptr; }      // (a later zkg_init may pick another device)
    j.hs_scalars.release();
    {
        MsmJob &k = g_piece_job;
        if (g_sort_hi) { (void)hipStreamSynchronize(g_sort_hi); (void)hipStreamDestroy(g_sort_hi); g_sort_hi = nullptr; }
        for (DevBuf *b : {&k.digits, &k.hist, &k.counts, &k.offsets, &k.scan_sums, &k.class_hist, &k.order, &k.sorted, &k.rx_tmp, &k.rx_meta}) b->release();
        for (auto &gr : k.group) gr.release();
        if (k.aux) { (void)hipStreamSynchronize(k.aux); (void)hipStreamDestroy(k.aux); k.aux = nullptr; }
        if (k.ev_fork) { (void)hipEventDestroy(k.ev_fork); k.ev_fork = nullptr; }
        if (k.ev_tail) { (void)hipEventDestroy(k.ev_tail); k.ev_tail = nullptr; k.tail_recorded = false; }
        if (k.ev_join) { (void)hipEventDestroy(k.ev_join); k.ev_join = nullptr; }
        for (MsmJob *w : {&j, &k}) {
            if (w->ev_sorted) { (void)hipEventDestroy(w->ev_sorted); w->ev_sorted = nullptr; }
            if (w->ev_acc_done) { (void)hipEventDestroy(w->ev_acc_done); w->ev_acc_done = nullptr; }
        }
        k.stream = nullptr;
    }
    if (j.copy) { (void)hipStreamSynchronize(j.copy); (void)hipStreamDestroy(j.copy); j.copy = nullptr; }
    for (auto &e : j.ev_piece) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (j.ev_fork) { (void)hipEventDestroy(j.ev_fork); j.ev_fork = nullptr; }
    if (j.ev_tail) { (void)hipEventDestroy(j.ev_tail); j.ev_tail = nullptr; j.tail_recorded = false; }
    if (j.ev_join) { (void)hipEventDestroy(j.ev_join); j.ev_join = nullptr; }
}

}  // namespace zk
