// prover.hip — Groth16 prover pipeline on a device-resident proving key.
//
// Replaces libsnark's r1cs_gg_ppzksnark_prover<ppT>(pk, primary_input, auxiliary_input), the call at
// /root/reference/zklaim/snark.cpp:126, including r1cs_to_qap_witness_map (3 sparse mat-vecs, 3 iFFT,
// 3 cosetFFT, pointwise H = (A.B - C)/Z on the coset, 1 icosetFFT) and the four multi-exponentiations
// (A query; B query in G2 and G1; H query; L query), and the proof serialisation operator<< reached from
// zklaim/libsnark_wrapper.cpp:170-181.
//
// What differs from the reference by design: the pk is parsed and uploaded ONCE (zkg_crs_upload) instead
// of on every call (libsnark_wrapper.cpp:230 + the by-value copy at snark.cpp:107-109); iFFT's 1/m and
// the following cosetFFT's g^i are one fused table multiplication; the H query and the witness queries' non-bit elements live on
// the device as per-window tables (2^(c w) P_i), so a multi-exponentiation is one bucket set, one reduction and no host doublings; the witness queries are split
// as multi_exp_with_mixed_addition splits them (zeros skipped, ones summed flat, the rest through the bucket method, A / B_g1 / L
// sharing one digit sort); the prover randomness (r, s) is an explicit input.
#include "common.hpp"
#include "../../include/zkg.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <thread>
#include <vector>

namespace zk {

struct DevCsr { DevBuf rowptr, col, val; size_t nnz = 0; };

}  // namespace zk

// Everything one proof in flight needs on top of the shared key: its [1 | w], the three evaluation vectors, the transform
// scratch, a stream for the mat-vec / NTT pipeline, three MSM jobs (stream + workspace + pinned landing zone each) and the
// events that order them.  Up to three per key for small domains, created when callers overlap (see zkg_crs): a large proof's
// concurrent MSMs already fill the chip, a small one leaves room.
// One persistent helper thread per prover slot: it queues the witness streams' work while the calling thread queues the critical path, and
// later computes one of the two variable-base products of the assembly.  (std::async spawned a thread for each: 30-50 us apiece on every
// proof, and the occasional multi-millisecond outlier when the spawn was slow.)
struct Helper {
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::function<int()> task; bool has_task = false, done = true, stop = false; int result = 0;
    int device = 0;                       // the key's device: a new thread starts on device 0, and this one launches kernels on the slot's streams
    void start() { (void)hipGetDevice(&device); th = std::thread([this] { loop(); }); }
    void loop() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_task || stop; });
            if (stop) return;
            std::function<int()> t = std::move(task); has_task = false;
            lk.unlock();
            int r = ZKG_ERROR;
            try { r = t(); } catch (...) { r = ZKG_ERROR; }
            lk.lock();
            result = r; done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<int()> f) { std::lock_guard<std::mutex> lk(mu); task = std::move(f); has_task = true; done = false; cv.notify_all(); }
    int wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); return result; }
    void shutdown() { if (!th.joinable()) return; { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); th.join(); }
};

struct ProverSlot {
    zk::DevBuf z, aABC, flag, up_tags, up_idx, up_vals;    // up_*: staging of a sparse witness                   // [1 | w] and aA | aB | aC back to back (batched NTTs)
    zk::DevBuf ntt_scratch;                     // inter-pass scratch, 3 m elements
    zk::DevBuf wtags, wlisted, wcount;          // the witness split: tag per element of z, indices of the non-bit elements, their count
    hipStream_t stream = nullptr;               // upload + split + mat-vec + NTT stream
    // witness multi-exponentiations: one job for the three G1 queries (A, B_g1, L share the digit sort), one for B_g2; H on its own
    zk::MsmJob *job_w1 = nullptr, *job_w2 = nullptr, *job_h = nullptr;
    // H sharded over several devices (zkg_crs_shard_h): this slot's job, scalar slice and "slice copied" event on every shard's device
    struct HShardRun { int device = 0; zk::MsmJob *job = nullptr; zk::DevBuf scalars; };
    std::vector<HShardRun> h_runs; uint64_t h_epoch = 0;                       // h_epoch: the sharding (zkg_crs::h_epoch) these were made for
    zk::OnesSum ones_g1, ones_g2;               // flat sums of the bases whose witness element is one: (A, B_g1, L) and B_g2, on streams of their own
    std::vector<uint8_t> scan_tags; std::vector<uint32_t> scan_idx; std::vector<uint64_t> scan_vals;    // a dense witness rewritten as tags + listed values (witness_to_sparse)
    hipStream_t stream_o = nullptr;
    Helper helper;
    hipEvent_t ev[20]; bool ev_ok = false, ready = false;
    float stage_ms[8] = {0};
    // the proof in flight between prove_enqueue and prove_finish
    uint32_t *flag_host = nullptr;              // pinned: [0] lands the satisfiability flag, [1] the count of non-bit witness elements
    bool check = false; zk::Fr r, s;
    std::chrono::steady_clock::time_point t0;
};

// Host-side fixed-base table of one key element P (alpha_1, beta_1, delta_1, delta_2): entry [j][d-1] = d * 16^j * P, affine.  k * P for a
// 254-bit k is then 64 mixed additions and no doubling (~20 us for G1) instead of 254 doublings + ~127 additions (~130 us): the products
// r*delta, s*delta, rs*delta, s*alpha, r*beta that every proof needs cost less than one variable-base multiplication together.
template <class F> struct CombTable {
    std::vector<zk::Affine<F>> t;               // 64 x 15
    void build(const zk::Affine<F> &base) {
        // all 960 multiples in XYZZ first, then ONE inversion for the lot (Montgomery's trick over their ZZZ): a to_affine per entry was
        // 960 Fermat inversions, 10 - 25 ms of every key load
        std::vector<zk::XYZZ<F>> p(64 * 15);
        zk::XYZZ<F> row = zk::XYZZ<F>::from_affine(base);                        // 16^j * P
        for (int j = 0; j < 64; ++j) {
            zk::XYZZ<F> cur = row;
            for (int d = 1; d <= 15; ++d) { p[j * 15 + d - 1] = cur; cur.add(row); }
            for (int k = 0; k < 4; ++k) row = row.dbl();
        }
        t.assign(64 * 15, zk::Affine<F>::inf());
        std::vector<F> pre(p.size());
        F run = F::one();
        for (size_t i = 0; i < p.size(); ++i) { pre[i] = run; if (!p[i].is_inf()) run = run * p[i].zzz; }
        F inv = run.inverse();
        for (size_t i = p.size(); i-- > 0;) {
            if (p[i].is_inf()) continue;
            F zi = inv * pre[i];                                                 // 1 / ZZZ_i
            inv = inv * p[i].zzz;
            F z1 = zi * p[i].zz, zi2 = z1.sqr();                                 // ZZ / ZZZ = 1 / Z;  1 / ZZ
            t[i] = zk::Affine<F>{p[i].x * zi2, p[i].y * zi};
        }
    }
    zk::XYZZ<F> mul(const uint32_t k[8]) const {                                  // canonical little-endian scalar
        zk::XYZZ<F> acc = zk::XYZZ<F>::inf();
        for (int j = 0; j < 64; ++j) { uint32_t d = (k[j >> 3] >> (4 * (j & 7))) & 15u; if (d) acc.madd(t[j * 15 + d - 1]); }
        return acc;
    }
};

struct zkg_crs {
    uint32_t n = 0, l = 0, C = 0, log_m = 0; size_t m = 0;
    zk::DevCsr A, B, Cm;
    // Per-window tables (level w = 2^(c w) P_i, level 0 = the query itself): windows share one bucket set, one reduction per
    // multi-exponentiation and no doubling on the host.  The H query gets one over all of its m - 1 points (its scalars are dense).
    // The witness queries stay as uploaded (the flat sums over the ones read them) and get tables over a SUBSET of their elements only:
    // the bucket method sees the non-bit variables of a witness, ~3 % of a credential's, and the same ones proof after proof.
    zk::WindowTable H_query;
    // One proof's H query over several GPUs ("the four MSMs are mutually independent -> task-parallel across GPUs first,
    // then point-sharding"): H is the largest of the four (m - 1 uniformly random scalars, 40 - 50 % of a proof's critical path at
    // m >= 2^18) and shards by points exactly like the plain multi-exponentiation: shard i holds the per-window table of its contiguous
    // slice of the query on devices[i]; a proof copies that slice of coefficients_for_H to it (32 B per point: 4 MiB per shard at m = 2^20
    // over eight devices), every shard runs the single-GPU table launch on a stream of its own, and the partial points are added on the host.
    struct HShard { int device = 0; size_t first = 0, n = 0; zk::WindowTable table; };
    std::vector<HShard> h_shards; int home_device = 0; uint64_t h_epoch = 0;
    zk::DevBuf A_query, B_g1, B_g2, L_query;    // n + 1, n + 1, n + 1 (G2), n - l affine points
    struct SubsetTables {
        std::vector<uint8_t> member;            // host: is element i of z = [1 | w] covered
        zk::DevBuf pos, idx;                    // device: position of every element in the subset (SUBSET_NONE = absent); the covered elements, ascending
        size_t count = 0; uint32_t rebuilds = 0;
        zk::WindowTable A, B1, B2, L;           // count points per level; the L table holds infinity for the constant and the public inputs
    } sub;
    int c_w = 8, c_w_forced = 0;                // window bits of the witness tables (set when they are built), ZKG_TABLE_C_W override
    zk::G1Affine alpha_g1, beta_g1, delta_g1; zk::G2Affine beta_g2, delta_g2;
    CombTable<zk::Fq> alpha1_comb, beta1_comb, delta1_comb; CombTable<zk::Fq2> delta2_comb;
    zk::NttDomain *dom = nullptr;               // basic_radix2_domain (m = 2^log_m) ...
    zk::StepDomain *sdom = nullptr;             // ... or step_radix2_domain (m = 2^(log_m-1) + 2^b); exactly one is set
    zk::DevBuf coset_over_m, coset_over_m29;    // g^i / m : iFFT post-scale fused with the next cosetFFT's pre-scale (and its 29-bit records)
    zk::DevBuf long_rows; uint32_t n_long = 0;  // (matrix << 30 | row) of every row with more than LONG_ROW terms
    zk::Fr z_inv_coset;                         // 1 / (g^m - 1)
    // Prover slots.  A lone caller only ever uses slot 0 (created with the key).  When callers arrive on several threads, a second and a
    // third slot are created on first need: from 8 payloads up one proof fills the chip and a second in flight adds 4 %, but a one-payload
    // proof is latency chains and three in flight give 1.6x the proofs per second (tools/prove_throughput.py).  The witness tables are
    // shared: a proof that has to extend them waits, holding its slot, until every other proof in flight has either finished or is waiting
    // for the same reason, and no new proof starts meanwhile (`extending`, `waiting_ext`).
    static constexpr int MAX_SLOTS = 3;
    ProverSlot slot[MAX_SLOTS];
    float stage_ms[8] = {0};
    std::mutex mu; std::condition_variable cv;
    bool busy[MAX_SLOTS] = {false, false, false}; int leases = 0, waiting_ext = 0; bool extending = false;
};

namespace zk {

// <A_i,z>, <B_i,z>, <C_i,z> for every constraint row.  Rows are short (1-3 terms) except packing and 32-bit-addition rows
// (up to 253 terms): a lane walking such a row alone would set the latency of the whole stage, so rows longer than
// LONG_ROW terms are left to k_r1cs_long (one wavefront per row, lanes stride over the terms, shuffle reduction).
// Rows C..C+l of aA carry the input-consistency terms (r1cs_to_qap_witness_map).
static constexpr uint32_t LONG_ROW = 24;

ZK_D Fr row_dot_short(const uint32_t *rp, const uint32_t *col, const Fr *val, const Fr *z, size_t i, bool &is_long) {
    Fr acc = Fr::zero();
    uint32_t k = rp[i], e = rp[i + 1];
    if (e - k > LONG_ROW) { is_long = true; return acc; }  // filled in by k_r1cs_long
    for (; k < e; ++k) acc += val[k] * z[col[k]];
    return acc;
}
// Device-scope atomics only (no fence: a release fence writes back the XCD's whole L2, and one per workgroup behind the mat-vec's 96 MB of
// fresh output doubled that stage at 2^20): the OR returns its old value and the lane waits for it, so it has been performed before the
// workgroup's barrier and ticket.
ZK_D void or_and_wait(uint32_t *word) { const uint32_t old = atomicOr(word, 1u); asm volatile("" : : "v"(old) : "memory"); }
// one wavefront per long row entry ((matrix << 30) | row): lanes stride over the terms, shuffle reduction
ZK_D void r1cs_long_entry(uint32_t e, uint32_t lane, const uint32_t *a_rp, const uint32_t *a_col, const Fr *a_val, const uint32_t *b_rp, const uint32_t *b_col, const Fr *b_val,
                          const uint32_t *c_rp, const uint32_t *c_col, const Fr *c_val, const Fr *z, Fr *aA, Fr *aB, Fr *aC) {
    const uint32_t mtx = e >> 30, row = e & 0x3fffffffu;
    const uint32_t *rp = mtx == 0 ? a_rp : mtx == 1 ? b_rp : c_rp, *col = mtx == 0 ? a_col : mtx == 1 ? b_col : c_col;
    const Fr *val = mtx == 0 ? a_val : mtx == 1 ? b_val : c_val;
    Fr acc = Fr::zero();
    for (uint32_t k = rp[row] + lane; k < rp[row + 1]; k += 64) acc += val[k] * z[col[k]];
    for (int d = 32; d >= 1; d >>= 1) {
        Fr o;
        for (int j = 0; j < 8; ++j) o.v[j] = __shfl_xor(acc.v[j], d, 64);
        acc += o;
    }
    if (lane == 0) (mtx == 0 ? aA : mtx == 1 ? aB : aC)[row] = acc.normalized();
}
// flag (or null): the satisfiability gate for rows without a long side, see below.  long_list (or null): the long row entries are filled in
// by the first `long_blocks` workgroups of the SAME launch — they read z only and write what the others leave alone — instead of
// by a launch of their own behind this one (k_r1cs_long: 12 ... 33 us and a launch boundary in front of the transforms).
__global__ __launch_bounds__(256) void k_r1cs_eval(const uint32_t *a_rp, const uint32_t *a_col, const Fr *a_val,
                                                    const uint32_t *b_rp, const uint32_t *b_col, const Fr *b_val,
                                                    const uint32_t *c_rp, const uint32_t *c_col, const Fr *c_val,
                                                    const Fr *z, uint32_t C, uint32_t l, size_t m, Fr *aA, Fr *aB, Fr *aC, int critical, uint32_t *flag /* or null */,
                                                    const uint32_t *long_list /* or null */, uint32_t n_long, uint32_t long_blocks) {
    crit_wave_priority(critical);
    if (blockIdx.x < long_blocks) {                                          // (first in the grid: their chains start with the launch, not at its tail)
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (wave < n_long) r1cs_long_entry(long_list[wave], threadIdx.x & 63, a_rp, a_col, a_val, b_rp, b_col, b_val, c_rp, c_col, c_val, z, aA, aB, aC);
        return;
    }
    size_t i = (size_t)(blockIdx.x - long_blocks) * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr a = Fr::zero(), b = Fr::zero(), c = Fr::zero();
    bool la = false, lb = false, lc = false;
    if (i < C) {
        a = row_dot_short(a_rp, a_col, a_val, z, i, la);
        b = row_dot_short(b_rp, b_col, b_val, z, i, lb);
        c = row_dot_short(c_rp, c_col, c_val, z, i, lc);
        // the satisfiability gate (snark.cpp:121-124) for this row, here where its three values are in registers: a kernel of its own over
        // the stored vectors stood 14 us (8 payloads) ... 53 us (37) in front of the transforms.  Rows with a long side: k_r1cs_check_rows.
        if (flag && !(la | lb | lc) && a * b != c) or_and_wait(flag);
    } else if (i <= (size_t)C + l) {
        a = z[i - C];
    }
    const bool keep_long = long_list != nullptr;                             // the long sides belong to the launch's first workgroups
    if (!(keep_long && la)) aA[i] = a.normalized();
    if (!(keep_long && lb)) aB[i] = b.normalized();
    if (!(keep_long && lc)) aC[i] = c.normalized();
}
// long rows as a launch of their own (ZKG_CHECK_KERNEL=1: the round-3 sequence eval, long, check)
__global__ __launch_bounds__(256) void k_r1cs_long(const uint32_t *list, uint32_t n_long,
                                                    const uint32_t *a_rp, const uint32_t *a_col, const Fr *a_val,
                                                    const uint32_t *b_rp, const uint32_t *b_col, const Fr *b_val,
                                                    const uint32_t *c_rp, const uint32_t *c_col, const Fr *c_val,
                                                    const Fr *z, Fr *aA, Fr *aB, Fr *aC) {
    uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_long) return;
    r1cs_long_entry(list[wave], lane, a_rp, a_col, a_val, b_rp, b_col, b_val, c_rp, c_col, c_val, z, aA, aB, aC);
}
// flag[0] |= 1 when a row violates <A,z><B,z> = <C,z>: the pb.is_satisfied() gate of snark.cpp:121-124.  The last workgroup to finish
// (ticket in flag[1]) writes the verdict straight into the caller's pinned word: no copy launch between the mat-vec and the transforms.
__global__ __launch_bounds__(256) void k_r1cs_check(const Fr *aA, const Fr *aB, const Fr *aC, uint32_t C, uint32_t *flag, uint32_t *host_flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C && aA[i] * aB[i] != aC[i]) or_and_wait(flag);
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(flag + 1, 1u) == gridDim.x - 1) host_flag[0] = atomicOr(flag, 0u);
}
// the same gate over the rows k_r1cs_long filled in (entries of its list; a row listed twice is checked twice), behind it in stream order —
// k_r1cs_eval has tested every other row — and the verdict to the caller's pinned word
__global__ __launch_bounds__(256) void k_r1cs_check_rows(const uint32_t *list, uint32_t n_long, const Fr *aA, const Fr *aB, const Fr *aC, uint32_t *flag, uint32_t *host_flag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_long) { const uint32_t row = list[j] & 0x3fffffffu; if (aA[row] * aB[row] != aC[row]) or_and_wait(flag); }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(flag + 1, 1u) == gridDim.x - 1) host_flag[0] = atomicOr(flag, 0u);
}

// the same on a step_radix2_domain: Z(g x_i) takes big/small distinct values on the first big points and one on the rest
__global__ __launch_bounds__(256) void k_pointwise_h_step(Fr *aA, const Fr *aB, const Fr *aC, size_t m, size_t big, const Fr *zinv, uint32_t period_mask, Fr zinv_small, int critical) {
    crit_wave_priority(critical);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr zi = i < big ? zinv[i & period_mask] : zinv_small;
    aA[i] = ((aA[i] * aB[i] - aC[i]) * zi).normalized();
}

__global__ void k_set_one(Fr *z) { if (threadIdx.x == 0 && blockIdx.x == 0) z[0] = Fr::one(); }

// Sparse witness upload: a credential's witness is ~97 % zeros and ones, so the host sends one tag byte per variable (0 zero, 1 one,
// 2 listed) and the listed values as (index, value) records — 0.5 MB instead of 17.5 MB at 20 payloads.  The two kernels also ARE the
// multi_exp_with_mixed_addition split of z = [1 | w] (what k_classify derives from a dense witness): k_expand_tags writes the constant,
// the zeros and ones and their tags and clears the proof's counters, k_scatter_full writes the listed values, tags each by its VALUE (a
// listed 0 or 1 is a bit like any other), lists the others and tells the host how many there are — the last workgroup to finish writes
// into the pinned words, so no fill, classify or copy launch stands between the upload and the mat-vec.
// words: [0] satisfiability flag, [1] k_r1cs_check's ticket, [2] k_scatter_full's ticket, [3] bad listed entry; count: [0] listed, [1] a listed element misses the witness tables
static constexpr uint32_t TAG_UNCLAIMED = 3;
__global__ __launch_bounds__(256) void k_expand_tags(const uint8_t *tags, size_t n, Fr *z /* z[0] is the constant */, uint8_t *wtags, uint32_t *words, uint32_t *count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { z[0] = Fr::one(); wtags[0] = 1; words[0] = 0; words[1] = 0; words[2] = 0; words[3] = 0; count[0] = 0; count[1] = 0; }
    if (i >= n) return;
    const bool one = tags[i] == 1;
    z[i + 1] = one ? Fr::one() : Fr::zero();                                   // listed entries are overwritten by k_scatter_full
    wtags[i + 1] = one ? 1 : (tags[i] == 2 ? TAG_UNCLAIMED : 0);               // a tag-2 variable waits for its one listing (never listed: counts as zero; no consumer reads 3 as a bit)
}
// A listed variable's tag byte goes from TAG_UNCLAIMED to the tag of its value exactly once: the second listing of an index finds the byte
// taken and fails the call (it used to put the position into the gather list twice: A / B / L counted z[pos] twice while H saw it once —
// ZKG_OK with an invalid proof).  Byte-wide compare-and-swap on the containing word; the neighbours' bytes may change under it.
ZK_D bool claim_listed_tag(uint8_t *wtags, uint32_t pos, uint32_t tag) {
    uint32_t *w = reinterpret_cast<uint32_t *>(wtags + (pos & ~3u)); const uint32_t sh = 8 * (pos & 3u);
    uint32_t old = atomicOr(w, 0u);
    for (;;) {
        if (((old >> sh) & 0xffu) != TAG_UNCLAIMED) return false;
        const uint32_t want = (old & ~(0xffu << sh)) | (tag << sh), prev = atomicCAS(w, old, want);
        if (prev == old) return true;
        old = prev;
    }
}
__global__ __launch_bounds__(256) void k_scatter_full(const uint32_t *idx, const Fr *vals, size_t cnt, size_t n, Fr *z, const uint8_t *tags, uint8_t *wtags,
                                                      uint32_t *listed, uint32_t *count, uint32_t *words, const uint32_t *subset_pos, uint32_t *host_words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tag = 0, pos = 0;
    if (i < cnt) {
        const uint32_t v = idx[i];
        if (v >= n || tags[v] != 2) or_and_wait(words + 3);                    // a listed index must be in range and tagged 2 (and listed once)
        else {
            const Fr val = vals[i];
            uint32_t any = 0, diff = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { any |= val.v[j]; diff |= val.v[j] ^ FrParams::ONE[j]; }
            tag = any == 0 ? 0u : (diff == 0 ? 1u : 2u);
            pos = v + 1;
            if (claim_listed_tag(wtags, pos, tag)) z[pos] = val;
            else { or_and_wait(words + 3); tag = 0; }                           // listed twice: the call fails (words[3]), nothing is listed again
        }
    }
    const unsigned long long mask = __ballot(tag == 2);
    if (mask) {
        const uint32_t lane = threadIdx.x & 63, leader = (uint32_t)__ffsll((long long)mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
        base = __shfl(base, leader, 64);
        if (tag == 2) {
            listed[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = pos;
            if (subset_pos && subset_pos[pos] == SUBSET_NONE) or_and_wait(count + 1);    // the witness tables do not cover this element (yet)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(words + 2, 1u) == gridDim.x - 1) {
        host_words[1] = atomicOr(count, 0u); host_words[2] = atomicOr(count + 1, 0u); host_words[3] = atomicOr(words + 3, 0u);
    }
}

// H_tmp = (aA . aB - aC) * Zinv  (divide_by_Z_on_coset fused with the pointwise product)
__global__ __launch_bounds__(256) void k_pointwise_h(Fr *aA, const Fr *aB, const Fr *aC, size_t m, Fr zinv, int critical) {
    crit_wave_priority(critical);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    aA[i] = ((aA[i] * aB[i] - aC[i]) * zinv).normalized();
}

static int upload(DevBuf &d, const void *src, size_t bytes) {
    if (d.reserve(bytes ? bytes : 16)) return ZKG_ERROR;
    if (bytes && !hip_ok(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__)) return ZKG_ERROR;
    return ZKG_OK;
}
static int upload_csr(DevCsr &d, const uint32_t *rp, const uint32_t *col, const uint64_t *val, uint32_t rows) {
    if (!rp) { set_error("crs: null CSR"); return ZKG_ERROR; }
    d.nnz = rp[rows];
    if (upload(d.rowptr, rp, (size_t)(rows + 1) * 4) || upload(d.col, col, d.nnz * 4) || upload(d.val, val, d.nnz * 32)) return ZKG_ERROR;
    return ZKG_OK;
}

// ---- host-side proof assembly ---------------------------------------------------------------------
static void canonical_limbs(const Fr &x, uint32_t out[8]) { Fr c = x.from_mont(); for (int i = 0; i < 8; ++i) out[i] = c.v[i]; }
static bool canonical_lsb(const Fq &y) { return y.from_mont().v[0] & 1u; }
static size_t ser_g1(uint8_t *out, const G1 &p) {        // libff operator<<(alt_bn128_G1): binary, Montgomery, compressed
    G1Affine a = p.to_affine(); bool inf = p.is_inf();
    Fq x = inf ? Fq::zero() : a.x, y = inf ? Fq::one() : a.y;
    out[0] = inf ? '1' : '0'; memcpy(out + 1, x.v, 32); out[33] = canonical_lsb(y) ? '1' : '0';
    return 34;
}
static size_t ser_g2(uint8_t *out, const G2 &p) {
    G2Affine a = p.to_affine(); bool inf = p.is_inf();
    Fq2 x = inf ? Fq2::zero() : a.x, y = inf ? Fq2::one() : a.y;
    out[0] = inf ? '1' : '0'; memcpy(out + 1, x.c0.v, 32); memcpy(out + 33, x.c1.v, 32); out[65] = canonical_lsb(y.c0) ? '1' : '0';
    return 66;
}

struct WitnessSrc {                           // dense: n x 4 limbs; or sparse: tags[n] + (idx[count], vals[count x 4 limbs])
    const uint64_t *dense = nullptr;
    const uint8_t *tags = nullptr; const uint32_t *idx = nullptr; const uint64_t *vals = nullptr; size_t count = 0;
};
// phase 1 of r1cs_to_qap_witness_map: z = [1 | w] resident and split, the three mat-vecs, the satisfiability flag
static unsigned floor_log2(size_t x) { unsigned r = 0; while (x >>= 1) ++r; return r; }
static int compute_h_matvec(zkg_crs *crs, ProverSlot &S, const WitnessSrc &W, bool want_flag) {
    const int crit = crit_priority_for(1, floor_log2(crs->m)) ? 1 : 0;   // (see ntt_run_ex)
    static const bool fused_check = getenv("ZKG_CHECK_KERNEL") == nullptr;                                // A/B switch: the satisfiability check as a kernel of its own over the stored vectors
    hipStream_t s = S.stream; uint32_t *flag_out = S.flag_host;
    const size_t m = crs->m;
    Fr *z = S.z.as<Fr>(), *aA = S.aABC.as<Fr>(), *aB = aA + m, *aC = aA + 2 * m;
    uint32_t *words = S.flag.as<uint32_t>(), *count = S.wcount.as<uint32_t>();
    const uint32_t *subset_pos = crs->sub.count ? crs->sub.pos.as<uint32_t>() : nullptr;
    if (W.dense || !crs->n) {
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, s, z);
        if (crs->n) ZK_HIP(hipMemcpyAsync(z + 1, W.dense, (size_t)crs->n * 32, hipMemcpyHostToDevice, s));
        ZK_HIP(hipMemsetAsync(words, 0, 16, s));
        // the multi_exp_with_mixed_addition split of z: tags, the indices of the non-bit elements, and their count (read by the host)
        // (and whether every one of them has its place in the witness tables: flag_host[2])
        if (witness_classify(z, (size_t)crs->n + 1, S.wtags.as<uint8_t>(), S.wlisted.as<uint32_t>(), count, s, subset_pos)) return ZKG_ERROR;
        ZK_HIP(hipMemcpyAsync(S.flag_host + 1, count, 8, hipMemcpyDeviceToHost, s));
    } else {
        const size_t n = crs->n, cnt = W.count;
        if (S.up_tags.reserve(n) || S.up_idx.reserve(cnt * 4 + 16) || S.up_vals.reserve(cnt * 32 + 16)) return ZKG_ERROR;
        ZK_HIP(hipMemcpyAsync(S.up_tags.p, W.tags, n, hipMemcpyHostToDevice, s));
        if (cnt) {
            ZK_HIP(hipMemcpyAsync(S.up_idx.p, W.idx, cnt * 4, hipMemcpyHostToDevice, s));
            ZK_HIP(hipMemcpyAsync(S.up_vals.p, W.vals, cnt * 32, hipMemcpyHostToDevice, s));
        }
        hipLaunchKernelGGL(k_expand_tags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S.up_tags.as<uint8_t>(), n, z, S.wtags.as<uint8_t>(), words, count);
        if (cnt) hipLaunchKernelGGL(k_scatter_full, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, S.up_idx.as<uint32_t>(), S.up_vals.as<Fr>(), cnt, n, z,
                                    S.up_tags.as<uint8_t>(), S.wtags.as<uint8_t>(), S.wlisted.as<uint32_t>(), count, words, subset_pos, S.flag_host);
    }
    if (S.ev_ok) (void)hipEventRecord(S.ev[0], s);                      // z = [1 | w] is resident and split from here on
    // ZKG_LONG_MERGED=1: the long rows' workgroups lead k_r1cs_eval's grid instead of being a launch of their own.  The stage gets shorter (8 payloads
    // 0.069 -> 0.055 ms, 37 payloads 0.18 -> 0.13) and the proof does not: at 8 payloads it gets LONGER (1.13 -> 1.15 ms, three alternating runs) — the
    // witness jobs' first small kernels used to slip onto the chip during the short launches in front of the first transform, and now find
    // its workgroups holding every CU's LDS — and at 37 payloads it is inside the noise (3.06 against 3.08).  Off.
    static const bool long_merged = getenv("ZKG_LONG_MERGED") != nullptr;
    const unsigned eval_blocks = (unsigned)((m + 255) / 256), long_blocks = (fused_check && long_merged) ? (crs->n_long + 3) / 4 : 0;
    hipLaunchKernelGGL(k_r1cs_eval, dim3(eval_blocks + long_blocks), dim3(256), 0, s,
                       crs->A.rowptr.as<uint32_t>(), crs->A.col.as<uint32_t>(), crs->A.val.as<Fr>(),
                       crs->B.rowptr.as<uint32_t>(), crs->B.col.as<uint32_t>(), crs->B.val.as<Fr>(),
                       crs->Cm.rowptr.as<uint32_t>(), crs->Cm.col.as<uint32_t>(), crs->Cm.val.as<Fr>(),
                       z, crs->C, crs->l, m, aA, aB, aC, crit, (want_flag && fused_check) ? words : nullptr,
                       long_blocks ? crs->long_rows.as<uint32_t>() : nullptr, crs->n_long, long_blocks);
    if (crs->n_long && !long_blocks)
        hipLaunchKernelGGL(k_r1cs_long, dim3((crs->n_long + 3) / 4), dim3(256), 0, s, crs->long_rows.as<uint32_t>(), crs->n_long,
                           crs->A.rowptr.as<uint32_t>(), crs->A.col.as<uint32_t>(), crs->A.val.as<Fr>(),
                           crs->B.rowptr.as<uint32_t>(), crs->B.col.as<uint32_t>(), crs->B.val.as<Fr>(),
                           crs->Cm.rowptr.as<uint32_t>(), crs->Cm.col.as<uint32_t>(), crs->Cm.val.as<Fr>(), z, aA, aB, aC);
    if (want_flag) {
        if (crs->C && fused_check) hipLaunchKernelGGL(k_r1cs_check_rows, dim3(std::max<uint32_t>(1, (crs->n_long + 255) / 256)), dim3(256), 0, s, crs->long_rows.as<uint32_t>(), crs->n_long, aA, aB, aC, words, flag_out);
        else if (crs->C) hipLaunchKernelGGL(k_r1cs_check, dim3((crs->C + 255) / 256), dim3(256), 0, s, aA, aB, aC, crs->C, words, flag_out);
        if (S.ev_ok) (void)hipEventRecord(S.ev[3], s);
    }
    if (S.ev_ok) (void)hipEventRecord(S.ev[1], s);
    if (hipGetLastError() != hipSuccess) { set_error("prover kernel launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}
// phase 2: the seven transforms and the pointwise step -> coefficients_for_H in aA
static int compute_h_transforms(zkg_crs *crs, ProverSlot &S) {
    const int crit = crit_priority_for(1, floor_log2(crs->m)) ? 1 : 0;   // (see ntt_run_ex)
    hipStream_t s = S.stream;
    const size_t m = crs->m;
    Fr *aA = S.aABC.as<Fr>(), *aB = aA + m, *aC = aA + 2 * m;
    Fr *scr = S.ntt_scratch.as<Fr>();
    const unsigned grid_m = (unsigned)((m + 255) / 256);
    if (crs->dom) {
        // iFFT then cosetFFT of aA, aB, aC as ONE batch of three (a single 2^18 transform fills half the chip; three fill it):
        // inverse transform with the fused post table g^i/m, then a plain forward transform
        const Fr *fused = crs->coset_over_m.as<Fr>();
        if (ntt_run_ex(crs->dom, aA, true, nullptr, fused, nullptr, s, scr, 3, 0, nullptr, crs->coset_over_m29.p)) return ZKG_ERROR;
        if (ntt_run_ex(crs->dom, aA, false, nullptr, nullptr, nullptr, s, scr, 3)) return ZKG_ERROR;
        hipLaunchKernelGGL(k_pointwise_h, dim3(grid_m), dim3(256), 0, s, aA, aB, aC, m, crs->z_inv_coset, crit);
        if (ntt_run_ex(crs->dom, aA, true, nullptr, crs->dom->icoset_post.as<Fr>(), nullptr, s, scr)) return ZKG_ERROR;   // icosetFFT -> coefficients_for_H[0..m)
    } else {
        // step_radix2_domain: the same four steps, each transform a fold/unfold pass around a 2^a and a 2^b radix-2 transform
        StepDomain *sd = crs->sdom;
        if (step_ntt_run(sd, aA, true, false, s, scr, 3, m)) return ZKG_ERROR;                     // iFFT  x3
        if (step_ntt_run(sd, aA, false, true, s, scr, 3, m)) return ZKG_ERROR;                     // cosetFFT x3
        hipLaunchKernelGGL(k_pointwise_h_step, dim3(grid_m), dim3(256), 0, s, aA, aB, aC, m, sd->shape.big, sd->zinv.as<Fr>(),
                           (uint32_t)(sd->shape.big / sd->shape.small - 1), sd->zinv_small, crit);
        if (step_ntt_run(sd, aA, true, true, s, scr, 1, m)) return ZKG_ERROR;                      // icosetFFT
    }
    if (S.ev_ok) (void)hipEventRecord(S.ev[2], s);
    if (hipGetLastError() != hipSuccess) { set_error("prover kernel launch failed"); return ZKG_ERROR; }
    return ZKG_OK;
}

static constexpr size_t H_ROW_MERGE_MIN = 49152;                               // points from which H's windows share rows of buckets in pairs (slot_create)
int table_window_bits(size_t n) {                                    // window size of a query's table, by the size of the query
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) ++lg;
    // (re-measured in round 3 with the 29-bit kernels, tools/ch_sweep.sh + tools/prove_throughput.py: one payload, 2^15 - 1 points: c = 12 / 13 / 15 / 16
    //  -> 0.75 / 0.71 / 0.67 / 0.71 ms for a lone caller but 1907 / 1874 / 1632 / 1539 proofs/s with three callers, and no difference through
    //  the seam, whose witness pass dominates there: 17 windows x 2^14 buckets cost chip time that other proofs could use.  12 stays.)
    return lg >= 15 ? 16 : lg >= 9 ? 12 : 8;
}
static int slot_create(zkg_crs *crs, ProverSlot &S) {
    if (S.ready) return ZKG_OK;
    const size_t n = crs->n, m = crs->m;
    bool ok = S.z.reserve((n + 1) * 32) == 0 && S.aABC.reserve(3 * m * 32) == 0 && S.flag.reserve(16) == 0 && S.ntt_scratch.reserve(3 * m * NTT_SCRATCH_BYTES) == 0 &&
              S.wtags.reserve(n + 8) == 0 && S.wlisted.reserve((n + 1) * 4) == 0 && S.wcount.reserve(8) == 0 &&
              hip_ok(hipHostMalloc((void **)&S.flag_host, 64, hipHostMallocDefault), "hipHostMalloc", __FILE__, __LINE__);
    if (ok) {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);             // numerically lower = higher priority
        // Four streams, so that each can have a hardware queue (a compute pipe) of its own — with more, two streams share a pipe and a
        // small kernel waits until every workgroup of the other stream's big kernel has been dispatched (seen in the kernel trace: the G1
        // witness job started 1.3 ms into a 1.9 ms proof):
        //   stream    upload, split, mat-vec, 7 NTTs and the H multi-exponentiation (H needs the NTTs anyway): the critical path
        //   job_w2    B_g2 over the non-bit witness elements (the longest latency chain)
        //   job_w1    A, B_g1 and L over the same elements, batched
        //   stream_o  the flat sums over the ones: B_g2, then A / B_g1 / L batched
        // ZKG_PRIO (tuning aid): bit 0 = `stream` high priority, bit 1 = witness jobs high, bit 2 = ones-sums high.
        static const char *pe = getenv("ZKG_PRIO");
        const int pr = pe ? atoi(pe) : 1;
        auto prio = [&](int bit) { return (pr >> bit) & 1 ? prio_hi : prio_lo; };
        ok = hip_ok(hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, prio(0)), "hipStreamCreate", __FILE__, __LINE__);
        S.job_w1 = msm_job_create(nullptr, true, prio(1) == prio_hi); S.job_w2 = msm_job_create(nullptr, true, prio(1) == prio_hi);
        S.job_h = ok ? msm_job_create(S.stream, false) : nullptr;
        ok = ok && S.job_w1 && S.job_w2 && S.job_h &&
             hip_ok(hipStreamCreateWithPriority(&S.stream_o, hipStreamNonBlocking, prio(2)), "hipStreamCreate", __FILE__, __LINE__);
        if (ok) {                                                            // a table launch runs at the table's window size
            msm_job_set_window(S.job_w1, crs->c_w); msm_job_set_window(S.job_w2, crs->c_w); msm_job_set_window(S.job_h, crs->H_query.c);
            msm_job_set_critical(S.job_h, crit_priority_for(4, floor_log2(crs->m)));                              // the H query's multi-exponentiation ends the proof; the witness jobs beside it have slack
            // H: sixteen windows in eight rows of buckets from 49152 points on — where the unmerged launch already takes the two-pass sort
            // (2 / 4 / 8 payloads: 0.78 -> 0.72, 0.93 -> 0.88, 1.22 -> 1.17 ms; four rows at 8 payloads: 1.26 — one round of lanes, the
            // longest chain sets the time; one payload, 2^15 points: 0.75 -> 0.89, the doubled rows leave the one-pass sort's range)
            { static const int force = getenv("ZKG_H_ROW_MERGE") ? atoi(getenv("ZKG_H_ROW_MERGE")) : 0; msm_job_set_row_merge(S.job_h, force ? (uint32_t)force : (crs->H_query.n >= H_ROW_MERGE_MIN ? 2u : 1u)); }
        }
    }
    if (ok) {
        S.ev_ok = true;
        for (auto &e : S.ev) if (hipEventCreate(&e) != hipSuccess) S.ev_ok = false;
        if (!S.ev_ok) { set_error("hipEventCreate failed"); ok = false; }
    }
    if (ok) S.helper.start();
    S.ready = ok;
    return ok ? ZKG_OK : ZKG_ERROR;
}
static void slot_destroy(ProverSlot &S) {
    S.helper.shutdown();
    for (DevBuf *b : {&S.z, &S.aABC, &S.flag, &S.ntt_scratch, &S.up_tags, &S.up_idx, &S.up_vals, &S.wtags, &S.wlisted, &S.wcount}) b->release();
    msm_job_destroy(S.job_w1); msm_job_destroy(S.job_w2); msm_job_destroy(S.job_h);
    S.job_w1 = S.job_w2 = S.job_h = nullptr;
    for (OnesSum *o : {&S.ones_g1, &S.ones_g2}) o->release();
    for (hipStream_t *st : {&S.stream, &S.stream_o}) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
    if (S.ev_ok) for (auto &e : S.ev) (void)hipEventDestroy(e);
    S.ev_ok = false;
    if (S.flag_host) (void)hipHostFree(S.flag_host);
    S.flag_host = nullptr; S.ready = false;
}

}  // namespace zk

using namespace zk;

extern "C" {

static zkg_crs *zkg_crs_upload_impl(const zkg_pk *pk, bool queries_on_device = false, const std::function<bool()> &constraint_system_ready = nullptr) {
    if (!pk) { set_error("zkg_crs_upload: null pk"); return nullptr; }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("zkg_crs_upload: no HIP device (call zkg_init)"); return nullptr; }
    const zkg_r1cs &cs = pk->cs;
    DomainShape shape;
    if (pk->log_m > 28 || !domain_shape_of(pk->domain_size ? (size_t)pk->domain_size : ((size_t)1 << pk->log_m), shape) || shape.log_m != pk->log_m ||
        ((size_t)cs.num_constraints + cs.num_inputs + 1) > shape.m || cs.num_inputs > cs.num_variables) {
        set_error("zkg_crs_upload: inconsistent sizes (domain must be 2^log_m or a step_radix2 size 2^(log_m-1) + 2^b)"); return nullptr;
    }
    auto t_begin = std::chrono::steady_clock::now();
    static const bool dbg_timing = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto lap = [&](const char *what) { if (dbg_timing) fprintf(stderr, "[zkg key upload] %-24s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    zkg_crs *crs = new zkg_crs();
    crs->n = cs.num_variables; crs->l = cs.num_inputs; crs->C = cs.num_constraints; crs->log_m = pk->log_m; crs->m = shape.m;
    const size_t n = crs->n, l = crs->l, m = crs->m;
    bool ok = true;
    {
        // H becomes a per-window table (level 0 is the query as uploaded); A, B_g1, B_g2 and L go up as they are — their tables are built
        // over the elements the first proofs list (subset_extend).  They are indexed by the same witness and share one digit sort per
        // proof, so they share one window size; H has its own.
        static const char *cw = getenv("ZKG_TABLE_C_W"), *ch = getenv("ZKG_TABLE_C_H");                  // tuning aids
        // The witness tables' window is chosen when they are built, from the number of elements they cover (subset_extend); 0 = that rule.
        const int c_w = cw ? atoi(cw) : 0, c_h = ch ? atoi(ch) : table_window_bits(m - 1);
        crs->c_w_forced = c_w;
        // a query: from the host (zkg_pk as the ABI hands it over) or already on the device (the blob path decompresses there)
        auto query = [&](DevBuf &q, const uint64_t *src, size_t count, bool g2) {
            const size_t bytes = count * (g2 ? 128 : 64);
            if (q.reserve(bytes + 16)) return false;
            return !bytes || hip_ok(hipMemcpy(q.p, src, bytes, queries_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice), "query upload", __FILE__, __LINE__);
        };
        ok = (c_w == 0 || (c_w >= 2 && c_w <= 16)) && c_h >= 2 && c_h <= 16 &&
             query(crs->A_query, pk->A_query, n + 1, false) && query(crs->B_g1, pk->B_g1, n + 1, false) && query(crs->B_g2, pk->B_g2, n + 1, true) &&
             query(crs->L_query, pk->L_query, n - l, false);
        if (ok) {
            DevBuf stage; const G1Affine *h_dev = (const G1Affine *)pk->H_query;
            if (!queries_on_device) { ok = upload(stage, pk->H_query, (m - 1) * 64) == 0; h_dev = stage.as<G1Affine>(); }
            static const bool h32 = getenv("ZKG_ACCUM_32") != nullptr;                          // (A/B switch: no 29-bit records)
            ok = ok && window_table_build_g1(crs->H_query, h_dev, m - 1, c_h, nullptr) == 0 && (h32 || window_table_records29(crs->H_query, nullptr) == 0) &&
                 hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__);
            stage.release();
        }
    }
    lap("queries up, H table built");
    if (ok) {
        memcpy(&crs->alpha_g1, pk->alpha_g1, 64); memcpy(&crs->beta_g1, pk->beta_g1, 64); memcpy(&crs->delta_g1, pk->delta_g1, 64);
        memcpy(&crs->beta_g2, pk->beta_g2, 128); memcpy(&crs->delta_g2, pk->delta_g2, 128);
        host_parallel_for(4, [&](int i) {
            if (i == 0) crs->delta2_comb.build(crs->delta_g2); else if (i == 1) crs->delta1_comb.build(crs->delta_g1);
            else if (i == 2) crs->alpha1_comb.build(crs->alpha_g1); else crs->beta1_comb.build(crs->beta_g1);
        });
        if (shape.step) crs->sdom = step_domain(shape.m, nullptr); else crs->dom = ntt_domain(pk->log_m, nullptr);
        ok = crs->dom != nullptr || crs->sdom != nullptr;
    }
    if (ok && crs->dom) {
        Fr g = Fr::from_u64(5);
        crs->z_inv_coset = (g.pow_u64(m) - Fr::one()).inverse();           // basic_radix2_domain::divide_by_Z_on_coset
        ok = crs->coset_over_m.reserve(m * 32) == 0 && powers_table(crs->coset_over_m.as<Fr>(), m, g, crs->dom->n_inv, nullptr) == 0 &&
             ntt_table29(crs->coset_over_m29, crs->coset_over_m.as<Fr>(), m, nullptr) == 0;
    }
    lap("comb tables + domain");
    ok = ok && hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__) && slot_create(crs, crs->slot[0]) == ZKG_OK;
    lap("prover slot created");
    // the constraint system last — behind the comb tables, the domain and the prover slot, which do not need it: the blob path parses it on another
    // thread meanwhile (8 payloads: the parse ends 2.4 ms after the H table stands; what used to follow it was 4.8 ms, now 1.2)
    if (ok && constraint_system_ready && !constraint_system_ready()) ok = false;
    ok = ok && upload_csr(crs->A, cs.a_rowptr, cs.a_col, cs.a_val, crs->C) == 0 && upload_csr(crs->B, cs.b_rowptr, cs.b_col, cs.b_val, crs->C) == 0 &&
         upload_csr(crs->Cm, cs.c_rowptr, cs.c_col, cs.c_val, crs->C) == 0;
    if (ok) {                                                                // rows left to the wavefront-per-row kernel
        std::vector<uint32_t> lr;
        const uint32_t *rps[3] = {cs.a_rowptr, cs.b_rowptr, cs.c_rowptr};
        for (uint32_t mtx = 0; mtx < 3; ++mtx)
            for (uint32_t r = 0; r < crs->C; ++r) if (rps[mtx][r + 1] - rps[mtx][r] > LONG_ROW) lr.push_back((mtx << 30) | r);
        crs->n_long = (uint32_t)lr.size();
        ok = crs->C < (1u << 30) && upload(crs->long_rows, lr.data(), lr.size() * 4) == 0;
    }
    ok = ok && hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__);
    lap("constraint system up");
    if (!ok) { zkg_crs_free(crs); return nullptr; }
    return crs;
}
zkg_crs *zkg_crs_upload(const zkg_pk *pk) {
    try { return zkg_crs_upload_impl(pk); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("zkg_crs_upload: ") + e.what()); return nullptr; }
    catch (...) { zk::set_error("zkg_crs_upload: unexpected exception"); return nullptr; }
}


static void slot_release_h_runs(ProverSlot &S) {                              // leaves the current device changed; callers restore it
    for (auto &r : S.h_runs) {
        (void)hipSetDevice(r.device);
        if (r.job) { (void)hipStreamSynchronize(msm_job_stream(r.job)); msm_job_destroy(r.job); }
        r.scalars.release();
    }
    S.h_runs.clear();
}
void zkg_crs_free(zkg_crs *crs) {
    if (!crs) return;
    for (DevBuf *b : {&crs->A.rowptr, &crs->A.col, &crs->A.val, &crs->B.rowptr, &crs->B.col, &crs->B.val, &crs->Cm.rowptr, &crs->Cm.col, &crs->Cm.val,
                      &crs->coset_over_m, &crs->coset_over_m29, &crs->long_rows})
        b->release();
    for (WindowTable *t : {&crs->H_query, &crs->sub.A, &crs->sub.B1, &crs->sub.B2, &crs->sub.L}) t->release();
    for (DevBuf *b : {&crs->A_query, &crs->B_g1, &crs->B_g2, &crs->L_query, &crs->sub.pos, &crs->sub.idx}) b->release();
    int cur = 0; (void)hipGetDevice(&cur);
    for (ProverSlot &S : crs->slot) slot_release_h_runs(S);
    for (auto &sh : crs->h_shards) { (void)hipSetDevice(sh.device); sh.table.release(); }
    (void)hipSetDevice(cur);
    for (ProverSlot &S : crs->slot) slot_destroy(S);
    delete crs;
}

uint32_t zkg_crs_num_variables(const zkg_crs *crs) { return crs ? crs->n : 0; }

// the largest number of proofs any one key has had in flight at once since the last reset (zkg_prover_peak_in_flight): what a test of
// "callers run side by side" can assert without a stopwatch
static std::atomic<int> g_peak_in_flight{0};
// A caller's hold on one prover slot of a key (see zkg_crs): blocks until a slot is free and no table extension is pending.
struct SlotLease {
    zkg_crs *crs; int i = -1;
    explicit SlotLease(zkg_crs *c) : crs(c) {
        // how many proofs of this key may be in flight: measured proofs per second with 1 / 2 / 3 callers — one payload (m = 2^15) 1069 /
        // 1142 / 1706, eight payloads (2^18) 609 / 698 / 576, 37 payloads (2^20) 250 / 247.  ZKG_PROVER_SLOTS caps it (1 = callers queue).
        static const int env_slots = [] { const char *e = getenv("ZKG_PROVER_SLOTS"); int v = e ? atoi(e) : zkg_crs::MAX_SLOTS; return v < 1 ? 1 : v > zkg_crs::MAX_SLOTS ? zkg_crs::MAX_SLOTS : v; }();
        const int max_slots = std::min(env_slots, c->m < ((size_t)1 << 18) ? 3 : c->m < ((size_t)1 << 19) ? 2 : 1);
        std::unique_lock<std::mutex> lk(crs->mu);
        for (;;) {
            if (!crs->extending && crs->waiting_ext == 0) {
                int pick = -1;
                for (int k = 0; k < max_slots && pick < 0; ++k) if (!crs->busy[k] && crs->slot[k].ready) pick = k;
                for (int k = 0; k < max_slots && pick < 0; ++k) if (!crs->busy[k]) pick = k;              // not created yet
                if (pick >= 0) {
                    crs->busy[pick] = true; ++crs->leases;
                    { int seen = g_peak_in_flight.load(std::memory_order_relaxed); while (crs->leases > seen && !g_peak_in_flight.compare_exchange_weak(seen, crs->leases)) {} }
                    if (crs->slot[pick].ready) { i = pick; return; }
                    lk.unlock();
                    const int rc = slot_create(crs, crs->slot[pick]);                                     // device allocations: outside the lock
                    lk.lock();
                    if (rc == ZKG_OK) { i = pick; return; }
                    slot_destroy(crs->slot[pick]);
                    crs->busy[pick] = false; --crs->leases; crs->cv.notify_all();
                    if (pick == 0) return;                                                                // not even one slot: give up (i stays -1)
                    // no memory for another slot: wait for one of the existing ones
                    crs->cv.wait(lk, [&] { for (int k = 0; k < max_slots; ++k) if (!crs->busy[k] && crs->slot[k].ready) return true; return false; });
                    continue;
                }
            }
            crs->cv.wait(lk);
        }
    }
    ~SlotLease() {
        if (i < 0) return;
        std::lock_guard<std::mutex> lk(crs->mu);
        crs->busy[i] = false; --crs->leases; crs->cv.notify_all();
    }
    bool ok() const { return i >= 0; }
    ProverSlot &S() { return crs->slot[i]; }
};

static int zkg_qap_witness_h_impl(const zkg_crs *crs_, const uint64_t *witness, uint64_t *h_out) {
    zkg_crs *crs = const_cast<zkg_crs *>(crs_);
    if (!crs || !h_out || (crs->n && !witness)) { set_error("zkg_qap_witness_h: bad argument"); return ZKG_ERROR; }
    SlotLease lease(crs);
    if (!lease.ok()) return ZKG_ERROR;
    ProverSlot &S = lease.S();
    WitnessSrc W; W.dense = witness;
    if (compute_h_matvec(crs, S, W, false) || compute_h_transforms(crs, S)) return ZKG_ERROR;
    ZK_HIP(hipStreamSynchronize(S.stream));
    ZK_HIP(hipMemcpy(h_out, S.aABC.p, crs->m * 32, hipMemcpyDeviceToHost));
    memset(h_out + 4 * crs->m, 0, 32);                                      // coefficients_for_H[m] = 0
    return ZKG_OK;
}
int zkg_qap_witness_h(const zkg_crs *crs_, const uint64_t *witness, uint64_t *h_out) {
    try { return zkg_qap_witness_h_impl(crs_, witness, h_out); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("zkg_qap_witness_h: ") + e.what()); return ZKG_ERROR; }
    catch (...) { zk::set_error("zkg_qap_witness_h: unexpected exception"); return ZKG_ERROR; }
}


// ---- one proof = prove_enqueue (everything the GPU does, queued without waiting) + prove_finish (host tails, assembly, bytes)
static const bool g_dbg_timing = getenv("ZKG_DEBUG_TIMING") != nullptr, g_serial_msm = getenv("ZKG_SERIAL_MSM") != nullptr;
// ZKG_WITNESS_START (tuning aid): which event the witness streams wait for — 0 the split (default), 1 the mat-vec, 2 the transforms
static const int g_witness_start = [] { const char *e = getenv("ZKG_WITNESS_START"); int v = e ? atoi(e) : 0; return v >= 0 && v <= 2 ? v : 0; }();
static void lap(const ProverSlot &S, const char *what) {
    if (g_dbg_timing) fprintf(stderr, "[zkg] %-22s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S.t0).count());
}
static MsmBases table_set(const WindowTable &t, bool g2, uint32_t index_sub, const uint32_t *remap = nullptr) {       // (a table not built yet is empty, not G1)
    MsmBases b; b.p = t.buf.p; b.g2 = g2; b.level_stride = t.n; b.index_sub = index_sub; b.remap = remap; b.p29 = t.rec29.p; return b;
}
static MsmBases query_set(const DevBuf &q, bool g2, uint32_t index_sub) { MsmBases b; b.p = q.p; b.g2 = g2; b.index_sub = index_sub; return b; }

// The witness tables grow to cover `listed` more elements (the first proof on a key; later only when a witness has a non-bit value where
// every earlier one had a bit).  Host: membership -> ascending element list and positions; device: level 0 gathered from the queries, then
// the levels.  Runs on the helper thread with the ones-sum stream, which has nothing of this proof queued yet.
static int subset_extend(zkg_crs *crs, ProverSlot &S, size_t listed) {
    zkg_crs::SubsetTables &T = crs->sub;
    const size_t n1 = (size_t)crs->n + 1; hipStream_t s = S.stream_o;
    // From here until one of the two success exits the key has NO witness tables: pos / idx / the tables / the window size are rewritten
    // below in several steps, and a failure between them (a device allocation under memory pressure) must leave a state the next proof
    // recognises — count == 0 sends it back here — instead of old tables under a new position map.
    T.count = 0;
    const auto t_ext0 = std::chrono::steady_clock::now();
    auto ext_lap = [&](const char *w) { if (g_dbg_timing) fprintf(stderr, "[zkg]       subset_extend %-28s %8.3f ms\n", w, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_ext0).count()); };
    std::vector<uint32_t> li(listed);
    ZK_HIP(hipMemcpyAsync(li.data(), S.wlisted.p, listed * 4, hipMemcpyDeviceToHost, s));
    ZK_HIP(hipStreamSynchronize(s));
    if (T.member.empty()) T.member.assign(n1, 0);
    for (uint32_t i : li) { if (i >= n1) { set_error("prover: witness split out of range"); return ZKG_ERROR; } T.member[i] = 1; }
    std::vector<uint32_t> pos(n1, SUBSET_NONE), idx;
    idx.reserve(listed + T.count);
    for (size_t i = 0; i < n1; ++i) if (T.member[i]) { pos[i] = (uint32_t)idx.size(); idx.push_back((uint32_t)i); }
    const size_t count = idx.size();
    ext_lap("listed read, positions built");
    if (T.pos.reserve(n1 * 4) || T.idx.reserve(count * 4 + 16)) return ZKG_ERROR;
    ZK_HIP(hipMemcpyAsync(T.pos.p, pos.data(), n1 * 4, hipMemcpyHostToDevice, s));
    ZK_HIP(hipMemcpyAsync(T.idx.p, idx.data(), count * 4, hipMemcpyHostToDevice, s));
    // Window size: only the non-bit elements of a witness reach the bucket method — a credential's values, packings and public inputs,
    // ~30 per payload, a thousand of a million variables at 37 payloads — so the window is sized for `count` entries per window, about one
    // per bucket: c = log2(count) + 2 within [10, 14] (measured flat between 10 and 14 at 8 and 37 payloads, 3 - 5 % slower at 16).  Small
    // bucket sets keep this latency-bound work light: the digit sort's LDS histogram is 2^(c-1) counters, the fold and the reduction
    // shrink with it.
    int lg_count = 0; while (((size_t)1 << (lg_count + 1)) <= count) ++lg_count;
    const int c_new = crs->c_w_forced ? crs->c_w_forced : std::min(14, std::max(10, lg_count + 2));     // becomes crs->c_w once the tables stand
    ScopedDevBuf stage, stage2;                                                                       // released on every way out
    if (stage.reserve(3 * count * sizeof(G1Affine) + 16) || stage2.reserve(count * sizeof(G2Affine) + 16)) return ZKG_ERROR;
    const uint32_t *d_idx = T.idx.as<uint32_t>();
    static const size_t host_limit = getenv("ZKG_SUBSET_HOST_LIMIT") ? (size_t)atoi(getenv("ZKG_SUBSET_HOST_LIMIT")) : 1024;   // tuning aid
    if (count <= host_limit) {
        // A small subset (every credential size zklaim's benchmark covers): on the GPU each table is a chain of 254 doublings and W / 4
        // inversions per lane whatever the count — 10 ms for 22 points; the host pool walks the same chains at 0.08 ms per G1 point (0.25 ms
        // per G2 point), a task per (table, point), and uploads the finished levels.  Same-box A/B: 8 ms less at 12 payloads (324
        // elements), 6 ms less at 20 (552); the costs meet around a thousand elements.
        G1Affine *d1 = stage.as<G1Affine>();
        int rc = gather_points_g1(crs->A_query.as<G1Affine>(), d_idx, count, 0, d1, s) || gather_points_g1(crs->B_g1.as<G1Affine>(), d_idx, count, 0, d1 + count, s) ||
                 gather_points_g1(crs->L_query.as<G1Affine>(), d_idx, count, (uint32_t)(crs->l + 1), d1 + 2 * count, s) ||
                 gather_points_g2(crs->B_g2.as<G2Affine>(), d_idx, count, 0, stage2.as<G2Affine>(), s);
        std::vector<G1Affine> p1(3 * count); std::vector<G2Affine> p2(count);
        const bool got = !rc && hip_ok(hipMemcpyAsync(p1.data(), d1, 3 * count * sizeof(G1Affine), hipMemcpyDeviceToHost, s), "D2H", __FILE__, __LINE__) &&
                         hip_ok(hipMemcpyAsync(p2.data(), stage2.p, count * sizeof(G2Affine), hipMemcpyDeviceToHost, s), "D2H", __FILE__, __LINE__) &&
                         hip_ok(hipStreamSynchronize(s), "sync", __FILE__, __LINE__);
        stage.release(); stage2.release();
        if (!got) return ZKG_ERROR;
        ext_lap("points gathered and read back");
        const int c = c_new, W = (254 + c) / c;                               // = (SCALAR_BITS + c - 1) / c of window_table_build
        std::vector<G1Affine> t1((size_t)3 * W * count); std::vector<G2Affine> t2((size_t)W * count);
        auto levels = [&](auto base, auto *out /* level w of this point at out[w * count] */) {
            typedef decltype(base.x) F;
            if (base.is_inf()) { for (int w = 0; w < W; ++w) out[(size_t)w * count] = base; return; }
            std::vector<XYZZ<F>> lv(W);
            XYZZ<F> p = XYZZ<F>::from_affine(base);
            for (int w = 1; w < W; ++w) { for (int k = 0; k < c; ++k) p = p.dbl(); lv[w] = p; }
            std::vector<F> pre(W); F run = F::one();
            for (int w = 1; w < W; ++w) { pre[w] = run; if (!lv[w].is_inf()) run = run * lv[w].zzz; }
            F inv = run.inverse();
            out[0] = base;
            for (int w = W - 1; w >= 1; --w) {
                if (lv[w].is_inf()) { out[(size_t)w * count] = decltype(base)::inf(); continue; }
                F zi = inv * pre[w]; inv = inv * lv[w].zzz;
                F z1 = zi * lv[w].zz, zi2 = z1.sqr();
                out[(size_t)w * count] = {lv[w].x * zi2, lv[w].y * zi};
            }
        };
        host_parallel_for_wait((int)(4 * count), [&](int task) {
            const size_t tb = (size_t)task / count, i = (size_t)task % count;
            if (tb < 3) levels(p1[tb * count + i], &t1[tb * (size_t)W * count + i]); else levels(p2[i], &t2[i]);
        });
        ext_lap("levels on the host pool");
        WindowTable *g1t[3] = {&T.A, &T.B1, &T.L};
        bool up = true;
        for (int tb = 0; tb < 3 && up; ++tb) {
            WindowTable &t = *g1t[tb]; t.n = count; t.c = c; t.W = W; t.g2 = false;
            up = t.buf.reserve((size_t)W * count * sizeof(G1Affine)) == 0 &&
                 hip_ok(hipMemcpyAsync(t.buf.p, &t1[tb * (size_t)W * count], (size_t)W * count * sizeof(G1Affine), hipMemcpyHostToDevice, s), "H2D", __FILE__, __LINE__);
        }
        T.B2.n = count; T.B2.c = c; T.B2.W = W; T.B2.g2 = true;
        up = up && T.B2.buf.reserve((size_t)W * count * sizeof(G2Affine)) == 0 &&
             hip_ok(hipMemcpyAsync(T.B2.buf.p, t2.data(), (size_t)W * count * sizeof(G2Affine), hipMemcpyHostToDevice, s), "H2D", __FILE__, __LINE__);
        up = hip_ok(hipStreamSynchronize(s), "sync", __FILE__, __LINE__) && up;          // (host vectors go out of scope)
        if (!up) return ZKG_ERROR;
        ext_lap("tables uploaded");
        crs->c_w = c_new; T.count = count; ++T.rebuilds;
        if (g_dbg_timing) fprintf(stderr, "[zkg]     witness tables over %zu of %zu elements (rebuild %u, levels on the host)\n", count, n1, T.rebuilds);
        return ZKG_OK;
    }
    // the G2 table on the B_g2 job's stream (idle as well), beside the three G1 tables: a few thousand points per launch are latency chains
    hipStream_t s2 = msm_job_stream(S.job_w2);
    bool ok = hip_ok(hipEventRecord(S.ev[11], s), "event", __FILE__, __LINE__) && hip_ok(hipStreamWaitEvent(s2, S.ev[11], 0), "wait", __FILE__, __LINE__);   // idx is up
    int rc = !ok || gather_points_g2(crs->B_g2.as<G2Affine>(), d_idx, count, 0, stage2.as<G2Affine>(), s2) || window_table_build_g2(T.B2, stage2.as<G2Affine>(), count, c_new, s2) ||
             gather_points_g1(crs->A_query.as<G1Affine>(), d_idx, count, 0, stage.as<G1Affine>(), s) || window_table_build_g1(T.A, stage.as<G1Affine>(), count, c_new, s) ||
             gather_points_g1(crs->B_g1.as<G1Affine>(), d_idx, count, 0, stage.as<G1Affine>(), s) || window_table_build_g1(T.B1, stage.as<G1Affine>(), count, c_new, s) ||
             gather_points_g1(crs->L_query.as<G1Affine>(), d_idx, count, (uint32_t)(crs->l + 1), stage.as<G1Affine>(), s) || window_table_build_g1(T.L, stage.as<G1Affine>(), count, c_new, s);
    const bool synced1 = hip_ok(hipStreamSynchronize(s), "sync", __FILE__, __LINE__), synced2 = hip_ok(hipStreamSynchronize(s2), "sync", __FILE__, __LINE__);
    const bool synced = synced1 && synced2;                                     // (both streams waited for: host vectors and staging go out of scope)
    stage.release(); stage2.release();
    if (rc || !synced) return ZKG_ERROR;
    crs->c_w = c_new; T.count = count; ++T.rebuilds;
    if (g_dbg_timing) fprintf(stderr, "[zkg]     witness tables over %zu of %zu elements (rebuild %u)\n", count, n1, T.rebuilds);
    return ZKG_OK;
}
// ---- H over several devices: launch (after the transforms, in the slot's stream order) and finish
static int h_shards_launch(zkg_crs *crs, ProverSlot &S) {
    int cur = 0; (void)hipGetDevice(&cur);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{cur};
    if (S.h_epoch != crs->h_epoch) {                                       // this slot's first proof under this sharding: a job and a buffer per shard
        slot_release_h_runs(S);
        S.h_runs.resize(crs->h_shards.size());
        for (size_t i = 0; i < crs->h_shards.size(); ++i) {
            ProverSlot::HShardRun &run = S.h_runs[i];
            run.device = crs->h_shards[i].device;
            ZK_HIP(hipSetDevice(run.device));
            run.job = msm_job_create(nullptr, true);
            if (!run.job || run.scalars.reserve(crs->h_shards[i].n * 32 + 16)) { set_error("prover: H shard workspace"); return ZKG_ERROR; }
            msm_job_set_window(run.job, crs->h_shards[i].table.c); msm_job_set_row_merge(run.job, crs->h_shards[i].n >= H_ROW_MERGE_MIN ? 2u : 1u);
        }
        S.h_epoch = crs->h_epoch;
    }
    // ev[2]: coefficients_for_H complete on the slot's stream (recorded by compute_h_transforms)
    const Fr *coeff = S.aABC.as<Fr>();
    for (size_t i = 0; i < crs->h_shards.size(); ++i) {
        const zkg_crs::HShard &sh = crs->h_shards[i]; ProverSlot::HShardRun &run = S.h_runs[i];
        ZK_HIP(hipSetDevice(sh.device));
        hipStream_t st = msm_job_stream(run.job);
        ZK_HIP(hipStreamWaitEvent(st, S.ev[2], 0));
        ZK_HIP(hipMemcpyAsync(run.scalars.p, coeff + sh.first, sh.n * 32, hipMemcpyDefault, st));       // peer copy over xGMI (or a device copy on one GPU)
        const MsmBases h = table_set(sh.table, false, 0);
        if (msm_job_launch(run.job, &h, 1, run.scalars.as<uint32_t>(), sh.n, true)) return ZKG_ERROR;
    }
    return ZKG_OK;
}
static int h_shards_finish(zkg_crs *crs, ProverSlot &S, G1 &out) {
    int cur = 0; (void)hipGetDevice(&cur);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{cur};
    out = G1::inf();
    for (size_t i = 0; i < crs->h_shards.size(); ++i) {
        ZK_HIP(hipSetDevice(crs->h_shards[i].device));
        G1 part;
        if (msm_job_finish(S.h_runs[i].job, &part, nullptr)) return ZKG_ERROR;
        out.add(part);
    }
    return ZKG_OK;
}

// event slots: 0 witness resident + split done, 1 mat-vec done, 2 H coefficients done, 3 satisfiability flag landed,
//              4-5 G1 witness job, 6-7 G2 witness job, 8-9 H job
static int prove_enqueue(zkg_crs *crs, ProverSlot &S, const WitnessSrc &witness, const uint64_t r_[4], const uint64_t s_[4], bool check) {
    S.t0 = std::chrono::steady_clock::now();
    S.check = check; memcpy(S.r.v, r_, 32); memcpy(S.s.v, s_, 32);
    S.flag_host[0] = 0; S.flag_host[1] = 0; S.flag_host[2] = 0; S.flag_host[3] = 0;
    if (compute_h_matvec(crs, S, witness, check)) return ZKG_ERROR;
    lap(S, "upload + mat-vec enqueued");
    const size_t n = crs->n, l = crs->l, m = crs->m;
    // The witness queries (libff multi_exp_with_mixed_addition): zeros skipped, ones summed flat, the rest — 0.1 % of a credential's
    // witness — through the bucket method as a gathered subset.  The count of "the rest" sizes the launches, so a helper thread waits
    // for the split (event 0), reads it and queues the four witness streams' work while this thread queues the critical path.
    auto witness_fn = [&]() -> int {
        ZK_HIP(hipEventSynchronize(S.ev[0]));
        lap(S, "  (helper) split landed");
        const size_t listed = S.flag_host[1];
        if (S.flag_host[3]) { set_error("prover: a listed witness index is out of range, not tagged 2 or listed twice"); return ZKG_ERROR; }
        if (listed > n + 1) { set_error("prover: witness split out of range"); return ZKG_ERROR; }
        const uint8_t *tags = S.wtags.as<uint8_t>(); const uint32_t *gather = S.wlisted.as<uint32_t>(), *z = S.z.as<uint32_t>();
        if (listed && (S.flag_host[2] || !crs->sub.count)) {
            // the tables are shared by the key's slots: extend them only while no other proof is using them (see zkg_crs)
            std::unique_lock<std::mutex> lk(crs->mu);
            ++crs->waiting_ext; crs->cv.notify_all();
            crs->cv.wait(lk, [&] { return !crs->extending && crs->leases - crs->waiting_ext == 0; });
            --crs->waiting_ext; crs->extending = true;
            lk.unlock();
            int rc_ext = ZKG_ERROR;
            try { rc_ext = subset_extend(crs, S, listed); } catch (...) { set_error("prover: witness table extension failed"); }   // the flag below must be cleared whatever happens
            lk.lock();
            crs->extending = false; crs->cv.notify_all();
            lk.unlock();
            if (rc_ext) return ZKG_ERROR;
        }
        msm_job_set_window(S.job_w1, crs->c_w); msm_job_set_window(S.job_w2, crs->c_w);   // (another slot's proof may have built the tables)
        const uint32_t *pos = crs->sub.pos.as<uint32_t>();
        // bucket method: the subset tables, addressed through the element positions; flat sums over the ones: the queries as uploaded
        const MsmBases g1[3] = {table_set(crs->sub.A, false, 0, pos), table_set(crs->sub.B1, false, 0, pos), table_set(crs->sub.L, false, 0, pos)}, b2 = table_set(crs->sub.B2, true, 0, pos);
        const MsmBases o1[3] = {query_set(crs->A_query, false, 0), query_set(crs->B_g1, false, 0), query_set(crs->L_query, false, (uint32_t)(l + 1))},
                       o2 = query_set(crs->B_g2, true, 0);
        // ZKG_WITNESS_START (tuning aid): which event the witness streams wait for — 0 the split (default), 1 the mat-vec, 2 the transforms
        hipEvent_t go = S.ev[g_witness_start];
        {
            hipStream_t js = msm_job_stream(S.job_w2);                         // G2 first: the longest chains
            ZK_HIP(hipStreamWaitEvent(S.stream_o, go, 0));
            if (ones_sum_launch(S.ones_g2, &o2, 1, tags, n + 1, S.stream_o)) return ZKG_ERROR;
            lap(S, "  (helper) G2 ones-sum enqueued");
            (void)hipEventRecord(S.ev[10], S.stream_o);                        // the G2 sum has landed (the G1 sums follow on the same stream)
            if (g_serial_msm) (void)hipStreamSynchronize(S.stream_o);          // (profiling aid: the flat sum alone on the chip, then the job's kernels)
            ZK_HIP(hipStreamWaitEvent(js, go, 0));
            (void)hipEventRecord(S.ev[6], js);
            if (msm_job_launch(S.job_w2, &b2, 1, z, listed, true, gather)) return ZKG_ERROR;
            lap(S, "  (helper) G2 job enqueued");
            (void)hipEventRecord(S.ev[7], js);
            if (g_serial_msm) { (void)hipStreamSynchronize(S.stream_o); (void)hipStreamSynchronize(js); }
        }
        {
            hipStream_t js = msm_job_stream(S.job_w1);
            if (ones_sum_launch(S.ones_g1, o1, 3, tags, n + 1, S.stream_o)) return ZKG_ERROR;
            lap(S, "  (helper) G1 ones-sums enqueued");
            if (g_serial_msm) (void)hipStreamSynchronize(S.stream_o);
            ZK_HIP(hipStreamWaitEvent(js, go, 0));
            (void)hipEventRecord(S.ev[4], js);
            if (msm_job_launch(S.job_w1, g1, 3, z, listed, true, gather)) return ZKG_ERROR;
            (void)hipEventRecord(S.ev[5], js);
            if (g_serial_msm) { (void)hipStreamSynchronize(S.stream_o); (void)hipStreamSynchronize(js); }
        }
        return ZKG_OK;
    };
    const bool helper = !g_serial_msm && g_witness_start == 0;                  // a later start event must have been recorded before it is waited for
    if (helper) S.helper.submit(witness_fn);
    int rc = compute_h_transforms(crs, S);
    lap(S, "transforms enqueued");
    // H: uniformly random scalars, follows the transforms in stream order
    if (rc == ZKG_OK) {
        hipStream_t js = msm_job_stream(S.job_h);                              // == S.stream
        (void)hipEventRecord(S.ev[8], js);
        if (!crs->h_shards.empty()) rc = h_shards_launch(crs, S);
        else {
        const MsmBases h = table_set(crs->H_query, false, 0);
        rc = msm_job_launch(S.job_h, &h, 1, S.aABC.as<uint32_t>(), m - 1, true);
        }
        (void)hipEventRecord(S.ev[9], js);
        if (g_serial_msm) (void)hipStreamSynchronize(js);
    }
    const int rc_w = helper ? S.helper.wait() : witness_fn();          // (profiling aid: every job alone on the chip, one after the other)
    lap(S, "msm jobs enqueued");
    return rc == ZKG_OK ? rc_w : rc;
}
static void slot_drain(const zkg_crs *, ProverSlot &S) {                      // after an error: nothing of this slot may still be running
    (void)hipStreamSynchronize(S.stream);
    for (MsmJob *j : {S.job_w1, S.job_w2, S.job_h}) if (j) (void)hipStreamSynchronize(msm_job_stream(j));
    if (S.stream_o) (void)hipStreamSynchronize(S.stream_o);
    for (auto &r : S.h_runs) if (r.job) (void)hipStreamSynchronize(msm_job_stream(r.job));
}
static int prove_finish(zkg_crs *crs, ProverSlot &S, uint8_t *proof_out, size_t *proof_len) {
    hipStream_t s = S.stream;
    G1 W1[3]; G2 Bt2; G1 Ht;
    // host work that needs only the key and (r, s), from the fixed-base tables: overlaps the GPU.
    //   A = alpha + W_a + r delta,  B = beta + W_b + s delta (in G2, and its copy B_1 in G1),
    //   C = H + L + s A + r B_1 - rs delta = H + L + (s alpha + r beta_1 + rs delta) + s W_a + r W_b
    // (W_a, W_b, L, H: the four multi-exponentiations).  The same group elements as libsnark's expression order gives; bytes identical.
    uint32_t rc[8], sc[8], rsc[8];
    canonical_limbs(S.r, rc); canonical_limbs(S.s, sc); canonical_limbs(S.r * S.s, rsc);
    G1 gA = G1::from_affine(crs->alpha_g1); gA.add(crs->delta1_comb.mul(rc));                          // alpha + r delta
    G1 c_fixed = crs->alpha1_comb.mul(sc); c_fixed.add(crs->beta1_comb.mul(rc)); c_fixed.add(crs->delta1_comb.mul(rsc));   // s alpha + r beta_1 + rs delta
    G2 gB2 = G2::from_affine(crs->beta_g2); gB2.add(crs->delta2_comb.mul(sc));                         // beta + s delta
    lap(S, "key-only host products");
    if (S.check) {
        ZK_HIP(hipEventSynchronize(S.ev[3]));
        if (S.flag_host[0]) {                                                // drain the speculative MSMs, then refuse like snark.cpp:121-124
            slot_drain(crs, S);
            set_error("constraint system not satisfied; not creating proof"); return ZKG_UNSATISFIED;
        }
    }
    // ---- finish + assembly (host), ordered so that nothing the GPU has already delivered waits for what it is still computing
    if (msm_job_finish(S.job_w1, W1, nullptr) || !hip_ok(hipStreamSynchronize(S.stream_o), "sync", __FILE__, __LINE__)) { slot_drain(crs, S); return ZKG_ERROR; }
    lap(S, "A, B_1, L landed");
    for (int i = 0; i < 3; ++i) W1[i].add(S.ones_g1.g1(i));                   // bucket method over the non-bit elements + flat sum over the ones
    // the two variable-base products s W_a and r W_b: the second one on a helper thread
    G1 rWb;
    S.helper.submit([&]() -> int { rWb = W1[1].mul(rc, 8); return ZKG_OK; });
    G1 gC = W1[0].mul(sc, 8);
    gA.add(W1[0]);
    size_t off = 0;
    off += ser_g1(proof_out + off, gA);
    (void)S.helper.wait();
    gC.add(rWb); gC.add(c_fixed); gC.add(W1[2]);
    lap(S, "A serialised, s*Wa + r*Wb + L");
    if (msm_job_finish(S.job_w2, nullptr, &Bt2) || !hip_ok(hipEventSynchronize(S.ev[10]), "sync", __FILE__, __LINE__)) { slot_drain(crs, S); return ZKG_ERROR; }
    lap(S, "B_2 landed");
    Bt2.add(S.ones_g2.g2pt(0));
    gB2.add(Bt2);
    off += ser_g2(proof_out + off, gB2);
    lap(S, "B serialised");
    if (crs->h_shards.empty() ? msm_job_finish(S.job_h, &Ht, nullptr) : h_shards_finish(crs, S, Ht)) { slot_drain(crs, S); return ZKG_ERROR; }
    lap(S, "H landed");
    gC.add(Ht);                                                             // C = H_t + L_t + s A + r B_1 - rs delta
    off += ser_g1(proof_out + off, gC);
    *proof_len = off;
    lap(S, "assembled+serialised");
    ZK_HIP(hipStreamSynchronize(s));
    {
        float t;
        auto el = [&](int a, int b) { return hipEventElapsedTime(&t, S.ev[a], S.ev[b]) == hipSuccess ? t : -1.f; };
        S.stage_ms[0] = el(0, 1); S.stage_ms[1] = el(1, 2); S.stage_ms[2] = el(4, 5); S.stage_ms[3] = 0.f; S.stage_ms[4] = el(6, 7);
        S.stage_ms[5] = el(8, 9); S.stage_ms[6] = 0.f;
        S.stage_ms[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S.t0).count();
        for (int i = 0; i < 8; ++i) crs->stage_ms[i] = S.stage_ms[i];
    }
    return ZKG_OK;
}

// A dense witness whose variables are mostly bits (a credential circuit's: 97 %) is cheaper to SCAN on the host pool than to upload: 32 bytes per
// variable over PCIe (0.58 ms at 2^20 variables) against a read of the same bytes from host memory in 32 chunks (~0.2 ms) and an upload of one tag byte per
// variable plus the listed values — the sparse form zkg_groth16_prove_sparse takes, built here for callers of the dense entry point.  Gives up (false: the
// dense upload) as soon as more than 1/16 of the variables are not bits, and below 2^19 variables, where waking the pool costs what the upload did
// (tools/r4_dense_scan_ab.sh, dense upload -> scan: 37 payloads / 2^20 domain 3.78 -> 3.53 ms, 8 payloads 1.29 -> 1.32, 2 payloads 0.78 -> 0.83).
// ZKG_DENSE_UPLOAD=1 switches it off (A/B).
static bool witness_to_sparse(const uint64_t *w, size_t n, std::vector<uint8_t> &tags, std::vector<uint32_t> &idx, std::vector<uint64_t> &vals) {
    static const bool off = getenv("ZKG_DENSE_UPLOAD") != nullptr;
    if (off || n < ((size_t)1 << 19)) return false;
    const Fr one_fr = Fr::one();
    uint64_t one[4]; memcpy(one, one_fr.v, 32);
    const size_t cap = n / 16;
    tags.resize(n);
    const int chunks = (int)std::min<size_t>(32, n / 32768 + 1);
    std::vector<std::vector<uint32_t>> found((size_t)chunks);
    std::atomic<bool> over{false};
    uint8_t *t = tags.data();
    zk::host_parallel_for(chunks, [&](int ch) {
        const size_t lo = n * (size_t)ch / (size_t)chunks, hi = n * (size_t)(ch + 1) / (size_t)chunks;
        std::vector<uint32_t> &f = found[(size_t)ch];
        for (size_t v = lo; v < hi; ++v) {
            const uint64_t *e = w + 4 * v;
            if ((e[0] | e[1] | e[2] | e[3]) == 0) t[v] = 0;
            else if (e[0] == one[0] && e[1] == one[1] && e[2] == one[2] && e[3] == one[3]) t[v] = 1;
            else { t[v] = 2; f.push_back((uint32_t)v); if (f.size() > cap) { over.store(true); return; } }
        }
    });
    if (over.load()) return false;
    size_t total = 0;
    for (auto &f : found) total += f.size();
    if (total > cap) return false;
    idx.clear(); idx.reserve(total);
    for (auto &f : found) idx.insert(idx.end(), f.begin(), f.end());
    vals.resize(4 * total);
    for (size_t i = 0; i < total; ++i) memcpy(&vals[4 * i], w + 4 * (size_t)idx[i], 32);
    return true;
}
static int groth16_prove_impl(const zkg_crs *crs_, const uint64_t *witness, const uint64_t r_[4], const uint64_t s_[4], int check_satisfied,
                              uint8_t *proof_out, size_t *proof_len) {
    zkg_crs *crs = const_cast<zkg_crs *>(crs_);
    if (!crs || !r_ || !s_ || !proof_out || !proof_len || (crs->n && !witness)) { set_error("zkg_groth16_prove: bad argument"); return ZKG_ERROR; }
    SlotLease lease(crs);
    if (!lease.ok()) return ZKG_ERROR;
    ProverSlot &S = lease.S();
    WitnessSrc W;
    if (witness_to_sparse(witness, crs->n, S.scan_tags, S.scan_idx, S.scan_vals)) { W.tags = S.scan_tags.data(); W.idx = S.scan_idx.data(); W.vals = S.scan_vals.data(); W.count = S.scan_idx.size(); }
    else W.dense = witness;
    if (prove_enqueue(crs, S, W, r_, s_, check_satisfied != 0)) { slot_drain(crs, S); return ZKG_ERROR; }
    return prove_finish(crs, S, proof_out, proof_len);
}

// the same proof from a sparse description of the witness: tags[n] (0 = zero, 1 = one, 2 = listed) and `count` listed variables as
// (index among the n variables, value as 4 Montgomery limbs).  What a witness generator that knows its bits hands over: the upload
// shrinks ~30x (see k_expand_tags).  Identical proof bytes to zkg_groth16_prove on the expanded vector.
static int groth16_prove_sparse_impl(const zkg_crs *crs_, const uint8_t *tags, const uint32_t *full_index, const uint64_t *full_values, size_t count,
                                     const uint64_t r_[4], const uint64_t s_[4], int check_satisfied, uint8_t *proof_out, size_t *proof_len) {
    zkg_crs *crs = const_cast<zkg_crs *>(crs_);
    if (!crs || !r_ || !s_ || !proof_out || !proof_len || (crs->n && !tags) || (count && (!full_index || !full_values)) || count > crs->n) {
        set_error("zkg_groth16_prove_sparse: bad argument"); return ZKG_ERROR;
    }
    SlotLease lease(crs);
    if (!lease.ok()) return ZKG_ERROR;
    ProverSlot &S = lease.S();
    WitnessSrc W; W.tags = tags; W.idx = full_index; W.vals = full_values; W.count = count;
    if (prove_enqueue(crs, S, W, r_, s_, check_satisfied != 0)) { slot_drain(crs, S); return ZKG_ERROR; }
    return prove_finish(crs, S, proof_out, proof_len);
}

// helper threads and host containers are used below these two: nothing may propagate through the C boundary
int zkg_groth16_prove(const zkg_crs *crs, const uint64_t *witness, const uint64_t r[4], const uint64_t s[4], int check_satisfied,
                      uint8_t *proof_out, size_t *proof_len) {
    try { return groth16_prove_impl(crs, witness, r, s, check_satisfied, proof_out, proof_len); }
    catch (const std::exception &e) { set_error(std::string("zkg_groth16_prove: ") + e.what()); return ZKG_ERROR; }
    catch (...) { set_error("zkg_groth16_prove: unexpected exception"); return ZKG_ERROR; }
}
int zkg_groth16_prove_sparse(const zkg_crs *crs, const uint8_t *tags, const uint32_t *full_index, const uint64_t *full_values, size_t count,
                             const uint64_t r[4], const uint64_t s[4], int check_satisfied, uint8_t *proof_out, size_t *proof_len) {
    try { return groth16_prove_sparse_impl(crs, tags, full_index, full_values, count, r, s, check_satisfied, proof_out, proof_len); }
    catch (const std::exception &e) { set_error(std::string("zkg_groth16_prove_sparse: ") + e.what()); return ZKG_ERROR; }
    catch (...) { set_error("zkg_groth16_prove_sparse: unexpected exception"); return ZKG_ERROR; }
}

// Shards the H query of a resident key over `ndev` devices (a device may be listed more than once: how a one-GPU box rehearses the
// path).  No proof of this key may be in flight.  The key's own device keeps everything else (witness queries, transforms, assembly).
static int crs_shard_h_impl(zkg_crs *crs, const int *devices, int ndev) {
    if (!crs || !devices || ndev < 1 || ndev > 64) { set_error("zkg_crs_shard_h: bad argument"); return ZKG_ERROR; }
    int count = 0; (void)hipGetDeviceCount(&count);
    for (int i = 0; i < ndev; ++i) if (devices[i] < 0 || devices[i] >= count) { set_error("zkg_crs_shard_h: bad device index"); return ZKG_ERROR; }
    std::unique_lock<std::mutex> lk(crs->mu);
    if (crs->leases) { set_error("zkg_crs_shard_h: a proof of this key is in flight"); return ZKG_ERROR; }
    int cur = 0; (void)hipGetDevice(&cur);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{cur};
    crs->home_device = cur;
    const size_t n = crs->H_query.n; const int c = crs->H_query.c;
    const G1Affine *level0 = crs->H_query.buf.as<G1Affine>();                 // level 0 of the table is the query as uploaded
    std::vector<zkg_crs::HShard> shards; shards.reserve((size_t)ndev);
    bool ok = true;
    for (int i = 0; i < ndev && ok; ++i) {
        const size_t first = n * (size_t)i / (size_t)ndev, count = n * (size_t)(i + 1) / (size_t)ndev - first;
        if (!count) continue;                                                  // more shards than points
        shards.emplace_back();
        zkg_crs::HShard &sh = shards.back();
        sh.device = devices[i]; sh.first = first; sh.n = count;
        ok = hipSetDevice(sh.device) == hipSuccess;
        if (ok && sh.device != cur) { int can = 0; if (hipDeviceCanAccessPeer(&can, sh.device, cur) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(cur, 0); (void)hipGetLastError(); }
        ScopedDevBuf slice;
        ok = ok && slice.reserve(sh.n * sizeof(G1Affine) + 16) == 0 &&
             hip_ok(hipMemcpy(slice.p, level0 + sh.first, sh.n * sizeof(G1Affine), hipMemcpyDefault), "H shard copy", __FILE__, __LINE__) &&
             window_table_build_g1(sh.table, slice.as<G1Affine>(), sh.n, c, nullptr) == 0 && window_table_records29(sh.table, nullptr) == 0 &&
             hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__);
    }
    if (!ok) { for (auto &sh : shards) { (void)hipSetDevice(sh.device); sh.table.release(); } return ZKG_ERROR; }
    crs->h_shards.swap(shards); ++crs->h_epoch;                               // slots rebuild their per-shard jobs at their next proof
    for (auto &sh : shards) { (void)hipSetDevice(sh.device); sh.table.release(); }     // a previous sharding's tables
    return ZKG_OK;
}
int zkg_crs_shard_h(zkg_crs *crs, const int *devices, int ndev) {
    try { return crs_shard_h_impl(crs, devices, ndev); }
    catch (const std::exception &e) { set_error(std::string("zkg_crs_shard_h: ") + e.what()); return ZKG_ERROR; }
    catch (...) { set_error("zkg_crs_shard_h: unexpected exception"); return ZKG_ERROR; }
}

int zkg_prover_peak_in_flight(int reset) {
    const int v = g_peak_in_flight.load();
    if (reset) g_peak_in_flight.store(0);
    return v;
}
int zkg_prove_stage_ms(const zkg_crs *crs, float ms[8]) {
    if (!crs || !ms) return ZKG_ERROR;
    for (int i = 0; i < 8; ++i) ms[i] = crs->stage_ms[i];
    return ZKG_OK;
}

}  // extern "C"

// the blob path (codec.hip): queries already on the device, constraint system possibly still being parsed
zkg_crs *crs_upload_device_queries(const zkg_pk *pk, const std::function<bool()> &constraint_system_ready) {
    return zkg_crs_upload_impl(pk, true, constraint_system_ready);
}
