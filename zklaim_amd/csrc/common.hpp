// common.hpp — host-side plumbing shared by the HIP translation units of libzkg.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <functional>
#include <map>
#include <mutex>
#include "curve.hip.hpp"

namespace zk {

void set_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what, const char *file, int line);
#define ZK_HIP(expr) do { if (!::zk::hip_ok((expr), #expr, __FILE__, __LINE__)) return ZKG_ERROR; } while (0)
#define ZK_HIP_V(expr) do { if (!::zk::hip_ok((expr), #expr, __FILE__, __LINE__)) return; } while (0)

// grow-only device buffer (workspaces are allocated outside the timed path and reused)
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// a DevBuf that lives for one scope (staging of a key load, of a table build): released on every way out, an exception's included.
// DevBuf itself has no destructor on purpose — resident buffers are members of objects with explicit lifetimes and are copied into caches.
struct ScopedDevBuf : DevBuf {
    ScopedDevBuf() = default;
    ScopedDevBuf(const ScopedDevBuf &) = delete;
    ScopedDevBuf &operator=(const ScopedDevBuf &) = delete;
    ~ScopedDevBuf() { release(); }
};

// HIP-event timing of the dominant kernel (bench.py's roofline.achieved), on the launch stream
struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs; size_t used = 0; bool enabled = true, pending = false;
    // The two event records around a launch are not free on the stream they sit on (traced: the four accumulation launches of a 2^20-point
    // step lose 38 us to them, 2 % of the step), so only every `stride`-th CALL of an entry point is timed — all of its launches, so that the
    // pieces of a piece-wise call weigh what they weigh — and drain() scales the launch count back up (ZKG_KERNEL_TIMER_STRIDE, default 4; 1: all).
    size_t calls = 0, calls_seen = 0, calls_timed = 0; bool sample = true;
    void new_call();                  // an entry point that launches the dominant kernel starts (under its job's mutex)
    void begin(hipStream_t s); void end(hipStream_t s);
    float drain(int *launches);       // average ms per launch over the timed calls since the last reset (synchronises the events); *launches: the
                                      // launches of ALL calls since then (timed launches x calls / timed calls)
    void reset();
};
extern KernelTimer g_dominant_timer;

// small persistent host thread pool (the per-window tails of the MSMs are independent; see msm.hip host_combine)
// host_parallel_for: a caller that finds the pool busy runs its tasks inline (short loops on a proof's latency path);
// host_parallel_for_wait: queues behind the running loop (bulk work that must be parallel: hashing a 200 MB key blob)
void host_parallel_for(int n, const std::function<void(int)> &fn);
bool host_pool_prewake_enabled();                 // ZKG_POOL_PREWAKE=1
void host_pool_prewake(unsigned spin_us);      // the caller will call host_parallel_for within spin_us: wake the workers now, let them poll for it
void host_parallel_for_wait(int n, const std::function<void(int)> &fn);
void host_parallel_for_spawn(int n, const std::function<void(int)> &fn);   // the pool if it is free, threads of its own if not (heavy loops only)

// ---------------- NTT (ntt.hip) ----------------
struct NttDomain {
    unsigned logn = 0;
    DevBuf tw_fwd, tw_inv;        // omega^i, omega^-i, i < N/2
    DevBuf coset_pre;             // g^i               (cosetFFT pre-multiplication)
    DevBuf icoset_post;           // g^-i / N          (icosetFFT post-multiplication, 1/N folded in)
    DevBuf tw_fwd29, tw_inv29, coset_pre29, icoset_post29;   // the same tables as 40-byte 29-bit records (fr29.hip.hpp), read by k_ntt_pass29
    DevBuf scratch;               // N elements (40-byte records: the 29-bit passes' form between passes)
    Fr n_inv;                     // 1/N
    std::mutex mu; bool first_stream_set = false; hipStream_t first_stream = nullptr;
    std::map<hipStream_t, DevBuf> stream_scratch;
    Fr *scratch_for(hipStream_t s);
    int init(unsigned logn, hipStream_t s);
    void release();
};
NttDomain *ntt_domain(unsigned logn, hipStream_t s);     // cached per size
// mode: inverse / coset as in zkg_ntt.  extra_post (device, N Fr, optional) replaces the default
// post table: the prover fuses iFFT's 1/N with the following cosetFFT's g^i through it.
int ntt_run(NttDomain *d, Fr *d_a, int inverse, int coset, hipStream_t s);
// batch > 1: `batch` vectors, batch_stride elements apart (0 = N: back to back) in d_a and in `scratch` (which must then be given),
// one launch per pass
// pre29 / post29: the 29-bit twins of a caller's own pre / post table (ntt_table29); without them such a call takes the 32-bit pass.
// A caller's scratch holds 40 bytes per element (NTT_SCRATCH_BYTES).
static constexpr size_t NTT_SCRATCH_BYTES = 40;
int ntt_run_ex(NttDomain *d, Fr *d_a, bool inverse, const Fr *pre, const Fr *post, const Fr *post_scalar, hipStream_t s, Fr *scratch = nullptr,
               unsigned batch = 1, size_t batch_stride = 0, const void *pre29 = nullptr, const void *post29 = nullptr);
int ntt_table29(DevBuf &out, const Fr *d_in, size_t n, hipStream_t s);       // a table in libff's form -> its 29-bit records

// libfqfft get_evaluation_domain(min_size) for min_size <= 2^28: basic_radix2_domain (m = 2^k) or step_radix2_domain (m = big + small,
// big = 2^(ceil_log2(m)-1), small = 2^b < big).  zklaim's circuits land on a step domain for 10 of the 20 payload counts.
struct DomainShape { size_t m = 0, big = 0, small = 0; bool step = false; unsigned log_m = 0; /* ceil(log2 m) */ };
bool evaluation_domain_shape(size_t min_size, DomainShape &d);
bool domain_shape_of(size_t m, DomainShape &d);           // m must be a size the rule maps to itself
Fr fr_root_of_unity_pow2(unsigned logn);                  // libff::get_root_of_unity(2^logn)
// evaluate_all_lagrange_polynomials(t) and compute_vanishing_polynomial(t) of the domain (host, one batched inversion)
int domain_lagrange(const DomainShape &d, const Fr &t, std::vector<Fr> &u, Fr &Zt);

// step_radix2_domain on the device: FFT = fold (mod x^big - 1 | x -> omega x, mod x^small - 1) + one radix-2 transform of each size
struct StepDomain {
    DomainShape shape;
    NttDomain *dbig = nullptr, *dsmall = nullptr;
    DevBuf w;                     // omega^i, i < big          (omega = root of unity of order 2 big)
    DevBuf winv_half;             // omega^-i / 2, i < small
    DevBuf g_pow, ginv_pow;       // g^i, g^-i, i < m          (coset variants)
    DevBuf zinv;                  // 1 / Z(g x_i) for i < big: period big/small entries (divide_by_Z_on_coset)
    Fr zinv_small;                // 1 / Z(g x_i), i >= big
    Fr big_inv, small_inv, half;
    int init(const DomainShape &sh, hipStream_t s);
    void release();
};
StepDomain *step_domain(size_t m, hipStream_t s);         // cached per size
// a: batch vectors of m elements, batch_stride apart; scratch: same shape
int step_ntt_run(StepDomain *d, Fr *d_a, bool inverse, bool coset, hipStream_t s, Fr *scratch, unsigned batch = 1, size_t batch_stride = 0);
int powers_table(Fr *d_out, size_t n, const Fr &base, const Fr &scale, hipStream_t s);   // out[i] = scale * base^i
void ntt_release_all();
int ntt_configure();

// ---------------- MSM (msm.hip) ----------------
static constexpr int MSM_MAX_SETS = 4;                      // base sets sharing one digit sort (the prover: A, B_g1, L and B_g2 over one witness)
// one base set of a launch.  level_stride > 0: a per-window table (window_table_build_*): level w of the table, at p + w * level_stride
// elements, holds 2^(c w) P_i, so every window shares one bucket set and the host has no doublings left.  index_sub: entry i of the
// scalars stands for element i - index_sub of the set (the L query starts behind the constant and the public inputs); smaller i: no base.
// remap (optional, device): element i of the scalars stands for entry remap[i] of the set — a table that holds only a subset of the key's
// elements (the prover's witness tables cover the non-bit variables only).
// p29 (optional, G1 only): the same points, every level, as the 64-byte packed 29-bit records (Rec64) of fq29.hip.hpp (window_table_records29): a launch of
// this set alone, without gather / remap / index_sub, then accumulates on the 29-bit representation without converting anything per call.
struct MsmBases { const void *p = nullptr; bool g2 = false; size_t level_stride = 0; uint32_t index_sub = 0; const uint32_t *remap = nullptr; const void *p29 = nullptr; };
// CSR matrices A, B, C handed from a circuit to the key generator without a copy
struct OwnedCsr { std::vector<uint32_t> rp[3], col[3]; std::vector<uint64_t> val[3]; };
struct WindowTable { DevBuf buf, rec29; size_t n = 0; int c = 0, W = 0; bool g2 = false; void release() { buf.release(); rec29.release(); n = 0; } };
int window_table_records29(WindowTable &t, hipStream_t s);                // rec29 <- every level of a built G1 table as 29-bit records
int window_table_build_g1(WindowTable &t, const G1Affine *d_bases, size_t n, int c, hipStream_t s);
int table_window_bits(size_t n);                          // window size of a query's table, by the size of the query (prover.hip)
int window_table_build_g2(WindowTable &t, const G2Affine *d_bases, size_t n, int c, hipStream_t s);
struct MsmJob;                                             // one MSM in flight: stream, workspace, pinned landing zone
MsmJob *msm_job_create(hipStream_t s, bool own_stream, bool high_priority = false);
hipStream_t msm_job_stream(MsmJob *j);
void msm_job_set_window_subset(MsmJob *j, uint32_t w0, uint32_t ws);   // the job computes sum over windows w0, w0+ws, ... of 2^(cw) V_w only
void msm_job_set_skewed(MsmJob *j, bool skewed);            // scalars known to be mostly equal (0/1 witness): use the one-pass sort directly
void msm_job_set_row_merge(MsmJob *j, uint32_t f);        // table launches without a gather list: f consecutive windows share a row of buckets (1 = off)
void msm_job_set_window(MsmJob *j, int c);                 // window bits for the next launches (0 = the size-based rule)
void msm_job_set_critical(MsmJob *j, bool critical);   // the job is on its caller's critical path: its accumulate / fold / reduce wavefronts raise their issue priority (crit_wave_priority)
static constexpr unsigned CRIT_PRIORITY_MIN_LOG = 20;     // ... from this domain size on (tools/r4_crit_prio_ab.sh, sparse-witness proofs, off -> on: 37 payloads / 2^20: 3.14 -> 3.07 ms; 20 payloads / 2^19 + 2^18: 2.13 -> 2.16; 8 payloads / 2^18: 1.12 -> 1.23)
bool crit_priority_for(int part, unsigned log_m);        // part 1: matrix-vector + pointwise, 2: transforms, 4: the H job; tuning aids ZKG_CRIT_PRIO_PARTS (mask, default 7), ZKG_CRIT_PRIO_MIN_LOG
bool crit_priority_enabled();                           // ZKG_CRIT_PRIO=0 switches the raised priorities of the critical-path kernels off (A/B)
void msm_job_destroy(MsmJob *j);
// d_gather (optional, n entries): scalar i is d_scalars[d_gather[i]] and stands for element d_gather[i] of every base set
int msm_job_launch(MsmJob *job, const MsmBases *sets, int nsets, const uint32_t *d_scalars, size_t n, bool scalars_mont, const uint32_t *d_gather = nullptr);
int msm_job_finish(MsmJob *job, G1 *out_g1, G2 *out_g2);   // out_g1[k]: k-th G1 set of the launch, out_g2[k]: k-th G2 set
// the multi_exp_with_mixed_addition split of a witness z = [1 | w] (n1 elements, Montgomery): tags (0 zero, 1 one, 2 other), the
// indices of the others and their count; and the flat sum of the bases tagged one (result lands in pinned host memory)
// d_count: two words, [0] the number of listed elements, [1] set to 1 when a listed element has no entry in d_subset_pos (optional: position of
// every element in the subset the witness tables were built for, SUBSET_NONE = absent)
static constexpr uint32_t SUBSET_NONE = 0xffffffffu;
int witness_classify(const Fr *d_z, size_t n1, uint8_t *d_tags, uint32_t *d_listed, uint32_t *d_count, hipStream_t s, const uint32_t *d_subset_pos = nullptr);
// out[j] = idx[j] >= index_sub ? src[idx[j] - index_sub] : infinity   (level 0 of a subset table)
int gather_points_g1(const G1Affine *d_src, const uint32_t *d_idx, size_t count, uint32_t index_sub, G1Affine *d_out, hipStream_t s);
int gather_points_g2(const G2Affine *d_src, const uint32_t *d_idx, size_t count, uint32_t index_sub, G2Affine *d_out, hipStream_t s);
// out[idx[j]] = src[j] (a sparse vector's values into their dense places; out must be zero-filled = all infinity)
int scatter_points_g1(const G1Affine *d_src, const uint32_t *d_idx, size_t count, G1Affine *d_out, hipStream_t s);
int scatter_points_g2(const G2Affine *d_src, const uint32_t *d_idx, size_t count, G2Affine *d_out, hipStream_t s);
struct OnesSum {                                            // results: nsets points (G1 or G2, XYZZ) back to back in pinned host memory
    DevBuf partials; void *host = nullptr; bool g2 = false; int nsets = 0; void release();
    const G1 &g1(int i) const { return reinterpret_cast<const G1 *>(host)[i]; }
    const G2 &g2pt(int i) const { return reinterpret_cast<const G2 *>(host)[i]; }
};
// level 0 of each set (its plain bases) and its index_sub are read; all sets of one field
int ones_sum_launch(OnesSum &o, const MsmBases *sets, int nsets, const uint8_t *d_tags, size_t n1, hipStream_t s);
// scalars: n x 8 u32 (canonical, or Montgomery when scalars_mont).  Zero scalars are dropped and ones land in
// one heavy bucket, which is what libff's multi_exp_with_mixed_addition prefilter achieves.
int msm_g1(const G1Affine *d_bases, const uint32_t *d_scalars, size_t n, bool scalars_mont, G1 *out, hipStream_t s, bool mostly_bits = false);
int msm_g2(const G2Affine *d_bases, const uint32_t *d_scalars, size_t n, bool scalars_mont, G2 *out, hipStream_t s, bool mostly_bits = false);
int msm_g1_host_scalars(const G1Affine *d_bases, const uint32_t *h_scalars, size_t n, bool scalars_mont, G1 *out, hipStream_t s);   // bases resident, scalars uploaded in pieces under the work
// one digit/sort pass shared by several base sets (A, B_g1, B_g2 queries use the same scalars)
int msm_shared(const G1Affine *const *d_g1_bases, int n_g1, const G2Affine *d_g2_bases, const uint32_t *d_scalars, size_t n,
               bool scalars_mont, G1 *out_g1, G2 *out_g2, hipStream_t s, uint32_t w0 = 0, uint32_t ws = 1,   // w0, ws: window subset (see MsmGeom)
               bool mostly_bits = false);                 // the multi_exp_with_mixed_addition case: scalars mostly 0/1 -> one-pass sort
int fixed_base_g1(const G1Affine &base, const uint32_t *d_scalars, size_t n, G1Affine *d_out, hipStream_t s, bool scalars_mont = false);
int fixed_base_g2(const G2Affine &base, const uint32_t *d_scalars, size_t n, G2Affine *d_out, hipStream_t s, bool scalars_mont = false);
void msm_release_all();
int msm_configure();

// ---------------- key blobs (codec.hip) ----------------
// affine points on the device -> the pk blob's compressed records (34 bytes per G1; 100 per knowledge commitment G2 | G1 of the sparse
// B query, entry idx[j]); d_out: 16-byte aligned device staging for n x 34 (nidx x 100) bytes
int compress_g1_records(const G1Affine *d_in, size_t n, uint8_t *d_out, hipStream_t s);
int compress_kc_records(const G2Affine *d_g2, const G1Affine *d_g1, const uint32_t *d_idx, size_t nidx, uint8_t *d_out, hipStream_t s);

// ---------------- ABI encodings (capi.cpp) ----------------
void store_norm(uint64_t *out, const G1 &p);   // normalised jac, 12 limbs
void store_norm(uint64_t *out, const G2 &p);   // 24 limbs
G1 load_norm_g1(const uint64_t *in);
G2 load_norm_g2(const uint64_t *in);

}  // namespace zk
