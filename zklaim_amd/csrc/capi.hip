// capi.hip — the C ABI of include/zkg.h over the HIP kernels (host code only).
//
// This is the thin shim the reference's C/C++ front-end binds instead of libsnark: the entry
// points sit directly below r1cs_gg_ppzksnark_prover (/root/reference/zklaim/snark.cpp:126).
// There is no CPU fallback: every entry point fails with ZKG_ERROR when no HIP device is usable.
#include "common.hpp"
#include "fq29.hip.hpp"
#include "../../include/zkg.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>                  // types and prototypes only: the library is opened at run time (rccl_api), never linked

namespace zk {

static thread_local std::string t_error;
static std::string g_error;
static std::mutex g_err_mu;
void set_error(const std::string &msg) { t_error = msg; std::lock_guard<std::mutex> lk(g_err_mu); g_error = msg; }
bool hip_ok(hipError_t e, const char *what, const char *file, int line) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e) + " at " + file + ":" + std::to_string(line));
    return false;
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    if (!hip_ok(hipMalloc(&p, want), "hipMalloc", __FILE__, __LINE__)) { p = nullptr; return 1; }
    cap = want;
    return 0;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }

KernelTimer g_dominant_timer;
bool crit_priority_for(int part, unsigned log_m) {
    static const int parts = getenv("ZKG_CRIT_PRIO_PARTS") ? atoi(getenv("ZKG_CRIT_PRIO_PARTS")) : 7;
    static const unsigned min_log = getenv("ZKG_CRIT_PRIO_MIN_LOG") ? (unsigned)atoi(getenv("ZKG_CRIT_PRIO_MIN_LOG")) : CRIT_PRIORITY_MIN_LOG;
    return crit_priority_enabled() && (parts & part) && log_m >= min_log;
}
static std::mutex g_timer_mu;
static constexpr size_t TIMER_MAX_PAIRS = 4096;              // a long-running host that never drains the timer stops recording here
void KernelTimer::new_call() {
    static const size_t stride = getenv("ZKG_KERNEL_TIMER_STRIDE") ? (size_t)std::max(1, atoi(getenv("ZKG_KERNEL_TIMER_STRIDE"))) : 4;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    sample = (calls++ % stride) == 0;
    ++calls_seen; if (sample) ++calls_timed;
}
void KernelTimer::begin(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_timer_mu);
    pending = false;
    if (!sample) return;
    static const bool off = getenv("ZKG_KERNEL_TIMER") && atoi(getenv("ZKG_KERNEL_TIMER")) == 0;      // A/B switch: what the two event records per launch cost the step
    if (off || !enabled || used >= TIMER_MAX_PAIRS) return;
    // (timing only: no system-scope fence — the write-back of whatever the kernel before left dirty would be timed and would delay the kernel behind)
    static const unsigned tf = hipEventDisableSystemFence;
    if (used == pairs.size()) { hipEvent_t a, b; if (hipEventCreateWithFlags(&a, tf) != hipSuccess || hipEventCreateWithFlags(&b, tf) != hipSuccess) { enabled = false; return; } pairs.push_back({a, b}); }
    (void)hipEventRecord(pairs[used].first, s);
    pending = true;
}
void KernelTimer::end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if (!pending) return;
    (void)hipEventRecord(pairs[used].second, s); ++used; pending = false;
}
float KernelTimer::drain(int *launches) {
    std::lock_guard<std::mutex> lk(g_timer_mu);
    float total = 0; int n = 0;
    for (size_t i = 0; i < used; ++i) {
        float ms = 0;
        if (hipEventSynchronize(pairs[i].second) == hipSuccess && hipEventElapsedTime(&ms, pairs[i].first, pairs[i].second) == hipSuccess) { total += ms; ++n; }
    }
    if (launches) *launches = calls_timed ? (int)((size_t)n * calls_seen / calls_timed) : n;
    return n ? total / n : 0.f;
}
void KernelTimer::reset() { std::lock_guard<std::mutex> lk(g_timer_mu); used = 0; pending = false; calls = 0; calls_seen = 0; calls_timed = 0; sample = true; }

// ---- host thread pool -------------------------------------------------------------------------------------
namespace {
struct Run {                                   // one parallel_for: workers that wake late see next >= n and touch nothing else
    std::function<void(int)> fn; int n = 0; std::atomic<int> next{0}, done{0};
};
struct Pool {
    std::vector<std::thread> workers;
    std::mutex mu; std::condition_variable cv_work, cv_done;
    std::shared_ptr<Run> cur; uint64_t epoch = 0; bool stop = false;
    std::mutex run_mu;                         // one parallel_for at a time
    // prewake: a caller that is about to wait for the GPU and will hand out work right after (msm_job_finish) wakes the workers ahead of it; they
    // poll the epoch for at most `spin_us` and go back to sleep.  Waking fifteen sleeping threads through one condition variable costs the
    // work they were woken for 20 - 35 us (tools/pool_wake_bench.hip), a third of the host tail of a multi-exponentiation.
    std::atomic<uint64_t> epoch_a{0}, hint{0}; std::atomic<int64_t> spin_deadline_ns{0};
    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void prewake(unsigned spin_us) {
        if (workers.empty()) return;
        spin_deadline_ns.store(now_ns() + (int64_t)spin_us * 1000);
        { std::lock_guard<std::mutex> lk(mu); hint.fetch_add(1); }
        cv_work.notify_all();
    }
    Pool() {
        unsigned hw = std::thread::hardware_concurrency();
        int nw = (int)(hw > 16 ? 15 : (hw > 1 ? hw - 1 : 0));
        for (int i = 0; i < nw; ++i) workers.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
    }
    void drain(Run &r) {
        for (int i; (i = r.next.fetch_add(1)) < r.n;) {
            r.fn(i);
            if (r.done.fetch_add(1) + 1 == r.n) { std::lock_guard<std::mutex> lk(mu); cv_done.notify_all(); }
        }
    }
    void loop() {
        uint64_t seen = 0, seen_hint = 0;
        for (;;) {
            std::shared_ptr<Run> r;
            bool spin = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || epoch != seen || hint.load() != seen_hint; });
                if (stop) return;
                if (epoch != seen) { seen = epoch; r = cur; }
                else { seen_hint = hint.load(); spin = true; }
            }
            if (spin) {                                                       // woken ahead of the work: poll for it, bounded
                while (epoch_a.load(std::memory_order_acquire) == seen && now_ns() < spin_deadline_ns.load()) __builtin_ia32_pause();
                std::lock_guard<std::mutex> lk(mu);
                seen_hint = hint.load();
                if (stop) return;
                if (epoch != seen) { seen = epoch; r = cur; }
            }
            if (r) drain(*r);
        }
    }
    bool try_run(int count, const std::function<void(int)> &f) {              // the pool's turn if it is free; false (nothing done) if it is busy
        if (workers.empty()) return false;
        std::unique_lock<std::mutex> rl(run_mu, std::defer_lock);
        if (!rl.try_lock()) return false;
        auto r = std::make_shared<Run>(); r->fn = f; r->n = count;
        { std::lock_guard<std::mutex> lk(mu); cur = r; ++epoch; epoch_a.store(epoch, std::memory_order_release); }
        cv_work.notify_all();
        drain(*r);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return r->done.load() == r->n; });
        return true;
    }
    void run(int count, const std::function<void(int)> &f, bool wait_for_pool) {
        if (count <= 1 || workers.empty()) { for (int i = 0; i < count; ++i) f(i); return; }
        // the pool runs one loop at a time; a caller that finds it busy (the seam's key-digest thread hashing 200 MB, say) does its own
        // few tasks inline rather than queueing behind it — these loops sit on the latency path of a proof
        std::unique_lock<std::mutex> rl(run_mu, std::defer_lock);
        if (wait_for_pool) rl.lock();
        else if (!rl.try_lock()) { for (int i = 0; i < count; ++i) f(i); return; }
        auto r = std::make_shared<Run>(); r->fn = f; r->n = count;
        { std::lock_guard<std::mutex> lk(mu); cur = r; ++epoch; epoch_a.store(epoch, std::memory_order_release); }
        cv_work.notify_all();
        drain(*r);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return r->done.load() == r->n; });
    }
};
}  // namespace
static Pool &host_pool() { static Pool pool; return pool; }
void host_parallel_for(int n, const std::function<void(int)> &fn) { host_pool().run(n, fn, false); }
// Off unless ZKG_POOL_PREWAKE=1.  Measured (tools/r4_prewake_ab.sh, tools/pool_wake_bench.hip): the pool's sixteen 18-us tasks take 39 us instead of 51 when
// the workers were woken 250 us ahead, the host tail of a 2^20-point job 64 - 74 us instead of 71 - 94, a sparse proof at 8 / 37 payloads 1.13 / 3.08 ms instead
// of 1.15 - 1.19 / 3.09 - 3.12 — and the 2^20-point step itself does not move (1.868 against 1.878 ms).  Fifteen threads polling for a quarter of a
// millisecond per multi-exponentiation is not worth that by default on a host shared by eight ranks.
bool host_pool_prewake_enabled() { static const bool on = getenv("ZKG_POOL_PREWAKE") && atoi(getenv("ZKG_POOL_PREWAKE")) != 0; return on; }
void host_pool_prewake(unsigned spin_us) { if (host_pool_prewake_enabled()) host_pool().prewake(spin_us); }
void host_parallel_for_wait(int n, const std::function<void(int)> &fn) { host_pool().run(n, fn, true); }
// heavy loops (tens of milliseconds of work: a credential circuit's synthesis with its constraints) that find the pool busy — the key loader hashing and
// parsing a blob, in the same first libsnark_prove — neither run inline (28 ms on one thread instead of 5) nor queue behind it: they get threads of
// their own for the call.  Creating them costs ~0.3 ms; not for the short loops of a proof's latency path.
void host_parallel_for_spawn(int n, const std::function<void(int)> &fn) {
    if (n <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    if (host_pool().try_run(n, fn)) return;
    const int nt = std::min(n, 15);
    std::atomic<int> next{0};
    std::vector<std::thread> th; th.reserve((size_t)nt);
    auto work = [&] { for (int i; (i = next.fetch_add(1)) < n;) fn(i); };
    for (int t = 0; t < nt - 1; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

// ---- ABI point encodings -----------------------------------------------------------------------
static void put(uint64_t *out, const Fq &a) { memcpy(out, a.v, 32); }
static Fq get(const uint64_t *in) { Fq r; memcpy(r.v, in, 32); return r; }
void store_norm(uint64_t *out, const G1 &p) {
    G1Affine a = p.to_affine();
    if (p.is_inf()) { put(out, Fq::zero()); put(out + 4, Fq::one()); put(out + 8, Fq::zero()); }
    else { put(out, a.x); put(out + 4, a.y); put(out + 8, Fq::one()); }
}
void store_norm(uint64_t *out, const G2 &p) {
    G2Affine a = p.to_affine();
    if (p.is_inf()) { put(out, Fq::zero()); put(out + 4, Fq::zero()); put(out + 8, Fq::one()); put(out + 12, Fq::zero()); put(out + 16, Fq::zero()); put(out + 20, Fq::zero()); }
    else { put(out, a.x.c0); put(out + 4, a.x.c1); put(out + 8, a.y.c0); put(out + 12, a.y.c1); put(out + 16, Fq::one()); put(out + 20, Fq::zero()); }
}
G1 load_norm_g1(const uint64_t *in) {
    Fq z = get(in + 8);
    if (z.is_zero()) return G1::inf();
    return {get(in), get(in + 4), Fq::one(), Fq::one()};           // normalised: Z == 1
}
G2 load_norm_g2(const uint64_t *in) {
    Fq2 z = {get(in + 16), get(in + 20)};
    if (z.is_zero()) return G2::inf();
    return {{get(in), get(in + 4)}, {get(in + 8), get(in + 12)}, Fq2::one(), Fq2::one()};
}

// ---- device field arithmetic, element-wise (the known-answer hook behind zkg_field_op) ---------------------------
template <class F> ZK_D F field_apply(int op, const F &x, const F &y) {
    switch (op) {
    case 0: return x * y;   case 1: return x + y;   case 2: return x - y;   case 3: return x.inverse();
    case 6: return x.neg(); case 7: return x.sqr();
    default: return x;
    }
}
template <class F> __global__ __launch_bounds__(64) void k_field_op(int op, const F *a, const F *b, size_t n, F *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = a[i], y = b ? b[i] : x;
    if constexpr (sizeof(F) == sizeof(Fq)) {
        if (op == 4) { out[i] = x.to_mont().normalized(); return; }
        if (op == 5) { out[i] = x.from_mont(); return; }
    }
    if constexpr (std::is_same<F, Fq>::value) {
        // the 29-bit representation of the accumulation kernel (fq29.hip.hpp), entered and left through its own conversions
        if (op >= 10 && op <= 15) {
            const Fq29 xa = f29::to29(x), yb = f29::to29(y);
            Fq29 r = xa;
            if (op == 10) r = f29::mul(xa, yb);
            else if (op == 15) r = f29::inverse(f29::norm(f29::add(f29::add(xa, xa), xa)));      // 1 / (3 a): an unreduced sum as input
            else if (op == 11) r = f29::norm(f29::add(xa, yb));
            else if (op == 12) r = f29::norm(f29::sub(xa, f29::S2_1, yb));
            else if (op == 14) {                 // a chain as the mixed addition builds them: unnormalised differences into products
                const Fq29 d = f29::sub(xa, f29::S6_1, yb), e = f29::norm(f29::sub(yb, f29::S4_1, xa));
                r = f29::norm(f29::sub(f29::mul(e, d), f29::S4_3, f29::add(f29::mul(e, e), f29::dbl(f29::mul(xa, yb)))));      // (b-a)(a-b) - (b-a)^2 - 2ab
            }
            Fq o = f29::from29(r);
            if (op == 13 && f29::is_zero_mod_p(f29::norm(f29::sub(xa, f29::S2_1, yb)))) o = Fq::zero();                          // zero test: out = 0 iff a == b
            out[i] = o; return;
        }
    }
    out[i] = field_apply(op, x, y).normalized();
}
template <class F> static int field_op_run(int op, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    DevBuf da, db, dout;
    const size_t bytes = n * sizeof(F);
    int rc = ZKG_ERROR;
    if (!da.reserve(bytes) && !dout.reserve(bytes) && (!b || !db.reserve(bytes)) &&
        hip_ok(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__) &&
        (!b || hip_ok(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__))) {
        hipLaunchKernelGGL(k_field_op<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, op, da.as<F>(), b ? db.as<F>() : (const F *)nullptr, n, dout.as<F>());
        if (hipGetLastError() == hipSuccess && hip_ok(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost), "D2H", __FILE__, __LINE__)) rc = ZKG_OK;
    }
    da.release(); db.release(); dout.release();
    return rc;
}

// known-answer hook for the 29-bit group law of the reduction kernels (fq29.hip.hpp xyzz29_add_quad): out[i] = a[i] + b[i], one DPP quad
// per pair, points as XYZZ<Fq> in and out (converted by the same quad_load29 / from29 the kernels use)
__global__ __launch_bounds__(64) void k_add_quad29(const XYZZ<Fq> *a, const XYZZ<Fq> *b, size_t n, int chain, XYZZ<Fq> *out) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; const uint32_t q = threadIdx.x & 3;
    if (i >= n) return;
    const XYZZ<Fq> pa = a[i], pb = b[i];
    XYZZ29q x = pa.is_inf() ? XYZZ29q::inf() : quad_load29(pa.x, pa.y, pa.zz, pa.zzz, q);
    const XYZZ29q y = pb.is_inf() ? XYZZ29q::inf() : quad_load29(pb.x, pb.y, pb.zz, pb.zzz, q);
    if (chain < 0) xyzz29_add_lane(x, y); else
    xyzz29_add_quad(x, y, q);
    for (int k = 0; k < chain; ++k) { XYZZ29q z = x; xyzz29_add_quad(x, y, q); xyzz29_add_quad(x, z, q); }     // sums of sums: the invariants across additions
    if (q == 0) out[i] = x.is_inf() ? XYZZ<Fq>::inf().normalized() : XYZZ<Fq>{f29::from29(x.x), f29::from29(x.y), f29::from29(x.zz), f29::from29(x.zzz)};
}

static std::mutex g_init_mu;
static int g_device = -1;
static void kernels_configure() { (void)ntt_configure(); (void)msm_configure(); }

}  // namespace zk

void seam_keygen_quiesce();          // setup_verify.hip

using namespace zk;

extern "C" {

int zkg_init(int device) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("zkg_init: no HIP device visible (this library has no CPU fallback)"); return ZKG_ERROR; }
    if (device < 0 || device >= count) { set_error("zkg_init: bad device index"); return ZKG_ERROR; }
    ZK_HIP(hipSetDevice(device));
    g_device = device;
    kernels_configure();
    return ZKG_OK;
}
void zkg_shutdown(void) {
    seam_keygen_quiesce();
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_device < 0) return;
    (void)hipDeviceSynchronize();
    ntt_release_all();
    msm_release_all();
    g_device = -1;
}
const char *zkg_last_error(void) { std::lock_guard<std::mutex> lk(g_err_mu); static std::string copy; copy = g_error; return copy.c_str(); }

int zkg_device_info(char *name, size_t name_len, int *compute_units) {
    if (g_device < 0) { set_error("zkg_init not called"); return ZKG_ERROR; }
    hipDeviceProp_t prop;
    ZK_HIP(hipGetDeviceProperties(&prop, g_device));
    if (name && name_len) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return ZKG_OK;
}

#define REQUIRE_INIT() do { if (g_device < 0) { set_error("zkg_init not called"); return ZKG_ERROR; } } while (0)

int zkg_ntt_dev(void *d_a, unsigned logN, int inverse, int coset, void *stream) {
    REQUIRE_INIT();
    if (logN > 28) { set_error("zkg_ntt: logN exceeds the 2-adicity (28) of Fr"); return ZKG_ERROR; }
    hipStream_t s = (hipStream_t)stream;
    NttDomain *d = ntt_domain(logN, s);
    if (!d) return ZKG_ERROR;
    return ntt_run(d, (Fr *)d_a, inverse, coset, s);
}
// The host-pointer transforms run on the null stream with the domain's shared inter-pass scratch: callers on several threads take turns
// (device-pointer callers pass their own stream and get a scratch vector per stream, NttDomain::scratch_for).
static std::mutex g_host_ntt_mu;
int zkg_ntt(uint64_t *a, unsigned logN, int inverse, int coset) {
    REQUIRE_INIT();
    if (!a || logN > 28) { set_error("zkg_ntt: bad argument"); return ZKG_ERROR; }
    std::lock_guard<std::mutex> lk(g_host_ntt_mu);
    size_t bytes = ((size_t)1 << logN) * 32;
    DevBuf buf;
    if (buf.reserve(bytes)) return ZKG_ERROR;
    int rc = ZKG_ERROR;
    if (hip_ok(hipMemcpy(buf.p, a, bytes, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__) &&
        zkg_ntt_dev(buf.p, logN, inverse, coset, nullptr) == ZKG_OK &&
        hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__) &&
        hip_ok(hipMemcpy(a, buf.p, bytes, hipMemcpyDeviceToHost), "D2H", __FILE__, __LINE__)) rc = ZKG_OK;
    buf.release();
    return rc;
}

int zkg_evaluation_domain_size(size_t min_size, size_t *m, int *is_step) {
    DomainShape sh;
    if (!evaluation_domain_shape(min_size, sh)) { set_error("zkg_evaluation_domain_size: no radix-2 or step domain for that size"); return ZKG_ERROR; }
    if (m) *m = sh.m;
    if (is_step) *is_step = sh.step ? 1 : 0;
    return ZKG_OK;
}
int zkg_ntt_domain_dev(void *d_a, size_t m, int inverse, int coset, void *stream) {
    REQUIRE_INIT();
    DomainShape sh;
    if (!domain_shape_of(m, sh)) { set_error("zkg_ntt_domain: m is neither 2^k nor 2^a + 2^b (get_evaluation_domain would not return it)"); return ZKG_ERROR; }
    if (!sh.step) return zkg_ntt_dev(d_a, sh.log_m, inverse, coset, stream);
    hipStream_t s = (hipStream_t)stream;
    StepDomain *d = step_domain(m, s);
    if (!d) return ZKG_ERROR;
    DevBuf scratch;                                           // one-off call: the prover keeps its own
    if (scratch.reserve(m * NTT_SCRATCH_BYTES)) return ZKG_ERROR;
    int rc = step_ntt_run(d, (Fr *)d_a, inverse != 0, coset != 0, s, scratch.as<Fr>());
    if (!hip_ok(hipStreamSynchronize(s), "sync", __FILE__, __LINE__)) rc = ZKG_ERROR;
    scratch.release();
    return rc;
}
int zkg_ntt_domain(uint64_t *a, size_t m, int inverse, int coset) {
    REQUIRE_INIT();
    if (!a || m < 2) { set_error("zkg_ntt_domain: bad argument"); return ZKG_ERROR; }
    std::lock_guard<std::mutex> lk(g_host_ntt_mu);
    size_t bytes = m * 32;
    DevBuf buf;
    if (buf.reserve(bytes)) return ZKG_ERROR;
    int rc = ZKG_ERROR;
    if (hip_ok(hipMemcpy(buf.p, a, bytes, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__) &&
        zkg_ntt_domain_dev(buf.p, m, inverse, coset, nullptr) == ZKG_OK &&
        hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__) &&
        hip_ok(hipMemcpy(a, buf.p, bytes, hipMemcpyDeviceToHost), "D2H", __FILE__, __LINE__)) rc = ZKG_OK;
    buf.release();
    return rc;
}

int zkg_msm_g1_dev(const void *d_bases, const void *d_scalars, size_t n, int scalars_mont, uint64_t out_jac[12], void *stream) {
    REQUIRE_INIT();
    G1 r;
    if (msm_g1((const G1Affine *)d_bases, (const uint32_t *)d_scalars, n, (scalars_mont & ZKG_SCALARS_MONT) != 0, &r, (hipStream_t)stream, (scalars_mont & ZKG_SCALARS_MOSTLY_BITS) != 0)) return ZKG_ERROR;
    store_norm(out_jac, r);
    return ZKG_OK;
}
int zkg_msm_g1_host_scalars(const void *d_bases, const uint64_t *scalars, size_t n, int scalars_mont, uint64_t out_jac[12], void *stream) {
    REQUIRE_INIT();
    if (!out_jac || (n && (!d_bases || !scalars))) { set_error("zkg_msm_g1_host_scalars: bad argument"); return ZKG_ERROR; }
    G1 r;
    if (msm_g1_host_scalars((const G1Affine *)d_bases, (const uint32_t *)scalars, n, (scalars_mont & ZKG_SCALARS_MONT) != 0, &r, (hipStream_t)stream)) return ZKG_ERROR;
    store_norm(out_jac, r);
    return ZKG_OK;
}
int zkg_msm_g1_windows_dev(const void *d_bases, const void *d_scalars, size_t n, int scalars_mont, unsigned first_window, unsigned window_stride,
                           uint64_t out_jac[12], void *stream) {
    REQUIRE_INIT();
    if (!window_stride) { set_error("zkg_msm_g1_windows_dev: window_stride must be positive"); return ZKG_ERROR; }
    G1 r; const G1Affine *b = (const G1Affine *)d_bases;
    if (msm_shared(&b, 1, nullptr, (const uint32_t *)d_scalars, n, (scalars_mont & ZKG_SCALARS_MONT) != 0, &r, nullptr, (hipStream_t)stream, first_window, window_stride,
                   (scalars_mont & ZKG_SCALARS_MOSTLY_BITS) != 0)) return ZKG_ERROR;
    store_norm(out_jac, r);
    return ZKG_OK;
}
// ---- fixed bases kept resident with their per-window tables (what the prover's H query gets): include/zkg.h
struct zkg_msm_bases { WindowTable table; MsmJob *job = nullptr; std::mutex mu; int device = 0; hipEvent_t ev_in = nullptr; };
zkg_msm_bases *zkg_msm_g1_bases_upload(const void *d_bases, size_t n) {
    if (g_device < 0) { set_error("zkg_init not called"); return nullptr; }
    if (!d_bases || n == 0 || n >= ((size_t)1 << 27)) { set_error("zkg_msm_g1_bases_upload: bad argument"); return nullptr; }
    zkg_msm_bases *h = new (std::nothrow) zkg_msm_bases();
    if (!h) return nullptr;
    (void)hipGetDevice(&h->device);
    h->job = msm_job_create(nullptr, true);
    bool ok = h->job != nullptr && hip_ok(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming), "hipEventCreate", __FILE__, __LINE__) && window_table_build_g1(h->table, (const G1Affine *)d_bases, n, table_window_bits(n), nullptr) == 0 && window_table_records29(h->table, nullptr) == 0 &&
              hip_ok(hipDeviceSynchronize(), "sync", __FILE__, __LINE__);
    if (!ok) { zkg_msm_g1_bases_free(h); return nullptr; }
    msm_job_set_window(h->job, h->table.c); msm_job_set_row_merge(h->job, n >= 49152 ? 2u : 1u);
    return h;
}
void zkg_msm_g1_bases_free(zkg_msm_bases *h) {
    if (!h) return;
    int cur = 0; (void)hipGetDevice(&cur);
    if (cur != h->device) (void)hipSetDevice(h->device);               // the handle's memory, stream and event live on the device it was made on
    if (h->job) { (void)hipStreamSynchronize(msm_job_stream(h->job)); msm_job_destroy(h->job); }
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    h->table.release();
    if (cur != h->device) (void)hipSetDevice(cur);
    delete h;
}
int zkg_msm_g1_resident(zkg_msm_bases *h, const void *d_scalars, size_t n, int scalars_mont, uint64_t out_jac[12], void *stream) {
    REQUIRE_INIT();
    if (!h || !d_scalars || !out_jac || n != h->table.n) { set_error("zkg_msm_g1_resident: bad argument (n must be the handle's point count)"); return ZKG_ERROR; }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device) { set_error("zkg_msm_g1_resident: the calling thread's device is not the one the bases were uploaded on"); return ZKG_ERROR; }
    std::lock_guard<std::mutex> lk(h->mu);                              // the handle owns one job: calls take turns
    // the job runs on the handle's own (non-blocking) stream: order it behind whatever the caller queued on `stream` (the work that
    // produces d_scalars).  The call returns after the result has landed, so nothing of it is left running for the caller to order after.
    ZK_HIP(hipEventRecord(h->ev_in, (hipStream_t)stream));
    ZK_HIP(hipStreamWaitEvent(msm_job_stream(h->job), h->ev_in, 0));
    MsmBases b; b.p = h->table.buf.p; b.g2 = false; b.level_stride = h->table.n; b.p29 = h->table.rec29.p;
    G1 r;
    if (msm_job_launch(h->job, &b, 1, (const uint32_t *)d_scalars, n, (scalars_mont & ZKG_SCALARS_MONT) != 0, nullptr) || msm_job_finish(h->job, &r, nullptr)) return ZKG_ERROR;
    store_norm(out_jac, r);
    return ZKG_OK;
}
int zkg_msm_g2_dev(const void *d_bases, const void *d_scalars, size_t n, int scalars_mont, uint64_t out_jac[24], void *stream) {
    REQUIRE_INIT();
    G2 r;
    if (msm_g2((const G2Affine *)d_bases, (const uint32_t *)d_scalars, n, (scalars_mont & ZKG_SCALARS_MONT) != 0, &r, (hipStream_t)stream, (scalars_mont & ZKG_SCALARS_MOSTLY_BITS) != 0)) return ZKG_ERROR;
    store_norm(out_jac, r);
    return ZKG_OK;
}
static int msm_host(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t *out, int g2) {
    REQUIRE_INIT();
    if (n && (!bases || !scalars)) { set_error("zkg_msm: null input"); return ZKG_ERROR; }
    size_t bb = n * (g2 ? 128 : 64), sb = n * 32;
    DevBuf db, ds;
    if (db.reserve(bb + 16) || ds.reserve(sb + 16)) return ZKG_ERROR;
    int rc = ZKG_ERROR;
    if ((!n || (hip_ok(hipMemcpy(db.p, bases, bb, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__) &&
                hip_ok(hipMemcpy(ds.p, scalars, sb, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__))))
        rc = g2 ? zkg_msm_g2_dev(db.p, ds.p, n, 0, out, nullptr) : zkg_msm_g1_dev(db.p, ds.p, n, 0, out, nullptr);
    db.release(); ds.release();
    return rc;
}
int zkg_msm_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_jac[12]) { return msm_host(bases, scalars, n, out_jac, 0); }
int zkg_msm_g2(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_jac[24]) { return msm_host(bases, scalars, n, out_jac, 1); }

int zkg_g1_sum(const uint64_t *points_jac, size_t count, uint64_t out_jac[12]) {
    G1 acc = G1::inf();
    for (size_t i = 0; i < count; ++i) acc.add(load_norm_g1(points_jac + 12 * i));
    store_norm(out_jac, acc);
    return ZKG_OK;
}
// out[i] = a[i] + b[i] (then `chain` rounds of x <- 2x + b) on the GPU through the 29-bit quad addition; normalised Jacobian in and out
int zkg_g1_add_quad29(const uint64_t *a_jac, const uint64_t *b_jac, size_t n, int chain, uint64_t *out_jac) {
    REQUIRE_INIT();
    if (!a_jac || !b_jac || !out_jac || chain < -1 || chain > 64) { set_error("zkg_g1_add_quad29: bad argument"); return ZKG_ERROR; }
    if (!n) return ZKG_OK;
    std::vector<G1> ha(n), hb(n), ho(n);
    for (size_t i = 0; i < n; ++i) { ha[i] = load_norm_g1(a_jac + 12 * i).normalized(); hb[i] = load_norm_g1(b_jac + 12 * i).normalized(); }
    ScopedDevBuf da, db, dout;
    if (da.reserve(n * sizeof(G1)) || db.reserve(n * sizeof(G1)) || dout.reserve(n * sizeof(G1))) return ZKG_ERROR;
    ZK_HIP(hipMemcpy(da.p, ha.data(), n * sizeof(G1), hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(db.p, hb.data(), n * sizeof(G1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_add_quad29, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, nullptr, da.as<G1>(), db.as<G1>(), n, chain, dout.as<G1>());
    if (hipGetLastError() != hipSuccess) { set_error("zkg_g1_add_quad29: launch failed"); return ZKG_ERROR; }
    ZK_HIP(hipMemcpy(ho.data(), dout.p, n * sizeof(G1), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) store_norm(out_jac + 12 * i, ho[i]);
    return ZKG_OK;
}
// the same hook for the pair form (fq29.hip.hpp xyzz29_add_pair): one pair of lanes per point pair, lane 0 holding (X, ZZ), lane 1 (Y, ZZZ)
__global__ __launch_bounds__(64) void k_add_pair29(const XYZZ<Fq> *a, const XYZZ<Fq> *b, size_t n, int chain, XYZZ<Fq> *out) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1; const uint32_t r = threadIdx.x & 1;
    if (i >= n) return;
    auto half = [&](const XYZZ<Fq> &p) {
        Half29 h = Half29::inf();
        if (!p.is_inf()) { h.c0 = f29::to29(r ? p.y : p.x); h.c1 = f29::to29(r ? p.zzz : p.zz); }
        return h;
    };
    Half29 x = half(a[i]); const Half29 y = half(b[i]);
    xyzz29_add_pair(x, y, r);
    for (int k = 0; k < chain; ++k) { const Half29 z = x; xyzz29_add_pair(x, y, r); xyzz29_add_pair(x, z, r); }
    // both halves through a scratch record in LDS, then lane 0 converts and writes
    __shared__ uint32_t rec[32][36];
    uint32_t *w = rec[threadIdx.x >> 1] + 9 * r;
    for (int j = 0; j < 9; ++j) { w[j] = x.c0.v[j]; w[18 + j] = x.c1.v[j]; }
    __syncthreads();
    if (r == 0) {
        Fq29 c[4]; const uint32_t *q = rec[threadIdx.x >> 1]; uint32_t any = 0;
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 9; ++j) c[k].v[j] = q[9 * k + j];
        for (int j = 0; j < 9; ++j) any |= c[2].v[j];
        out[i] = any ? XYZZ<Fq>{f29::from29(c[0]), f29::from29(c[1]), f29::from29(c[2]), f29::from29(c[3])} : XYZZ<Fq>::inf().normalized();
    }
}
int zkg_g1_add_pair29(const uint64_t *a_jac, const uint64_t *b_jac, size_t n, int chain, uint64_t *out_jac) {
    REQUIRE_INIT();
    if (!a_jac || !b_jac || !out_jac || chain < 0 || chain > 64) { set_error("zkg_g1_add_pair29: bad argument"); return ZKG_ERROR; }
    if (!n) return ZKG_OK;
    std::vector<G1> ha(n), hb(n), ho(n);
    for (size_t i = 0; i < n; ++i) { ha[i] = load_norm_g1(a_jac + 12 * i).normalized(); hb[i] = load_norm_g1(b_jac + 12 * i).normalized(); }
    ScopedDevBuf da, db, dout;
    if (da.reserve(n * sizeof(G1)) || db.reserve(n * sizeof(G1)) || dout.reserve(n * sizeof(G1))) return ZKG_ERROR;
    ZK_HIP(hipMemcpy(da.p, ha.data(), n * sizeof(G1), hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(db.p, hb.data(), n * sizeof(G1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_add_pair29, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, nullptr, da.as<G1>(), db.as<G1>(), n, chain, dout.as<G1>());
    if (hipGetLastError() != hipSuccess) { set_error("zkg_g1_add_pair29: launch failed"); return ZKG_ERROR; }
    ZK_HIP(hipMemcpy(ho.data(), dout.p, n * sizeof(G1), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) store_norm(out_jac + 12 * i, ho[i]);
    return ZKG_OK;
}
int zkg_g2_sum(const uint64_t *points_jac, size_t count, uint64_t out_jac[24]) {
    G2 acc = G2::inf();
    for (size_t i = 0; i < count; ++i) acc.add(load_norm_g2(points_jac + 24 * i));
    store_norm(out_jac, acc);
    return ZKG_OK;
}

int zkg_g1_fixed_base_dev(const uint64_t base[8], const void *d_scalars, size_t n, void *d_out_affine, void *stream) {
    REQUIRE_INIT();
    G1Affine b; memcpy(&b, base, 64);
    return fixed_base_g1(b, (const uint32_t *)d_scalars, n, (G1Affine *)d_out_affine, (hipStream_t)stream);
}
int zkg_g2_fixed_base_dev(const uint64_t base[16], const void *d_scalars, size_t n, void *d_out_affine, void *stream) {
    REQUIRE_INIT();
    G2Affine b; memcpy(&b, base, 128);
    return fixed_base_g2(b, (const uint32_t *)d_scalars, n, (G2Affine *)d_out_affine, (hipStream_t)stream);
}

int zkg_field_op(int field, int op, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    REQUIRE_INIT();
    const bool binary = (op >= 0 && op <= 2) || (op >= 10 && op <= 14 && field == 0), known = binary || op == 3 || op == 6 || op == 7 || (op == 15 && field == 0) || ((op == 4 || op == 5) && field != 2);
    if (!known || field < 0 || field > 2 || !a || !out || (binary && !b)) { set_error("zkg_field_op: bad argument"); return ZKG_ERROR; }
    if (!n) return ZKG_OK;
    if (field == 0) return field_op_run<Fq>(op, a, binary ? b : nullptr, n, out);
    if (field == 1) return field_op_run<Fr>(op, a, binary ? b : nullptr, n, out);
    return field_op_run<Fq2>(op, a, binary ? b : nullptr, n, out);
}

// ---- one process, several GPUs: the G1 multi-exponentiation sharded by points.  Every shard is a resident slice of
//      the bases on one device with its own stream and workspace; a call hands each shard its slice of the scalars on a host thread of
//      its own (hipSetDevice is per thread), the shards run the complete single-GPU Pippenger concurrently, and the partial points — 96
//      bytes each — are added on the host.  There is no data-path collective: RCCL has no elliptic-curve reduction, and a sum of
//      ndev points is not worth a kernel.  (The one-process-per-GPU form of the same exchange is zklaim_amd/dist.py over RCCL.)
namespace { struct JoinAll { std::vector<std::thread> &t; ~JoinAll() { for (auto &x : t) if (x.joinable()) x.join(); } }; }   // also on an exception
// RCCL for the exchange of the shards' partial points (ZKG_MULTI_RCCL=1; BASELINE.json north_star: "RCCL ... of partial ... sums over xGMI").
// One process drives every GPU, so the communicators come from ncclCommInitAll and the all-gather of the 96-byte partials is one group call.
// The library is dlopen'ed on first use (a process that already holds PyTorch's copy gets that one): libzkg.so itself links no RCCL, and the
// single-GPU entry points never touch it.
extern "C++" {
namespace {
struct RcclApi {
    void *lib = nullptr; bool tried = false;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};
RcclApi &rccl_api() {
    static RcclApi a; static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!a.tried) {
        a.tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) { a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (a.lib) break; }
        if (a.lib) {
            a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.lib, "ncclCommInitAll"); a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
            a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather"); a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
            a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd"); a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        }
    }
    return a;
}
}  // namespace
}  // extern "C++"
struct zkg_msm_shards {
    struct Shard { int device = 0; size_t first = 0, n = 0; DevBuf bases, scalars; MsmJob *job = nullptr; DevBuf xsend, xrecv; ncclComm_t comm = nullptr; };
    std::vector<Shard> shards; size_t n = 0;
    std::mutex mu;                       // a call owns every shard's scalar buffer, job workspace and stream: callers of one handle take turns
    int rccl_state = 0;                  // 0 not tried, 1 communicators up, -1 unavailable (no library, a device listed twice, init failed): host exchange
    uint64_t *pinned = nullptr;          // ndev x 12 words: the partials on their way to the devices, and the gathered set coming back
};
// the shards' communicators, on first use: one rank per shard, which RCCL only allows on distinct devices
static bool shards_rccl_ready(zkg_msm_shards *h) {
    if (h->rccl_state) return h->rccl_state > 0;
    h->rccl_state = -1;
    const size_t ns = h->shards.size();
    for (size_t i = 0; i < ns; ++i) for (size_t j = i + 1; j < ns; ++j) if (h->shards[i].device == h->shards[j].device) return false;   // (the single-GPU rehearsal lists a device twice)
    RcclApi &R = rccl_api();
    if (!R.ok()) return false;
    std::vector<int> devs(ns); std::vector<ncclComm_t> comms(ns, nullptr);
    for (size_t i = 0; i < ns; ++i) devs[i] = h->shards[i].device;
    int cur = 0; (void)hipGetDevice(&cur);
    bool ok = R.CommInitAll(comms.data(), (int)ns, devs.data()) == ncclSuccess;
    for (size_t i = 0; ok && i < ns; ++i) {
        zkg_msm_shards::Shard &sh = h->shards[i];
        sh.comm = comms[i];
        ok = hipSetDevice(sh.device) == hipSuccess && !sh.xsend.reserve(96) && !sh.xrecv.reserve(ns * 96);
    }
    ok = ok && hipHostMalloc((void **)&h->pinned, 2 * ns * 96, hipHostMallocDefault) == hipSuccess;
    (void)hipSetDevice(cur);
    if (!ok) { for (size_t i = 0; i < ns; ++i) if (comms[i]) { (void)R.CommDestroy(comms[i]); h->shards[i].comm = nullptr; } return false; }
    h->rccl_state = 1;
    return true;
}

int zkg_init_multi(const int *devices, int ndev) {
    if (!devices || ndev < 1) { set_error("zkg_init_multi: bad argument"); return ZKG_ERROR; }
    if (zkg_init(devices[0])) return ZKG_ERROR;                                   // also the calling thread's device for the single-GPU entry points
    std::lock_guard<std::mutex> lk(g_init_mu);
    int count = 0;
    (void)hipGetDeviceCount(&count);
    for (int i = 1; i < ndev; ++i) {
        if (devices[i] < 0 || devices[i] >= count) { set_error("zkg_init_multi: bad device index"); return ZKG_ERROR; }
        ZK_HIP(hipSetDevice(devices[i]));
        kernels_configure();                                                      // kernel attributes are per device
    }
    ZK_HIP(hipSetDevice(devices[0]));
    return ZKG_OK;
}

static zkg_msm_shards *zkg_msm_g1_shards_upload_impl(const uint64_t *bases, size_t n, const int *devices, int ndev) {
    if (g_device < 0) { set_error("zkg_init not called"); return nullptr; }
    if ((n && !bases) || !devices || ndev < 1 || ndev > 64) { set_error("zkg_msm_g1_shards_upload: bad argument"); return nullptr; }
    std::unique_ptr<zkg_msm_shards, void (*)(zkg_msm_shards *)> holder(new zkg_msm_shards(), zkg_msm_g1_shards_free);   // released on every early exit
    zkg_msm_shards *h = holder.get();
    h->n = n; h->shards.resize((size_t)ndev);
    std::vector<int> rc((size_t)ndev, ZKG_OK);
    std::vector<std::thread> th;
    {
    JoinAll join_guard{th};
    for (int i = 0; i < ndev; ++i) {
        zkg_msm_shards::Shard &sh = h->shards[(size_t)i];
        sh.device = devices[i]; sh.first = n * (size_t)i / (size_t)ndev; sh.n = n * (size_t)(i + 1) / (size_t)ndev - sh.first;
        th.emplace_back([&sh, &rc, i, bases] {
            if (hipSetDevice(sh.device) != hipSuccess) { set_error("zkg_msm_g1_shards_upload: hipSetDevice failed"); rc[(size_t)i] = ZKG_ERROR; return; }
            sh.job = msm_job_create(nullptr, true);
            if (!sh.job || sh.bases.reserve(sh.n * 64 + 16) || sh.scalars.reserve(sh.n * 32 + 16) ||
                (sh.n && !hip_ok(hipMemcpy(sh.bases.p, bases + 8 * sh.first, sh.n * 64, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__))) rc[(size_t)i] = ZKG_ERROR;
        });
    }
    }
    for (int r : rc) if (r) return nullptr;
    return holder.release();
}
zkg_msm_shards *zkg_msm_g1_shards_upload(const uint64_t *bases, size_t n, const int *devices, int ndev) {
    try { return zkg_msm_g1_shards_upload_impl(bases, n, devices, ndev); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("zkg_msm_g1_shards_upload: ") + e.what()); return nullptr; }
    catch (...) { zk::set_error("zkg_msm_g1_shards_upload: unexpected exception"); return nullptr; }
}

void zkg_msm_g1_shards_free(zkg_msm_shards *h) {
    if (!h) return;
    int cur = 0; (void)hipGetDevice(&cur);
    for (auto &sh : h->shards) {
        (void)hipSetDevice(sh.device); msm_job_destroy(sh.job); sh.bases.release(); sh.scalars.release(); sh.xsend.release(); sh.xrecv.release();
        if (sh.comm) (void)rccl_api().CommDestroy(sh.comm);
    }
    if (h->pinned) (void)hipHostFree(h->pinned);
    (void)hipSetDevice(cur);
    delete h;
}
size_t zkg_msm_g1_shards_count(const zkg_msm_shards *h, size_t *points) { if (points) *points = h ? h->n : 0; return h ? h->shards.size() : 0; }

static std::atomic<unsigned> g_multi_rccl_calls{0};                                // zkg_msm_g1_multi calls that exchanged over RCCL (zkg_multi_rccl_calls)
unsigned zkg_multi_rccl_calls(void) { return g_multi_rccl_calls.load(); }
static int zkg_msm_g1_multi_impl(const zkg_msm_shards *h_, const uint64_t *scalars, uint64_t out_jac[12], uint64_t *partials_jac) {
    zkg_msm_shards *h = const_cast<zkg_msm_shards *>(h_);
    if (!h || !out_jac || (h->n && !scalars)) { set_error("zkg_msm_g1_multi: bad argument"); return ZKG_ERROR; }
    std::lock_guard<std::mutex> calls_take_turns(h->mu);
    const size_t ns = h->shards.size();
    std::vector<G1> part(ns, G1::inf()); std::vector<int> rc(ns, ZKG_OK);
    auto run = [&](size_t i) {
        zkg_msm_shards::Shard &sh = h->shards[i];
        if (hipSetDevice(sh.device) != hipSuccess) { set_error("zkg_msm_g1_multi: hipSetDevice failed"); rc[i] = ZKG_ERROR; return; }
        hipStream_t st = msm_job_stream(sh.job);
        MsmBases set; set.p = sh.bases.p;
        if ((sh.n && !hip_ok(hipMemcpyAsync(sh.scalars.p, scalars + 4 * sh.first, sh.n * 32, hipMemcpyHostToDevice, st), "H2D", __FILE__, __LINE__)) ||
            msm_job_launch(sh.job, &set, 1, sh.scalars.as<uint32_t>(), sh.n, false) || msm_job_finish(sh.job, &part[i], nullptr)) rc[i] = ZKG_ERROR;
    };
    std::vector<std::thread> th;
    {
        JoinAll join_guard{th};
        for (size_t i = 1; i < ns; ++i) th.emplace_back(run, i);
        int cur = 0; (void)hipGetDevice(&cur);
        run(0);                                                                   // shard 0 on the calling thread
        (void)hipSetDevice(cur);
    }
    for (int r : rc) if (r) return ZKG_ERROR;
    // the exchange.  Default: the partials are already on the host (each shard's finish step lands its chunk sums there), so they are simply
    // added.  ZKG_MULTI_RCCL=1: every device all-gathers the normalised partials over RCCL (xGMI) first and the sum is taken over what device 0
    // received — the collective form of the same 96-byte-per-GPU exchange (one more round trip; the points must agree with the host's copies).
    const bool want_rccl = getenv("ZKG_MULTI_RCCL") != nullptr;                  // (read per call: the tests switch it inside one process)
    if (want_rccl && shards_rccl_ready(h)) {
        RcclApi &R = rccl_api();
        int cur = 0; (void)hipGetDevice(&cur);
        bool ok = true;
        for (size_t i = 0; i < ns && ok; ++i) {
            store_norm(h->pinned + 12 * i, part[i]);
            ok = hipSetDevice(h->shards[i].device) == hipSuccess &&
                 hipMemcpyAsync(h->shards[i].xsend.p, h->pinned + 12 * i, 96, hipMemcpyHostToDevice, msm_job_stream(h->shards[i].job)) == hipSuccess;
        }
        ok = ok && R.GroupStart() == ncclSuccess;
        for (size_t i = 0; i < ns && ok; ++i)
            ok = hipSetDevice(h->shards[i].device) == hipSuccess &&
                 R.AllGather(h->shards[i].xsend.p, h->shards[i].xrecv.p, 12, ncclUint64, h->shards[i].comm, msm_job_stream(h->shards[i].job)) == ncclSuccess;
        ok = R.GroupEnd() == ncclSuccess && ok;
        uint64_t *got = h->pinned + 12 * ns;
        ok = ok && hipSetDevice(h->shards[0].device) == hipSuccess &&
             hipMemcpyAsync(got, h->shards[0].xrecv.p, ns * 96, hipMemcpyDeviceToHost, msm_job_stream(h->shards[0].job)) == hipSuccess;
        for (size_t i = 0; i < ns && ok; ++i) ok = hipSetDevice(h->shards[i].device) == hipSuccess && hipStreamSynchronize(msm_job_stream(h->shards[i].job)) == hipSuccess;
        (void)hipSetDevice(cur);
        if (!ok) { set_error("zkg_msm_g1_multi: the RCCL exchange failed"); return ZKG_ERROR; }
        if (memcmp(got, h->pinned, ns * 96) != 0) { set_error("zkg_msm_g1_multi: the gathered partials differ from the shards' own"); return ZKG_ERROR; }
        G1 acc = G1::inf();
        for (size_t i = 0; i < ns; ++i) { if (partials_jac) memcpy(partials_jac + 12 * i, got + 12 * i, 96); acc.add(load_norm_g1(got + 12 * i)); }
        store_norm(out_jac, acc);
        g_multi_rccl_calls.fetch_add(1);
        return ZKG_OK;
    }
    G1 acc = G1::inf();
    for (size_t i = 0; i < ns; ++i) { if (partials_jac) store_norm(partials_jac + 12 * i, part[i]); acc.add(part[i]); }
    store_norm(out_jac, acc);
    return ZKG_OK;
}
int zkg_msm_g1_multi(const zkg_msm_shards *h_, const uint64_t *scalars, uint64_t out_jac[12], uint64_t *partials_jac) {
    try { return zkg_msm_g1_multi_impl(h_, scalars, out_jac, partials_jac); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("zkg_msm_g1_multi: ") + e.what()); return ZKG_ERROR; }
    catch (...) { zk::set_error("zkg_msm_g1_multi: unexpected exception"); return ZKG_ERROR; }
}


void zkg_timing_reset(void) { g_dominant_timer.reset(); }
float zkg_timing_dominant_ms(int *launches) { return g_dominant_timer.drain(launches); }

}  // extern "C"
