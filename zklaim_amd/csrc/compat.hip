// compat.hip — the seam zklaim's C front-end links against, unchanged in name, signature and return codes:
//     int libsnark_trusted_setup(zklaim_ctx*), int libsnark_prove(zklaim_ctx*), int libsnark_verify(zklaim_ctx*)
// (declared /root/reference/zklaim/zklaim.h:257-259, defined zklaim/libsnark_wrapper.cpp:195-276, called from
// zklaim_trusted_setup / zklaim_proof_generate / zklaim_proof_verify at zklaim/zklaim.c:77-91).
// Behaviour kept: 0 on success; prove returns 1 for an unsatisfied credential (libsnark_wrapper.cpp:233-240); verify returns
// !valid (:269); ctx->pk / vk / proof are malloc'd here and freed by zklaim_ctx_free (zklaim.c:57-72).
// Behaviour changed on purpose: file descriptor 1 is never closed (the reference closes stdout around every call,
// :199-203,220-225,254-258), nothing is re-initialised per call, and the parsed key stays resident on the GPU between proofs
// of the same ctx->pk instead of being re-parsed (:230).
#include "common.hpp"
#include <cstdio>
#include <chrono>
#include "../../include/zkg.h"
#include "../../include/zklaim_abi.h"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <future>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <random>

using namespace zk;

extern "C" {
struct zkg_keypair;
zkg_circuit *zkg_zklaim_witness_new(const zklaim_ctx *ctx);
zkg_keypair *zkg_groth16_setup(const zkg_r1cs *cs, const uint64_t *trapdoor);
void zkg_keypair_free(zkg_keypair *kp);
size_t zkg_keypair_pk_blob(const zkg_keypair *kp, uint8_t *out, size_t cap);
size_t zkg_keypair_vk_blob(const zkg_keypair *kp, uint8_t *out, size_t cap);
int zkg_groth16_verify(const uint8_t *vk_blob, size_t vk_len, const uint64_t *primary_input, size_t n_inputs, const uint8_t *proof, size_t proof_len);
}

int seam_keygen(const zkg_r1cs *cs, zk::OwnedCsr *owned, const std::function<void()> &under_gpu, unsigned char **pk_out, size_t *pk_len,
                unsigned char **vk_out, size_t *vk_len, zkg_crs **crs_out, const std::function<void(const unsigned char *, size_t)> &on_blob);    // setup_verify.hip
void seam_keygen_quiesce();                                                                                              // setup_verify.hip
void circuit_release_csr(zkg_circuit *c, zk::OwnedCsr &out);                                                            // zklaim_circuit.hip

namespace {

// g_mu guards g_inited and the key cache's MAP only — never a key load, a witness pass or a proof.  Callers of one resident key meet again
// at that key's prover slots (prover.hip SlotLease: up to three proofs in flight below m = 2^18), callers of different keys nowhere.
std::mutex g_mu;
std::condition_variable g_cv;                                // a key that was being loaded has finished loading (or failed)
bool g_inited = false;
struct Digest128 { uint64_t a = 0, b = 0; bool operator==(const Digest128 &o) const { return a == o.a && b == o.b; } };
// One resident key.  `loading`: the thread that inserted the entry is still parsing / uploading the blob; later callers of the same key wait
// for it instead of uploading a second copy.  The key itself is shared: an entry evicted from the map (at most four stay) lives on until
// the last proof using it has returned.
struct CachedCrs { size_t size = 0; Digest128 full; std::shared_ptr<zkg_crs> crs; bool loading = true, failed = false; };
std::map<uint64_t, std::shared_ptr<CachedCrs>> g_crs_cache;    // one upload per key, not per proof (the reference re-parses ctx->pk on every call)
std::shared_ptr<zkg_crs> share_crs(zkg_crs *c) { return std::shared_ptr<zkg_crs>(c, [](zkg_crs *p) { zkg_crs_free(p); }); }

int seam_device() { static const int d = [] { const char *dev = getenv("ZKG_DEVICE"); return dev ? atoi(dev) : 0; }(); return d; }
int ensure_init() {                                          // caller holds g_mu
    if (!g_inited) {
        if (zkg_init(seam_device())) return 1;
        g_inited = true;
    }
    // the current device is per thread: every thread that enters the seam is bound to the seam's device once
    static thread_local bool bound = false;
    if (!bound) { if (hipSetDevice(seam_device()) != hipSuccess) return 1; bound = true; }
    return 0;
}
// Two digests of a pk blob.  `sampled` is the cache's lookup key: head, tail and 128 strided windows (80 KB), microseconds on a
// 10-200 MB blob.  It cannot see an edit between its windows, so a hit is only trusted once `full` — every byte, 128 bits, chunks
// hashed on the host pool — equals the digest recorded when the resident key was uploaded; libsnark_prove computes it on a helper
// thread while the GPU already proves with the candidate key and discards that proof on a mismatch.
inline uint64_t mix64(uint64_t h, uint64_t w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
uint64_t sampled_digest(const unsigned char *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](size_t lo, size_t hi) {
        if (hi > n) hi = n;
        size_t i = lo;
        for (; i + 8 <= hi; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = mix64(h, w); }
        for (; i < hi; ++i) h = mix64(h, p[i]);
    };
    if (n <= (1u << 16)) eat(0, n);
    else {
        eat(0, 8192); eat(n - 8192, n);
        size_t step = n / 128;
        for (size_t k = 1; k < 128; ++k) eat(k * step, k * step + 512);
    }
    return h ^ (n * 0x9E3779B97F4A7C15ull);
}
Digest128 full_digest(const unsigned char *p, size_t n) {
    const size_t CHUNK = (size_t)1 << 18;
    const size_t nchunks = (n + CHUNK - 1) / CHUNK;
    std::vector<Digest128> part(nchunks);
    host_parallel_for_wait((int)nchunks, [&](int c) {
        const size_t lo = (size_t)c * CHUNK, hi = std::min(n, lo + CHUNK);
        uint64_t l[4] = {0x243F6A8885A308D3ull ^ (uint64_t)c, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};    // four independent lanes
        size_t i = lo;
        for (; i + 32 <= hi; i += 32) { uint64_t w[4]; memcpy(w, p + i, 32); for (int k = 0; k < 4; ++k) l[k] = mix64(l[k], w[k]); }
        for (; i < hi; ++i) l[0] = mix64(l[0], p[i]);
        part[c] = {mix64(mix64(l[0], l[1]), l[2] + 0x9E37ull), mix64(mix64(l[3], l[2]), l[0] + 0x79B9ull)};
    });
    Digest128 d{n * 0x9E3779B97F4A7C15ull, ~(uint64_t)n};
    for (const Digest128 &x : part) { d.a = mix64(d.a, x.a); d.b = mix64(d.b ^ x.a, x.b); }
    return d;
}
void random_fr_mont(uint64_t out[4]) {
    static thread_local std::random_device rd;               // opened once per thread, not once per scalar
    for (;;) {
        uint32_t v[8]; for (auto &x : v) x = rd();
        v[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; --i) if (v[i] != FrParams::P[i]) { lt = v[i] < FrParams::P[i]; break; }
        if (lt) { memcpy(out, v, 32); return; }
    }
}

}  // namespace

extern "C" {

static int libsnark_trusted_setup_impl(zklaim_ctx *ctx) {
    if (!ctx) return ZKLAIM_ERROR;
    { std::lock_guard<std::mutex> lk(g_mu); if (ensure_init()) return ZKLAIM_ERROR; }
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) fprintf(stderr, "[zkg seam setup] %-24s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    zkg_circuit *ck = zkg_zklaim_circuit_new(ctx, 0);             // generate_keypair: gadget + constraints only (snark.cpp:76-92)
    if (!ck) return ZKLAIM_ERROR;
    lap("circuit built");
    zkg_r1cs cs;
    if (zkg_circuit_r1cs(ck, &cs) != 0) { zkg_circuit_free(ck); return ZKLAIM_ERROR; }
    // the generator takes the circuit's CSR matrices over (no copy), and what is left of the circuit — the constraint arena and the
    // variable store — is destroyed on a side thread while the GPU turns the generator's scalars into points
    zk::OwnedCsr csr;
    circuit_release_csr(ck, csr);
    cs.a_rowptr = cs.b_rowptr = cs.c_rowptr = nullptr; cs.a_col = cs.b_col = cs.c_col = nullptr; cs.a_val = cs.b_val = cs.c_val = nullptr;
    bool freed = false;
    unsigned char *pk = nullptr, *vk = nullptr; size_t pk_len = 0, vk_len = 0; zkg_crs *crs = nullptr;
    uint64_t key = 0; Digest128 full;
    // the reference's protocol proves with the key it has just generated (main_benchmark.c:113-140): the generator's points stay on the GPU
    // as that key, already resident when libsnark_prove looks ctx->pk up — by the digests of the very bytes handed to the caller
    const int rc_gen = seam_keygen(&cs, &csr, [&] { zkg_circuit_free(ck); freed = true; }, &pk, &pk_len, &vk, &vk_len, &crs,
                                   [&](const unsigned char *blob, size_t len) { key = sampled_digest(blob, len); full = full_digest(blob, len); });
    if (!freed) zkg_circuit_free(ck);                            // the generator failed before its GPU phase
    if (rc_gen != ZKG_OK) return ZKLAIM_ERROR;
    lap("keys generated");
    ctx->vk = vk; ctx->vk_size = vk_len; ctx->pk = pk; ctx->pk_size = pk_len;                     // libsnark_wrapper.cpp:207-208
    std::shared_ptr<zkg_crs> shared = share_crs(crs);
    // ZKG_SEAM_ISSUER_ONLY: a process that only issues keys (never proves) keeps nothing on the GPU after the call — no resident key
    // (2.4 GB at m = 2^20), no warm-up proof; a prover that later receives the blob loads it in its first libsnark_prove (14 - 86 ms)
    static const bool issuer_only = getenv("ZKG_SEAM_ISSUER_ONLY") != nullptr;
    static const bool no_warm = getenv("ZKG_SEAM_NO_WARM") != nullptr || issuer_only;
    if (issuer_only) { lap("issuer only: key not kept"); return ZKLAIM_OK; }
    if (!no_warm) {
        // Key preparation that needs a witness' SHAPE: which variables are not bits decides what the witness tables cover, and the first
        // proof on a key allocates every stream's workspace.  The ctx at hand names a credential of exactly this shape (the issuer's
        // own, in the reference's flow the very one proved next), so one proof is run on it and thrown away; a ctx without pre-images
        // still has the shape.  Failure here is not an error of the setup: the first real proof then does this work itself.
        zkg_circuit *wk = zkg_zklaim_witness_new(ctx);
        const uint8_t *tags = nullptr; const uint32_t *fidx = nullptr; const uint64_t *fval = nullptr; size_t nfull = 0;
        if (wk && zkg_circuit_sparse_witness(wk, &tags, &fidx, &fval, &nfull) == ZKG_OK && zkg_circuit_num_variables(wk) == zkg_crs_num_variables(crs)) {
            uint64_t r[4], s2[4]; unsigned char scratch[ZKG_PROOF_BYTES]; size_t len = 0;
            random_fr_mont(r); random_fr_mont(s2);
            (void)zkg_groth16_prove_sparse(crs, tags, fidx, fval, nfull, r, s2, 0, scratch, &len);
        }
        if (wk) zkg_circuit_free(wk);
        lap("key warmed");
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_crs_cache.size() >= 4) g_crs_cache.clear();
        auto entry = std::make_shared<CachedCrs>();
        entry->size = pk_len; entry->full = full; entry->crs = shared; entry->loading = false;
        auto it = g_crs_cache.find(key);
        if (it == g_crs_cache.end() || !it->second->loading) g_crs_cache[key] = entry;
    }
    lap("key resident");
    return ZKLAIM_OK;
}
int libsnark_trusted_setup(zklaim_ctx *ctx) {
    try { return libsnark_trusted_setup_impl(ctx); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("libsnark_trusted_setup: ") + e.what()); return ZKLAIM_ERROR; }
    catch (...) { zk::set_error("libsnark_trusted_setup: unexpected exception"); return ZKLAIM_ERROR; }
}


// The resident key of ctx->pk: found in the cache (a hit by the sampled digest is `speculative` until the full digest confirms it), or
// loaded by THIS caller — with the map's lock released — while later callers of the same key wait for that load instead of repeating it.
static std::shared_ptr<zkg_crs> resident_key(const zklaim_ctx *ctx, uint64_t key, bool &speculative, Digest128 &recorded_full) {
    std::shared_ptr<CachedCrs> entry;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        if (ensure_init()) return nullptr;
        for (;;) {
            auto it = g_crs_cache.find(key);
            if (it == g_crs_cache.end() || it->second->size != ctx->pk_size) break;
            std::shared_ptr<CachedCrs> e = it->second;
            g_cv.wait(lk, [&] { return !e->loading; });
            if (!e->failed) { speculative = true; recorded_full = e->full; return e->crs; }
            // the load failed (its loader has erased the entry, or another caller has replaced it): look again
        }
        if (g_crs_cache.size() >= 4) g_crs_cache.clear();                 // at most four resident keys; keys in use live on through their shared owners
        entry = std::make_shared<CachedCrs>();
        entry->size = ctx->pk_size;
        g_crs_cache[key] = entry;                                          // (replaces an entry of another size under the same sampled digest)
    }
    // a new key: its full digest (every byte) is computed while the key loads
    std::shared_ptr<zkg_crs> crs;
    Digest128 full;
    try {
        std::future<Digest128> digest = std::async(std::launch::async, [&] { return full_digest(ctx->pk, ctx->pk_size); });
        zkg_crs *c = zkg_crs_upload_blob(ctx->pk, ctx->pk_size);
        full = digest.get();
        if (c) crs = share_crs(c);
    } catch (...) { crs.reset(); }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        entry->loading = false;
        if (crs) { entry->crs = crs; entry->full = full; }
        else {
            entry->failed = true;
            auto it = g_crs_cache.find(key);
            if (it != g_crs_cache.end() && it->second == entry) g_crs_cache.erase(it);
        }
    }
    g_cv.notify_all();
    speculative = false;
    return crs;
}
// a speculative hit turned out to be another key (same size and samples, different bytes): load this one and put it in the other's place
static std::shared_ptr<zkg_crs> replace_key(const zklaim_ctx *ctx, uint64_t key, const Digest128 &full) {
    zkg_crs *c = zkg_crs_upload_blob(ctx->pk, ctx->pk_size);
    if (!c) return nullptr;
    std::shared_ptr<zkg_crs> crs = share_crs(c);
    auto entry = std::make_shared<CachedCrs>();
    entry->size = ctx->pk_size; entry->full = full; entry->crs = crs; entry->loading = false;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_crs_cache.find(key);
    if (it == g_crs_cache.end() || !it->second->loading) g_crs_cache[key] = entry;      // (an entry someone is loading right now is theirs to finish)
    return crs;
}

static int libsnark_prove_impl(zklaim_ctx *ctx) {
    if (!ctx || !ctx->pk || !ctx->pk_size) return ZKLAIM_ERROR;
    const uint64_t key = sampled_digest(ctx->pk, ctx->pk_size);
    bool speculative = false;                                    // a cache hit by the sampled digest: confirmed by the full one below
    Digest128 recorded_full;
    std::shared_ptr<zkg_crs> crs = resident_key(ctx, key, speculative, recorded_full);
    if (!crs) return ZKLAIM_ERROR;
    // the witness only: the constraint system already sits on the GPU inside the resident key, and pb.is_satisfied()
    // (snark.cpp:121-124) is evaluated there, fused with the R1CS mat-vec of the prover (check_satisfied = 1)
    zkg_circuit *ck = zkg_zklaim_witness_new(ctx);
    if (!ck) return ZKLAIM_ERROR;
    int rc = ZKLAIM_ERROR;
    uint64_t r[4], s[4];
    random_fr_mont(r); random_fr_mont(s);
    unsigned char *proof = (unsigned char *)malloc(ZKG_PROOF_BYTES);
    size_t len = 0;
    const uint8_t *tags = nullptr; const uint32_t *fidx = nullptr; const uint64_t *fval = nullptr; size_t nfull = 0;
    int prc = ZKG_ERROR;                                         // the witness goes up as tags + the few non-bit values (30x less PCIe traffic)
    if (proof && zkg_circuit_sparse_witness(ck, &tags, &fidx, &fval, &nfull) == ZKG_OK) {
        std::future<Digest128> confirm;
        if (speculative) confirm = std::async(std::launch::async, [&] { return full_digest(ctx->pk, ctx->pk_size); });     // under the GPU's work
        // a key made for another payload count has another variable count: refuse instead of reading past the witness
        if (zkg_circuit_num_variables(ck) != zkg_crs_num_variables(crs.get())) set_error("libsnark_prove: ctx->pk was generated for a different circuit (variable count differs)");
        else prc = zkg_groth16_prove_sparse(crs.get(), tags, fidx, fval, nfull, r, s, 1, proof, &len);
        if (speculative) {
            const Digest128 full = confirm.get();
            if (!(full == recorded_full)) {                      // same size and samples, different bytes: not the resident key after all
                prc = ZKG_ERROR;
                crs = replace_key(ctx, key, full);
                if (crs && zkg_circuit_num_variables(ck) == zkg_crs_num_variables(crs.get())) prc = zkg_groth16_prove_sparse(crs.get(), tags, fidx, fval, nfull, r, s, 1, proof, &len);
            }
        }
    }
    if (prc == ZKG_OK) { ctx->proof = proof; ctx->proof_size = len; rc = ZKLAIM_OK; }          // libsnark_wrapper.cpp:242
    else { free(proof); if (prc == ZKG_UNSATISFIED) rc = 1; }                                  // "system not satisfied!! not creating proof." -> 1
    zkg_circuit_free(ck);
    return rc;
}
int libsnark_prove(zklaim_ctx *ctx) {
    try { return libsnark_prove_impl(ctx); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("libsnark_prove: ") + e.what()); return ZKLAIM_ERROR; }
    catch (...) { zk::set_error("libsnark_prove: unexpected exception"); return ZKLAIM_ERROR; }
}


static int libsnark_verify_impl(zklaim_ctx *ctx) {
    if (!ctx || !ctx->vk || !ctx->proof) return 1;
    size_t n = zkg_zklaim_input_map(ctx, nullptr, 0);
    std::vector<uint64_t> input(4 * n + 4);
    zkg_zklaim_input_map(ctx, input.data(), n);                   // verify_proof: input = zklaim_input_map(ctx) (snark.cpp:58-62)
    return zkg_groth16_verify(ctx->vk, ctx->vk_size, input.data(), n, ctx->proof, ctx->proof_size) == 0 ? 0 : 1;
}
int libsnark_verify(zklaim_ctx *ctx) {
    try { return libsnark_verify_impl(ctx); }                      // nothing propagates through the C boundary
    catch (const std::exception &e) { zk::set_error(std::string("libsnark_verify: ") + e.what()); return 1; }
    catch (...) { zk::set_error("libsnark_verify: unexpected exception"); return 1; }
}


// drops the resident keys cached by libsnark_prove (tests / long-running hosts)
void zkg_compat_reset(void) {
    seam_keygen_quiesce();                                       // a finished generator's keypair may still be on its way back to the allocators
    std::lock_guard<std::mutex> lk(g_mu);
    g_crs_cache.clear();                                         // (a key still proving lives on until that proof returns)
}

}  // extern "C"
