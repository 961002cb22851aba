// codec.hip — libsnark proving-key blob -> device-resident CRS.
//
// zklaim moves keys around as byte blobs: libsnark_export_pk / libsnark_import_pk (/root/reference/zklaim/
// libsnark_wrapper.cpp:146-168) stream the key through libsnark's operator<< / operator>>, and libsnark_prove re-parses
// the whole blob on EVERY proof (libsnark_wrapper.cpp:230).  With point compression that parse costs one Fq (or Fq2) square
// root per curve point — about 4n + m of them — on one CPU thread.  Here the blob is walked once on the host (sizes are ASCII
// decimals, everything else fixed-size records), the compressed points are decompressed on the GPU (one lane per point:
// y = sqrt(x^3 + b), sign fixed by the stored parity bit), and the result is the same zkg_crs that zkg_crs_upload builds.
//
// Format restated from libsnark / libff defaults (BINARY_OUTPUT, MONTGOMERY_OUTPUT, point compression) — the serialisers
// live in the absent submodule lib/libsnark, so this layout is [UPSTREAM-RECALL] (SURVEY.md §7 hard part 3):
//   G1   : '0'|'1' (is_zero) , X (32 B Montgomery limbs) , '0'|'1' (lsb of canonical Y)                     34 B
//   G2   : '0'|'1' , X.c0 , X.c1 , '0'|'1' (lsb of canonical Y.c0)                                           66 B
//   pk   : alpha_g1 beta_g1 beta_g2 delta_g1 delta_g2 | A_query | B_query | H_query | L_query | constraint_system
//   vector<G1>                     : count '\n' , count x G1
//   knowledge_commitment_vector    : domain_size '\n' , #indices '\n' , (index '\n')* , #values '\n' , (G2 G1)*
//   r1cs_constraint_system         : primary '\n' auxiliary '\n' #constraints '\n' , per constraint a, b, c
//   linear_combination             : #terms '\n' , (index '\n' coeff(32 B))*
#include "common.hpp"
#include "../../include/zkg.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace zk {

// ---- square roots -------------------------------------------------------------------------------------------------
// q = 3 mod 4: sqrt(a) = a^((q+1)/4) when a is a square
ZK_D Fq fq_sqrt_candidate(const Fq &a) {
    // (q + 1) / 4
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    return a.pow(e, 8);
}
// complex method in Fq2 = Fq[u]/(u^2+1): returns false when a is not a square
ZK_D bool fq2_sqrt(const Fq2 &a, Fq2 &out) {
    if (a.c1.is_zero()) {                               // a in Fq: sqrt is either in Fq or purely imaginary
        Fq s = fq_sqrt_candidate(a.c0);
        if (s.sqr() == a.c0) { out = {s, Fq::zero()}; return true; }
        Fq t = fq_sqrt_candidate(a.c0.neg());
        if (t.sqr() == a.c0.neg()) { out = {Fq::zero(), t}; return true; }
        return false;
    }
    Fq norm = a.c0.sqr() + a.c1.sqr();
    Fq s = fq_sqrt_candidate(norm);
    if (s.sqr() != norm) return false;
    Fq two_inv = Fq::from_u64(2).inverse();
    Fq d = (a.c0 + s) * two_inv;
    Fq c0 = fq_sqrt_candidate(d);
    if (c0.sqr() != d) { d = (a.c0 - s) * two_inv; c0 = fq_sqrt_candidate(d); if (c0.sqr() != d) return false; }
    Fq c1 = a.c1 * (c0.dbl()).inverse();
    out = {c0, c1};
    return true;
}

ZK_D Fq load_fq_bytes(const uint8_t *p) {
    Fq r;
    for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    return r;
}

// records of `stride` bytes starting at rec: G1 at offset off (34 B).  flag |= 1 on a malformed point.
__global__ __launch_bounds__(256) void k_decompress_g1(const uint8_t *rec, size_t stride, size_t off, size_t n, G1Affine *out, uint32_t *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = rec + i * stride + off;
    if (p[0] == '1') { out[i] = G1Affine::inf(); return; }
    Fq x = load_fq_bytes(p + 1);
    Fq b; { Fq three = Fq::from_u64(3); b = three; }
    Fq rhs = x.sqr() * x + b;
    Fq y = fq_sqrt_candidate(rhs);
    if (y.sqr() != rhs || p[0] != '0') { atomicOr(flag, 1u); out[i] = G1Affine::inf(); return; }
    bool want_odd = p[33] == '1';
    if (((y.from_mont().v[0] & 1u) != 0) != want_odd) y = y.neg();
    out[i] = G1Affine{x, y}.normalized();
}
__global__ __launch_bounds__(256) void k_decompress_g2(const uint8_t *rec, size_t stride, size_t off, size_t n, G2Affine *out, uint32_t *flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = rec + i * stride + off;
    if (p[0] == '1') { out[i] = G2Affine::inf(); return; }
    Fq2 x = {load_fq_bytes(p + 1), load_fq_bytes(p + 33)};
    // twist coefficient b' = 3 / (9 + u), Montgomery limbs
    Fq2 b;
    { const uint32_t b0[8] = {0x77b802a8u, 0x3bf938e3u, 0x3633535du, 0x020b1b27u, 0x49755260u, 0x26b7edf0u, 0x4384a86du, 0x2514c632u};
      const uint32_t b1[8] = {0xd1dcff67u, 0x38e7ecccu, 0x93ce0d3eu, 0x65f0b37du, 0x22ac00aau, 0xd749d0ddu, 0x4a688d4du, 0x0141b9ceu};
      for (int k = 0; k < 8; ++k) { b.c0.v[k] = b0[k]; b.c1.v[k] = b1[k]; } }
    Fq2 rhs = x.sqr() * x + b, y;
    if (!fq2_sqrt(rhs, y) || p[0] != '0') { atomicOr(flag, 1u); out[i] = G2Affine::inf(); return; }
    bool want_odd = p[65] == '1';
    if (((y.c0.from_mont().v[0] & 1u) != 0) != want_odd) y = y.neg();
    out[i] = G2Affine{x, y}.normalized();
}

// ---- blob walker ---------------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t *p, *end; bool ok = true;
    size_t decimal() {                                   // ASCII digits terminated by '\n'
        size_t v = 0; int nd = 0;
        while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); ++p; if (++nd > 19) { ok = false; return 0; } }
        if (nd == 0 || p >= end || *p != '\n') { ok = false; return 0; }
        ++p;
        return v;
    }
    const uint8_t *take(size_t n) { if (!ok || (size_t)(end - p) < n) { ok = false; return nullptr; } const uint8_t *r = p; p += n; return r; }
    // a record count: refused unless `count` records of `record_bytes` each can still follow in the blob and count < 2^31, so that no
    // later product (count * stride, count * limbs, count * sizeof) can wrap and no 32-bit cast can truncate.  The blob comes from the
    // issuer (ctx->pk), i.e. it is not trusted input to the holder's prover.
    size_t count(size_t record_bytes) {
        size_t v = decimal();
        if (!ok) return 0;
        if (v >= ((size_t)1 << 31) || (record_bytes && v > (size_t)(end - p) / record_bytes)) { ok = false; return 0; }
        return v;
    }
    const uint8_t *take_records(size_t n, size_t record_bytes) { return take(n * record_bytes); }       // n came from count(record_bytes): no overflow
};

// n compressed records (stride bytes apart, the point at offset off) -> n affine points in d_out (device).  d_rec is staging for the records.
template <class A, class K>
static int decompress_dev(K kernel, const uint8_t *host_rec, size_t stride, size_t off, size_t n, A *d_out, DevBuf &d_rec, DevBuf &d_flag) {
    if (!n) return ZKG_OK;
    if (d_rec.reserve(n * stride) || d_flag.reserve(4)) return ZKG_ERROR;
    ZK_HIP(hipMemcpy(d_rec.p, host_rec, n * stride, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_rec.as<uint8_t>(), stride, off, n, d_out, d_flag.as<uint32_t>());
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}

// ---- the other direction: affine points on the device -> the blob's compressed records (libsnark_export_pk's operator<<,
//      /root/reference/zklaim/libsnark_wrapper.cpp:146-157).  The seam's key generator leaves its ~4n + m points on the device (they are the
//      resident key of the proofs that follow) and only these records — 34 / 100 bytes instead of 64 / 192 — travel to the host.
//      A workgroup assembles its 256 records in LDS (byte stores) and writes them out as whole words: a run of records starts 16-byte
//      aligned in the staging buffer and 256 records are a multiple of four bytes.
ZK_D void put_fq_bytes(uint8_t *p, const Fq &x) {
    for (int i = 0; i < 8; ++i) { p[4 * i] = (uint8_t)x.v[i]; p[4 * i + 1] = (uint8_t)(x.v[i] >> 8); p[4 * i + 2] = (uint8_t)(x.v[i] >> 16); p[4 * i + 3] = (uint8_t)(x.v[i] >> 24); }
}
ZK_D void put_g1_record(uint8_t *o, const G1Affine &a) {                  // ser::put_g1
    const bool inf = a.is_inf();
    const Fq x = inf ? Fq::zero() : a.x.normalized(), y = inf ? Fq::one() : a.y;
    o[0] = inf ? '1' : '0'; put_fq_bytes(o + 1, x); o[33] = (y.from_mont().v[0] & 1u) ? '1' : '0';
}
ZK_D void put_g2_record(uint8_t *o, const G2Affine &a) {                  // ser::put_g2
    const bool inf = a.is_inf();
    const Fq2 x = inf ? Fq2::zero() : a.x.normalized(); const Fq y0 = inf ? Fq::one() : a.y.c0;
    o[0] = inf ? '1' : '0'; put_fq_bytes(o + 1, x.c0); put_fq_bytes(o + 33, x.c1); o[65] = (y0.from_mont().v[0] & 1u) ? '1' : '0';
}
// REC = 34: record i = G1 in1[i].  REC = 100: record j = G2 in2[idx[j]] then G1 in1[idx[j]] (a knowledge commitment of the sparse B query)
template <int REC>
__global__ __launch_bounds__(256) void k_compress_records(const G1Affine *in1, const G2Affine *in2, const uint32_t *idx, size_t n, uint8_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[256 * REC];
    const size_t first = (size_t)blockIdx.x * 256, i = first + threadIdx.x;
    if (i < n) {
        uint8_t *o = stage + (size_t)threadIdx.x * REC;
        if constexpr (REC == 34) put_g1_record(o, in1[i]);
        else { const uint32_t e = idx[i]; put_g2_record(o, in2[e]); put_g1_record(o + 66, in1[e]); }
    }
    __syncthreads();
    const size_t bytes = (n - first < 256 ? n - first : 256) * REC;         // this block's run; its start, first * REC, is a multiple of 4
    uint8_t *dst = out + first * REC;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(stage);
    for (size_t k = threadIdx.x; k < bytes / 4; k += 256) reinterpret_cast<uint32_t *>(dst)[k] = w[k];
    for (size_t k = (bytes & ~(size_t)3) + threadIdx.x; k < bytes; k += 256) dst[k] = stage[k];
}
int compress_g1_records(const G1Affine *d_in, size_t n, uint8_t *d_out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_compress_records<34>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (const G2Affine *)nullptr, (const uint32_t *)nullptr, n, d_out);
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}
int compress_kc_records(const G2Affine *d_g2, const G1Affine *d_g1, const uint32_t *d_idx, size_t nidx, uint8_t *d_out, hipStream_t s) {
    if (nidx) hipLaunchKernelGGL(k_compress_records<100>, dim3((unsigned)((nidx + 255) / 256)), dim3(256), 0, s, d_g1, d_g2, d_idx, nidx, d_out);
    return hipGetLastError() == hipSuccess ? ZKG_OK : ZKG_ERROR;
}

}  // namespace zk

using namespace zk;
zkg_crs *crs_upload_device_queries(const zkg_pk *pk, const std::function<bool()> &constraint_system_ready);   // prover.hip

// The constraint system section of a pk blob -> CSR arrays.  Runs on a thread of its own while the GPU decompresses the points and builds the
// H table: the section is a serial walk (decimal counts and indices between 32-byte coefficients), 70 ms at 20 payloads.
struct ParsedSystem { std::vector<uint32_t> rp[3], col[3]; std::vector<uint64_t> val[3]; std::string error; bool ok = false; };
static void parse_constraint_system(Reader rd, size_t ncons, size_t n_elements, ParsedSystem &out) {
    try {
        const size_t bound = (size_t)(rd.end - rd.p) / 34;                  // a term takes at least 34 bytes: no more than this many in total
        for (int m = 0; m < 3; ++m) { out.rp[m].reserve(ncons + 1); out.rp[m].push_back(0); out.col[m].reserve(bound / 2); out.val[m].reserve(bound * 2); }
        for (size_t c = 0; c < ncons; ++c)
            for (int m = 0; m < 3; ++m) {
                size_t nt = rd.count(34);                                   // a term: index digits, newline, 32-byte coefficient
                for (size_t t = 0; t < nt && rd.ok; ++t) {
                    size_t index = rd.decimal(); const uint8_t *coeff = rd.take(32);
                    if (!rd.ok || index >= n_elements) { out.error = "pk blob: bad linear term"; return; }
                    out.col[m].push_back((uint32_t)index);
                    uint64_t l4[4]; memcpy(l4, coeff, 32); out.val[m].insert(out.val[m].end(), l4, l4 + 4);
                }
                if (!rd.ok) { out.error = "pk blob: truncated constraint"; return; }
                out.rp[m].push_back((uint32_t)out.col[m].size());
            }
        out.ok = true;
    } catch (const std::exception &e) { out.error = std::string("pk blob: ") + e.what(); }
}

// Where every section of a pk blob lies: the fixed head, the four queries (record runs), the sparse B vector's index list, the constraint
// system's sizes; rd is left at the first constraint.  Host only (pointer arithmetic and decimal counts), shared by the loader and
// zkg_pk_blob_inspect.
struct Sections {
    const uint8_t *alpha = nullptr, *beta1 = nullptr, *beta2 = nullptr, *delta1 = nullptr, *delta2 = nullptr, *recA = nullptr, *recB = nullptr, *recH = nullptr, *recL = nullptr;
    size_t nA = 0, nH = 0, nL = 0, primary = 0, auxiliary = 0, ncons = 0;
    std::vector<uint32_t> idx; DomainShape shape;
};
static bool walk_sections(Reader &rd, Sections &s) {
    s.alpha = rd.take(34); s.beta1 = rd.take(34); s.beta2 = rd.take(66); s.delta1 = rd.take(34); s.delta2 = rd.take(66);
    if (!rd.ok) { set_error("pk blob: truncated head"); return false; }
    s.nA = rd.count(34); s.recA = rd.take_records(s.nA, 34);
    if (!rd.ok || s.nA == 0) { set_error("pk blob: bad A_query"); return false; }
    // B_query (sparse knowledge commitments: G2 then G1 per value)
    size_t domain = rd.count(0), nidx = rd.count(2);                      // an index is at least one digit and a newline
    if (!rd.ok || domain != s.nA || nidx > domain) { set_error("pk blob: bad B_query header"); return false; }
    s.idx.resize(nidx);
    for (size_t i = 0; i < nidx; ++i) { size_t v = rd.decimal(); if (!rd.ok || v >= domain) { set_error("pk blob: bad B_query index"); return false; } s.idx[i] = (uint32_t)v; }
    size_t nval = rd.count(100); s.recB = rd.take_records(nval, 100);
    if (!rd.ok || nval != nidx) { set_error("pk blob: bad B_query values"); return false; }
    s.nH = rd.count(34); s.recH = rd.take_records(s.nH, 34);
    s.nL = rd.ok ? rd.count(34) : 0; s.recL = rd.take_records(s.nL, 34);
    if (!rd.ok) { set_error("pk blob: bad H/L query"); return false; }
    s.primary = rd.count(0); s.auxiliary = rd.count(0); s.ncons = rd.count(6);       // a constraint is at least three "0\n" term counts
    if (!rd.ok || s.primary + s.auxiliary + 1 != s.nA || s.nL != s.auxiliary) { set_error("pk blob: constraint system sizes disagree with the queries"); return false; }
    // the domain size is not stored in the blob: H_query has m - 1 entries
    if (!domain_shape_of(s.nH + 1, s.shape)) { set_error("pk blob: H_query length + 1 is neither a power of two nor a step_radix2 size 2^a + 2^b"); return false; }
    return true;
}

static zkg_crs *crs_upload_blob_impl(const void *blob, size_t len) {
    if (!blob || len < 34 * 3 + 66 * 2) { set_error("zkg_crs_upload_blob: blob too short"); return nullptr; }
    Reader rd{(const uint8_t *)blob, (const uint8_t *)blob + len};
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) fprintf(stderr, "[zkg key blob] %-26s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    // ---- pass 1 (host, pointer arithmetic and the sparse vector's index list): where every section lies
    Sections sec;
    if (!walk_sections(rd, sec)) return nullptr;
    const uint8_t *alpha = sec.alpha, *beta1 = sec.beta1, *beta2 = sec.beta2, *delta1 = sec.delta1, *delta2 = sec.delta2;
    const uint8_t *recA = sec.recA, *recB = sec.recB, *recH = sec.recH, *recL = sec.recL;
    const size_t nA = sec.nA, domain = sec.nA, nidx = sec.idx.size(), nval = nidx, nH = sec.nH, nL = sec.nL, primary = sec.primary, ncons = sec.ncons;
    const std::vector<uint32_t> &idx = sec.idx;
    const DomainShape shape = sec.shape;
    // ---- the constraint system on its own thread ...
    ParsedSystem cs;
    std::thread parser(parse_constraint_system, rd, ncons, nA, std::ref(cs));
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_guard{parser};       // on every return path, also an exception's
    // ---- ... while the GPU decompresses the points (one lane per point) straight into the buffers the resident key is built from
    // (scoped: hundreds of MB of staging are handed back on every way out of this function, an exception's unwinding included)
    ScopedDevBuf d_rec, d_flag, d_small, dA, dB1, dB2, dBv1, dBv2, dH, dL, d_idx;
    auto fail = [&](const char *msg) -> zkg_crs * {
        if (msg) set_error(msg);
        return nullptr;
    };
    if (d_flag.reserve(4) || d_small.reserve(3 * 64 + 2 * 128) || dA.reserve(nA * 64) || dB1.reserve(domain * 64) || dB2.reserve(domain * 128) || dBv1.reserve(nval * 64 + 16) ||
        dBv2.reserve(nval * 128 + 16) || dH.reserve(nH * 64 + 16) || dL.reserve(nL * 64 + 16) || d_idx.reserve(nidx * 4 + 16)) return fail(nullptr);
    if (!hip_ok(hipMemset(d_flag.p, 0, 4), "memset", __FILE__, __LINE__) || !hip_ok(hipMemset(dB1.p, 0, domain * 64), "memset", __FILE__, __LINE__) ||
        !hip_ok(hipMemset(dB2.p, 0, domain * 128), "memset", __FILE__, __LINE__)) return fail(nullptr);             // dense B: absent = infinity = zero bytes
    uint8_t head1[3 * 34], head2[2 * 66];
    memcpy(head1, alpha, 34); memcpy(head1 + 34, beta1, 34); memcpy(head1 + 68, delta1, 34);
    memcpy(head2, beta2, 66); memcpy(head2 + 66, delta2, 66);
    G1Affine *small1_dev = d_small.as<G1Affine>(); G2Affine *small2_dev = reinterpret_cast<G2Affine *>(small1_dev + 3);
    if (decompress_dev<G1Affine>(k_decompress_g1, head1, 34, 0, 3, small1_dev, d_rec, d_flag) ||
        decompress_dev<G2Affine>(k_decompress_g2, head2, 66, 0, 2, small2_dev, d_rec, d_flag) ||
        decompress_dev<G1Affine>(k_decompress_g1, recA, 34, 0, nA, dA.as<G1Affine>(), d_rec, d_flag) ||
        decompress_dev<G2Affine>(k_decompress_g2, recB, 100, 0, nval, dBv2.as<G2Affine>(), d_rec, d_flag) ||
        decompress_dev<G1Affine>(k_decompress_g1, recB, 100, 66, nval, dBv1.as<G1Affine>(), d_rec, d_flag)) return fail(nullptr);
    if (nidx) {
        if (!hip_ok(hipMemcpy(d_idx.p, idx.data(), nidx * 4, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__) ||
            scatter_points_g1(dBv1.as<G1Affine>(), d_idx.as<uint32_t>(), nidx, dB1.as<G1Affine>(), nullptr) ||
            scatter_points_g2(dBv2.as<G2Affine>(), d_idx.as<uint32_t>(), nidx, dB2.as<G2Affine>(), nullptr)) return fail(nullptr);
    }
    if (decompress_dev<G1Affine>(k_decompress_g1, recH, 34, 0, nH, dH.as<G1Affine>(), d_rec, d_flag) ||
        decompress_dev<G1Affine>(k_decompress_g1, recL, 34, 0, nL, dL.as<G1Affine>(), d_rec, d_flag)) return fail(nullptr);
    uint32_t flag = 0; uint64_t small[3 * 8 + 2 * 16];
    if (!hip_ok(hipMemcpy(&flag, d_flag.p, 4, hipMemcpyDeviceToHost), "D2H", __FILE__, __LINE__) ||
        !hip_ok(hipMemcpy(small, d_small.p, sizeof(small), hipMemcpyDeviceToHost), "D2H", __FILE__, __LINE__)) return fail(nullptr);
    if (flag) return fail("pk blob: a compressed point is not on the curve");
    lap("points decompressed (GPU)");
    zkg_pk pk; memset(&pk, 0, sizeof(pk));
    pk.cs.num_variables = (uint32_t)(nA - 1); pk.cs.num_inputs = (uint32_t)primary; pk.cs.num_constraints = (uint32_t)ncons;
    pk.log_m = shape.log_m; pk.domain_size = (uint32_t)shape.m;
    pk.alpha_g1 = small; pk.beta_g1 = small + 8; pk.delta_g1 = small + 16;
    pk.beta_g2 = small + 24; pk.delta_g2 = small + 40;
    pk.A_query = dA.as<uint64_t>(); pk.B_g1 = dB1.as<uint64_t>(); pk.B_g2 = dB2.as<uint64_t>(); pk.H_query = dH.as<uint64_t>(); pk.L_query = dL.as<uint64_t>();   // DEVICE pointers
    auto system_ready = [&]() -> bool {
        parser.join();
        lap("constraint system parsed");
        if (!cs.ok) { set_error(cs.error.empty() ? "pk blob: bad constraint system" : cs.error.c_str()); return false; }
        pk.cs.a_rowptr = cs.rp[0].data(); pk.cs.a_col = cs.col[0].data(); pk.cs.a_val = cs.val[0].data();
        pk.cs.b_rowptr = cs.rp[1].data(); pk.cs.b_col = cs.col[1].data(); pk.cs.b_val = cs.val[1].data();
        pk.cs.c_rowptr = cs.rp[2].data(); pk.cs.c_col = cs.col[2].data(); pk.cs.c_val = cs.val[2].data();
        return true;
    };
    zkg_crs *crs = crs_upload_device_queries(&pk, system_ready);
    lap("resident key built");
    return crs;                                                                 // (the staging buffers are released by their scope)
}

// Host-only walk of a pk blob, the loader's parsing without the GPU: sections, index list, every constraint's terms.  out (optional):
// A_query entries, B_query values, H_query entries, L_query entries, public inputs, constraints, terms in A + B + C, domain size.
// ZKG_OK or ZKG_ERROR (zkg_last_error says what was wrong); never reads outside [blob, blob + len).
static int pk_blob_inspect_impl(const void *blob, size_t len, uint64_t out[8]) {
    if (!blob) { set_error("zkg_pk_blob_inspect: null blob"); return ZKG_ERROR; }
    Reader rd{(const uint8_t *)blob, (const uint8_t *)blob + len};
    Sections sec;
    if (!walk_sections(rd, sec)) return ZKG_ERROR;
    ParsedSystem cs;
    parse_constraint_system(rd, sec.ncons, sec.nA, cs);
    if (!cs.ok) { set_error(cs.error.empty() ? "pk blob: bad constraint system" : cs.error.c_str()); return ZKG_ERROR; }
    if (out) {
        out[0] = sec.nA; out[1] = sec.idx.size(); out[2] = sec.nH; out[3] = sec.nL; out[4] = sec.primary; out[5] = sec.ncons;
        out[6] = cs.col[0].size() + cs.col[1].size() + cs.col[2].size(); out[7] = sec.shape.m;
    }
    return ZKG_OK;
}
extern "C" int zkg_pk_blob_inspect(const void *blob, size_t len, uint64_t out[8]) {
    try { return pk_blob_inspect_impl(blob, len, out); }
    catch (const std::exception &e) { set_error(std::string("zkg_pk_blob_inspect: ") + e.what()); return ZKG_ERROR; }
    catch (...) { set_error("zkg_pk_blob_inspect: unexpected exception"); return ZKG_ERROR; }
}

// Nothing may propagate through the C boundary: allocation failures on hostile sizes end up here as an error return.
extern "C" zkg_crs *zkg_crs_upload_blob(const void *blob, size_t len) {
    try { return crs_upload_blob_impl(blob, len); }
    catch (const std::exception &e) { set_error(std::string("zkg_crs_upload_blob: ") + e.what()); return nullptr; }
    catch (...) { set_error("zkg_crs_upload_blob: unexpected exception"); return nullptr; }
}
