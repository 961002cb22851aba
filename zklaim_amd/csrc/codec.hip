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
#include <cstring>
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

template <class A, class K>
static int decompress(K kernel, const uint8_t *host_rec, size_t stride, size_t off, size_t n, std::vector<uint64_t> &out, size_t limbs, DevBuf &d_rec, DevBuf &d_out, DevBuf &d_flag) {
    out.assign(n * limbs, 0);
    if (!n) return ZKG_OK;
    if (d_rec.reserve(n * stride) || d_out.reserve(n * sizeof(A)) || d_flag.reserve(4)) return ZKG_ERROR;
    ZK_HIP(hipMemcpy(d_rec.p, host_rec, n * stride, hipMemcpyHostToDevice));
    ZK_HIP(hipMemset(d_flag.p, 0, 4));
    hipLaunchKernelGGL(kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_rec.as<uint8_t>(), stride, off, n, d_out.as<A>(), d_flag.as<uint32_t>());
    uint32_t flag = 0;
    ZK_HIP(hipMemcpy(&flag, d_flag.p, 4, hipMemcpyDeviceToHost));
    if (flag) { set_error("pk blob: a compressed point is not on the curve"); return ZKG_ERROR; }
    ZK_HIP(hipMemcpy(out.data(), d_out.p, n * sizeof(A), hipMemcpyDeviceToHost));
    return ZKG_OK;
}

}  // namespace zk

using namespace zk;

static zkg_crs *crs_upload_blob_impl(const void *blob, size_t len) {
    if (!blob || len < 34 * 3 + 66 * 2) { set_error("zkg_crs_upload_blob: blob too short"); return nullptr; }
    Reader rd{(const uint8_t *)blob, (const uint8_t *)blob + len};
    DevBuf d_rec, d_out, d_flag;
    auto fail = [&](const char *msg) -> zkg_crs * { if (msg) set_error(msg); d_rec.release(); d_out.release(); d_flag.release(); return nullptr; };
    // fixed head: alpha_g1 beta_g1 beta_g2 delta_g1 delta_g2
    const uint8_t *alpha = rd.take(34), *beta1 = rd.take(34), *beta2 = rd.take(66), *delta1 = rd.take(34), *delta2 = rd.take(66);
    if (!rd.ok) return fail("pk blob: truncated head");
    std::vector<uint64_t> small1, small2, tmp;
    uint8_t head1[3 * 34], head2[2 * 66];
    memcpy(head1, alpha, 34); memcpy(head1 + 34, beta1, 34); memcpy(head1 + 68, delta1, 34);
    memcpy(head2, beta2, 66); memcpy(head2 + 66, delta2, 66);
    if (decompress<G1Affine>(k_decompress_g1, head1, 34, 0, 3, small1, 8, d_rec, d_out, d_flag) ||
        decompress<G2Affine>(k_decompress_g2, head2, 66, 0, 2, small2, 16, d_rec, d_out, d_flag)) return fail(nullptr);
    // A_query
    size_t nA = rd.count(34); const uint8_t *recA = rd.take_records(nA, 34);
    if (!rd.ok || nA == 0) return fail("pk blob: bad A_query");
    std::vector<uint64_t> A_query, H_query, L_query, Bv_g2, Bv_g1;
    if (decompress<G1Affine>(k_decompress_g1, recA, 34, 0, nA, A_query, 8, d_rec, d_out, d_flag)) return fail(nullptr);
    // B_query (sparse knowledge commitments: G2 then G1 per value)
    size_t domain = rd.count(0), nidx = rd.count(2);                      // an index is at least one digit and a newline
    if (!rd.ok || domain != nA || nidx > domain) return fail("pk blob: bad B_query header");
    std::vector<size_t> idx(nidx);
    for (size_t i = 0; i < nidx; ++i) { idx[i] = rd.decimal(); if (!rd.ok || idx[i] >= domain) return fail("pk blob: bad B_query index"); }
    size_t nval = rd.count(100); const uint8_t *recB = rd.take_records(nval, 100);
    if (!rd.ok || nval != nidx) return fail("pk blob: bad B_query values");
    if (decompress<G2Affine>(k_decompress_g2, recB, 100, 0, nval, Bv_g2, 16, d_rec, d_out, d_flag) ||
        decompress<G1Affine>(k_decompress_g1, recB, 100, 66, nval, Bv_g1, 8, d_rec, d_out, d_flag)) return fail(nullptr);
    std::vector<uint64_t> B_g1(domain * 8, 0), B_g2(domain * 16, 0);                    // dense, absent = infinity
    for (size_t i = 0; i < nidx; ++i) { memcpy(&B_g1[idx[i] * 8], &Bv_g1[i * 8], 64); memcpy(&B_g2[idx[i] * 16], &Bv_g2[i * 16], 128); }
    // H_query, L_query
    size_t nH = rd.count(34); const uint8_t *recH = rd.take_records(nH, 34);
    size_t nL = rd.ok ? rd.count(34) : 0; const uint8_t *recL = rd.take_records(nL, 34);
    if (!rd.ok) return fail("pk blob: bad H/L query");
    if (decompress<G1Affine>(k_decompress_g1, recH, 34, 0, nH, H_query, 8, d_rec, d_out, d_flag) ||
        decompress<G1Affine>(k_decompress_g1, recL, 34, 0, nL, L_query, 8, d_rec, d_out, d_flag)) return fail(nullptr);
    // constraint system
    size_t primary = rd.count(0), auxiliary = rd.count(0), ncons = rd.count(6);       // a constraint is at least three "0\n" term counts
    if (!rd.ok || primary + auxiliary + 1 != nA || nL != auxiliary) return fail("pk blob: constraint system sizes disagree with the queries");
    std::vector<uint32_t> rp[3], col[3]; std::vector<uint64_t> val[3];
    for (int m = 0; m < 3; ++m) { rp[m].reserve(ncons + 1); rp[m].push_back(0); }
    for (size_t c = 0; c < ncons; ++c)
        for (int m = 0; m < 3; ++m) {
            size_t nt = rd.count(34);                                       // a term: index digits, newline, 32-byte coefficient
            for (size_t t = 0; t < nt && rd.ok; ++t) {
                size_t index = rd.decimal(); const uint8_t *coeff = rd.take(32);
                if (!rd.ok || index >= nA) return fail("pk blob: bad linear term");
                col[m].push_back((uint32_t)index);
                size_t at = val[m].size(); val[m].resize(at + 4); memcpy(&val[m][at], coeff, 32);
            }
            if (!rd.ok) return fail("pk blob: truncated constraint");
            rp[m].push_back((uint32_t)col[m].size());
        }
    d_rec.release(); d_out.release(); d_flag.release();
    size_t m_dom = nH + 1;
    DomainShape shape;                                     // the domain size is not stored in the blob: H_query has m - 1 entries
    if (!domain_shape_of(m_dom, shape)) { set_error("pk blob: H_query length + 1 is neither a power of two nor a step_radix2 size 2^a + 2^b"); return nullptr; }
    zkg_pk pk; memset(&pk, 0, sizeof(pk));
    pk.cs.num_variables = (uint32_t)(nA - 1); pk.cs.num_inputs = (uint32_t)primary; pk.cs.num_constraints = (uint32_t)ncons;
    pk.cs.a_rowptr = rp[0].data(); pk.cs.a_col = col[0].data(); pk.cs.a_val = val[0].data();
    pk.cs.b_rowptr = rp[1].data(); pk.cs.b_col = col[1].data(); pk.cs.b_val = val[1].data();
    pk.cs.c_rowptr = rp[2].data(); pk.cs.c_col = col[2].data(); pk.cs.c_val = val[2].data();
    pk.log_m = shape.log_m; pk.domain_size = (uint32_t)shape.m;
    pk.alpha_g1 = small1.data(); pk.beta_g1 = small1.data() + 8; pk.delta_g1 = small1.data() + 16;
    pk.beta_g2 = small2.data(); pk.delta_g2 = small2.data() + 16;
    pk.A_query = A_query.data(); pk.B_g1 = B_g1.data(); pk.B_g2 = B_g2.data(); pk.H_query = H_query.data(); pk.L_query = L_query.data();
    return zkg_crs_upload(&pk);
}

// Nothing may propagate through the C boundary: allocation failures on hostile sizes end up here as an error return.
extern "C" zkg_crs *zkg_crs_upload_blob(const void *blob, size_t len) {
    try { return crs_upload_blob_impl(blob, len); }
    catch (const std::exception &e) { set_error(std::string("zkg_crs_upload_blob: ") + e.what()); return nullptr; }
    catch (...) { set_error("zkg_crs_upload_blob: unexpected exception"); return nullptr; }
}
