// setup_verify.hip — Groth16 key generation and verification for the zklaim seam.
//
// Replaces r1cs_gg_ppzksnark_generator (/root/reference/zklaim/snark.cpp:91, reached from libsnark_trusted_setup,
// zklaim/libsnark_wrapper.cpp:195-215) and r1cs_gg_ppzksnark_verifier_strong_IC (snark.cpp:62, reached from libsnark_verify,
// libsnark_wrapper.cpp:252-276), plus the vk / pk blob export (libsnark_wrapper.cpp:122-157).
//
// Generator: swap A and B when B touches more variables (libsnark's swap_AB_if_beneficial: fewer G2 bases), evaluate the QAP
// at the trapdoor point t on the host (Lagrange coefficients in closed form with one batched inversion, then one pass over
// the non-zeros), and turn the ~4n + m scalars into curve points with the fixed-base kernels of msm.hip on the GPU — the part
// that dominates the reference's setup time.  Domain: libfqfft's get_evaluation_domain(C + l + 1) rule
// (evaluation_domain_shape, ntt.hip): basic_radix2_domain or step_radix2_domain, Lagrange coefficients from domain_lagrange.
// Verifier: host pairing (host/pairing.hpp); the public input is folded into gamma_ABC with host scalar multiplications.
#include "common.hpp"
#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include "../../include/zkg.h"
#include "host/serialize.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace zk;
using zk::pairing::Fq12;

struct zkg_keypair {
    // the (possibly swapped) constraint system stored in the pk
    std::vector<uint32_t> rp[3], col[3]; std::vector<uint64_t> val[3];
    uint32_t n = 0, l = 0, C = 0, log_m = 0; size_t m = 0; bool swapped = false;
    ser::Bytes pk_blob; std::mutex blob_mu;            // serialised once, on first request
    G1Affine alpha_g1, beta_g1, delta_g1; G2Affine beta_g2, delta_g2, gamma_g2;
    std::vector<G1Affine> A_query, B_g1, H_query, L_query, IC;
    std::vector<G2Affine> B_g2;
    Fq12 alpha_beta;
    zkg_pk pk_view;
    // the seam's generator (seam_keygen) leaves the five queries on the DEVICE instead (they become the resident key of the proofs that
    // follow): the host vectors above stay empty, pk_view's query pointers are device pointers, b_idx lists the B query's non-zero entries
    bool on_device = false; DevBuf dA, dB1, dB2, dH, dL; std::vector<uint32_t> b_idx;
    ~zkg_keypair() { for (DevBuf *b : {&dA, &dB1, &dB2, &dH, &dL}) b->release(); }
};

namespace {

Fr fr_from_canonical(const uint64_t *limbs) { Fr x; memcpy(x.v, limbs, 32); return x.to_mont(); }

Fr random_fr() {
    std::random_device rd;                                   // the reference draws its toxic waste from std::random_device too
    for (;;) {
        Fr x;
        for (int i = 0; i < 8; ++i) x.v[i] = rd();
        x.v[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; --i) { if (x.v[i] != FrParams::P[i]) { lt = x.v[i] < FrParams::P[i]; break; } }
        if (lt && !x.is_zero()) return x;                    // uniform in [1, r); reading it as a Montgomery residue keeps it uniform
    }
}

void copy_csr(std::vector<uint32_t> &rp, std::vector<uint32_t> &col, std::vector<uint64_t> &val, const uint32_t *s_rp, const uint32_t *s_col, const uint64_t *s_val, uint32_t rows) {
    rp.assign(s_rp, s_rp + rows + 1);
    size_t nnz = s_rp[rows];
    col.assign(s_col, s_col + nnz);
    val.assign(s_val, s_val + 4 * nnz);
}

template <class A, class FN>
int batch_points(FN fixed_base_fn, const A &base, const std::vector<Fr> &scalars, std::vector<A> &out) {
    size_t n = scalars.size();
    out.resize(n);
    if (!n) return ZKG_OK;
    static_assert(sizeof(Fr) == 32, "Fr is the 32-byte Montgomery element the kernel reads");
    DevBuf d_s, d_o;
    if (d_s.reserve(n * 32) || d_o.reserve(n * sizeof(A))) return ZKG_ERROR;
    int rc = ZKG_ERROR;                                    // Montgomery scalars go up as they are: the kernel converts (no host pass)
    if (hip_ok(hipMemcpy(d_s.p, scalars.data(), n * 32, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__) &&
        fixed_base_fn(base, d_s.as<uint32_t>(), n, d_o.as<A>(), nullptr, true) == ZKG_OK &&
        hip_ok(hipMemcpy(out.data(), d_o.p, n * sizeof(A), hipMemcpyDeviceToHost), "D2H", __FILE__, __LINE__)) rc = ZKG_OK;
    d_s.release(); d_o.release();
    return rc;
}

// the same batch with its result left on the device
template <class A, class FN>
int batch_points_dev(FN fixed_base_fn, const A &base, const std::vector<Fr> &scalars, DevBuf &d_out) {
    const size_t n = scalars.size();
    if (d_out.reserve(n * sizeof(A) + 16)) return ZKG_ERROR;
    if (!n) return ZKG_OK;
    ScopedDevBuf d_s;
    if (d_s.reserve(n * 32) || !hip_ok(hipMemcpy(d_s.p, scalars.data(), n * 32, hipMemcpyHostToDevice), "H2D", __FILE__, __LINE__)) return ZKG_ERROR;
    return fixed_base_fn(base, d_s.as<uint32_t>(), n, d_out.as<A>(), nullptr, true);
}

G1Affine g1_generator() { return {Fq::from_u64(1), Fq::from_u64(2)}; }
G2Affine g2_generator() {
    auto limbs = [](std::initializer_list<uint32_t> l) { Fq x; int i = 0; for (uint32_t v : l) x.v[i++] = v; return x; };
    return {{limbs({0x02bc2026u, 0x8e83b5d1u, 0x497b0172u, 0xdceb1935u, 0x97811adfu, 0xfbb82647u, 0xaf96503bu, 0x19573841u}),
             limbs({0xa84c6140u, 0xafb4737du, 0x5802d8c4u, 0x6043dd5au, 0x52a02f86u, 0x09e950fcu, 0x3aea7b6bu, 0x14fef083u})},
            {limbs({0x886be9f6u, 0x619dfa9du, 0xf59e9b78u, 0xfe7fd297u, 0x231b7dfeu, 0xff9e1a62u, 0xae9e4206u, 0x28fd7eebu}),
             limbs({0xc71856eeu, 0x64095b56u, 0x327d3cbbu, 0xdc57f922u, 0x33351076u, 0x55f935beu, 0x93fd6482u, 0x0da4a0e6u})}};
}
void fr_limbs(const Fr &x, uint32_t out[8]) { Fr c = x.from_mont(); memcpy(out, c.v, 32); }

}  // namespace

extern "C" {

// The generator proper.  The constraint system comes either as the ABI's view (copied) or — `owned` — as CSR vectors the caller gives up
// (the seam's circuit: 100 MB at 20 payloads that would otherwise be copied and then freed twice); `under_gpu` runs on a thread of its own
// while the GPU turns the scalars into points (the seam destroys its circuit there).
// keep_on_device: the seam's variant (seam_keygen below).  after_csr(kp) runs when the generator's host loops are done and its GPU phase
// begins (kp holds the constraint system it will store) — the seam starts writing the pk blob's constraint rows there; before_delete()
// runs before a failing generator deletes kp (whatever after_csr started must have let go of it).
static zkg_keypair *groth16_setup_impl(const zkg_r1cs *cs, const uint64_t *trapdoor, OwnedCsr *owned, const std::function<void()> &under_gpu, bool keep_on_device = false,
                                       const std::function<void(zkg_keypair *)> &after_csr = nullptr, const std::function<void()> &before_delete = nullptr) {
    if (!cs || (!owned && (!cs->a_rowptr || !cs->b_rowptr || !cs->c_rowptr))) { set_error("zkg_groth16_setup: null constraint system"); return nullptr; }
    zkg_keypair *kp = new zkg_keypair();
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) fprintf(stderr, "[zkg setup] %-28s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    kp->n = cs->num_variables; kp->l = cs->num_inputs; kp->C = cs->num_constraints;
    const size_t n = kp->n, l = kp->l, C = kp->C;
    if (owned) {
        for (int k = 0; k < 3; ++k) { kp->rp[k].swap(owned->rp[k]); kp->col[k].swap(owned->col[k]); kp->val[k].swap(owned->val[k]); }
        for (int k = 0; k < 3; ++k)
            if (kp->rp[k].size() != C + 1 || kp->col[k].size() != kp->rp[k][C] || kp->val[k].size() != 4 * kp->col[k].size()) { set_error("zkg_groth16_setup: inconsistent CSR"); delete kp; return nullptr; }
    } else {
        copy_csr(kp->rp[0], kp->col[0], kp->val[0], cs->a_rowptr, cs->a_col, cs->a_val, kp->C);
        copy_csr(kp->rp[1], kp->col[1], kp->val[1], cs->b_rowptr, cs->b_col, cs->b_val, kp->C);
        copy_csr(kp->rp[2], kp->col[2], kp->val[2], cs->c_rowptr, cs->c_col, cs->c_val, kp->C);
    }
    {   // swap_AB_if_beneficial: count the variables each of A and B touches
        std::vector<char> ta(n + 1, 0), tb(n + 1, 0);
        for (uint32_t c : kp->col[0]) ta[c] = 1;
        for (uint32_t c : kp->col[1]) tb[c] = 1;
        size_t na = 0, nb = 0;
        for (size_t i = 0; i <= n; ++i) { na += ta[i]; nb += tb[i]; }
        if (nb > na) { kp->rp[0].swap(kp->rp[1]); kp->col[0].swap(kp->col[1]); kp->val[0].swap(kp->val[1]); kp->swapped = true; }
    }
    auto drop = [&]() -> zkg_keypair * { if (before_delete) before_delete(); delete kp; return nullptr; };
    DomainShape shape;                                                     // libfqfft get_evaluation_domain(C + l + 1)
    if (!evaluation_domain_shape(C + l + 1, shape)) { set_error("zkg_groth16_setup: system too large for the 2-adicity of Fr"); return drop(); }
    const unsigned log_m = shape.log_m;
    kp->log_m = log_m; kp->m = shape.m;
    const size_t m = shape.m;
    Fr t, alpha, beta, gamma, delta;
    if (trapdoor) { t = fr_from_canonical(trapdoor); alpha = fr_from_canonical(trapdoor + 4); beta = fr_from_canonical(trapdoor + 8); gamma = fr_from_canonical(trapdoor + 12); delta = fr_from_canonical(trapdoor + 16); }
    else { t = random_fr(); alpha = random_fr(); beta = random_fr(); gamma = random_fr(); delta = random_fr(); }
    lap("copy + swap");
    // ---- Lagrange coefficients u_i = L_i(t) and Z(t) on the chosen domain (closed forms, one batched inversion)
    Fr Zt; std::vector<Fr> u;
    if (domain_lagrange(shape, t, u, Zt)) return drop();
    lap("lagrange");
    // ---- QAP polynomials at t (r1cs_to_qap_instance_map_with_evaluation)
    std::vector<Fr> At(n + 1, Fr::zero()), Bt(n + 1, Fr::zero()), Ct(n + 1, Fr::zero());
    for (size_t i = 0; i <= l; ++i) At[i] = u[C + i];
    std::vector<Fr> *dst[3] = {&At, &Bt, &Ct};
    host_parallel_for(3, [&](int k) {                                       // the three matrices accumulate into separate vectors
        const Fr one = Fr::one(), minus_one = Fr::one().neg();               // a gadget circuit's coefficients are mostly +-1: no product needed
        for (size_t i = 0; i < C; ++i)
            for (uint32_t e = kp->rp[k][i]; e < kp->rp[k][i + 1]; ++e) {
                Fr c; memcpy(c.v, &kp->val[k][4 * (size_t)e], 32);
                Fr &acc = (*dst[k])[kp->col[k][e]];
                if (c == one) acc += u[i]; else if (c == minus_one) acc -= u[i]; else acc += u[i] * c;
            }
    });
    Fr dinv = delta.inverse(), ginv = gamma.inverse();
    std::vector<Fr> Hs(m - 1), Ls(n - l), ICs(l + 1);
    auto chunked = [&](size_t count, const std::function<void(size_t, size_t)> &f) {           // [0, count) in chunks on the host pool
        const int chunks = (int)std::min<size_t>(64, (count + 8191) / 8192);
        if (chunks <= 1) { f(0, count); return; }
        host_parallel_for(chunks, [&](int c) { f(count * (size_t)c / chunks, count * (size_t)(c + 1) / chunks); });
    };
    {
        const Fr zd = Zt * dinv;
        chunked(m - 1, [&](size_t lo, size_t hi) { Fr ti = t.pow_u64(lo) * zd; for (size_t i = lo; i < hi; ++i) { Hs[i] = ti; ti = ti * t; } });   // t^i Z(t) / delta
    }
    chunked(n + 1, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            Fr abc = beta * At[i] + alpha * Bt[i] + Ct[i];
            if (i <= l) ICs[i] = abc * ginv; else Ls[i - l - 1] = abc * dinv;
        }
    });
    lap("qap evaluation + scalars");
    // ---- scalars -> points (GPU fixed-base batches).  The host pool is free from here on: what after_csr starts (the seam: the pk blob's
    //      constraint rows) shares it with nothing but the short page pre-faulting below (started earlier it made the Lagrange and QAP
    //      loops above run inline on this thread: 37 -> 90 ms)
    if (after_csr) after_csr(kp);
    std::thread side;
    if (under_gpu) side = std::thread([&] { try { under_gpu(); } catch (...) {} });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_side{side};
    G1Affine g1 = g1_generator(); G2Affine g2 = g2_generator();
    std::vector<G1Affine> small1; std::vector<G2Affine> small2;
    // the result vectors (250 MB at 20 payloads) get their pages on the pool, side by side, instead of one after the other inside the batches
    static const bool no_prefault = getenv("ZKG_NO_PREFAULT") != nullptr;
    if (keep_on_device) {
        // the seam: ~4n + m points computed into device buffers and left there; only the 8 key elements and the l + 1 points of the
        // verification key come back
        kp->on_device = true;
        for (size_t i = 0; i < Bt.size(); ++i) if (!Bt[i].is_zero()) kp->b_idx.push_back((uint32_t)i);       // zero scalar <=> point at infinity
        const bool okd = batch_points<G1Affine>(fixed_base_g1, g1, {alpha, beta, delta}, small1) == 0 && batch_points<G2Affine>(fixed_base_g2, g2, {beta, delta, gamma}, small2) == 0 &&
                         batch_points_dev<G1Affine>(fixed_base_g1, g1, At, kp->dA) == 0 && batch_points_dev<G1Affine>(fixed_base_g1, g1, Bt, kp->dB1) == 0 &&
                         batch_points_dev<G2Affine>(fixed_base_g2, g2, Bt, kp->dB2) == 0 && batch_points_dev<G1Affine>(fixed_base_g1, g1, Hs, kp->dH) == 0 &&
                         batch_points_dev<G1Affine>(fixed_base_g1, g1, Ls, kp->dL) == 0 && batch_points<G1Affine>(fixed_base_g1, g1, ICs, kp->IC) == 0;
        if (!okd) return drop();
    } else {
    if (!no_prefault) host_parallel_for(5, [&](int i) {
        if (i == 0) kp->B_g2.resize(Bt.size()); else if (i == 1) kp->A_query.resize(At.size()); else if (i == 2) kp->B_g1.resize(Bt.size());
        else if (i == 3) kp->H_query.resize(Hs.size()); else kp->L_query.resize(Ls.size());
    });
    bool ok = batch_points<G1Affine>(fixed_base_g1, g1, {alpha, beta, delta}, small1) == 0 && batch_points<G2Affine>(fixed_base_g2, g2, {beta, delta, gamma}, small2) == 0 &&
              batch_points<G1Affine>(fixed_base_g1, g1, At, kp->A_query) == 0 && batch_points<G1Affine>(fixed_base_g1, g1, Bt, kp->B_g1) == 0 &&
              batch_points<G2Affine>(fixed_base_g2, g2, Bt, kp->B_g2) == 0 && batch_points<G1Affine>(fixed_base_g1, g1, Hs, kp->H_query) == 0 &&
              batch_points<G1Affine>(fixed_base_g1, g1, Ls, kp->L_query) == 0 && batch_points<G1Affine>(fixed_base_g1, g1, ICs, kp->IC) == 0;
    if (!ok) return drop();
    }
    kp->alpha_g1 = small1[0]; kp->beta_g1 = small1[1]; kp->delta_g1 = small1[2];
    kp->beta_g2 = small2[0]; kp->delta_g2 = small2[1]; kp->gamma_g2 = small2[2];
    lap("fixed-base batches (GPU)");
    kp->alpha_beta = pairing::reduced_pairing(kp->alpha_g1, kp->beta_g2);
    lap("pairing");
    zkg_pk &v = kp->pk_view; memset(&v, 0, sizeof(v));
    v.cs.num_variables = kp->n; v.cs.num_inputs = kp->l; v.cs.num_constraints = kp->C;
    v.cs.a_rowptr = kp->rp[0].data(); v.cs.a_col = kp->col[0].data(); v.cs.a_val = kp->val[0].data();
    v.cs.b_rowptr = kp->rp[1].data(); v.cs.b_col = kp->col[1].data(); v.cs.b_val = kp->val[1].data();
    v.cs.c_rowptr = kp->rp[2].data(); v.cs.c_col = kp->col[2].data(); v.cs.c_val = kp->val[2].data();
    v.log_m = log_m; v.domain_size = (uint32_t)m;
    v.alpha_g1 = (const uint64_t *)&kp->alpha_g1; v.beta_g1 = (const uint64_t *)&kp->beta_g1; v.delta_g1 = (const uint64_t *)&kp->delta_g1;
    v.beta_g2 = (const uint64_t *)&kp->beta_g2; v.delta_g2 = (const uint64_t *)&kp->delta_g2;
    v.A_query = (const uint64_t *)kp->A_query.data(); v.B_g1 = (const uint64_t *)kp->B_g1.data(); v.B_g2 = (const uint64_t *)kp->B_g2.data();
    v.H_query = (const uint64_t *)kp->H_query.data(); v.L_query = (const uint64_t *)kp->L_query.data();
    if (kp->on_device) {                                                         // DEVICE pointers: for crs_upload_device_queries only
        v.A_query = kp->dA.as<uint64_t>(); v.B_g1 = kp->dB1.as<uint64_t>(); v.B_g2 = kp->dB2.as<uint64_t>(); v.H_query = kp->dH.as<uint64_t>(); v.L_query = kp->dL.as<uint64_t>();
    }
    return kp;
}
zkg_keypair *zkg_groth16_setup(const zkg_r1cs *cs, const uint64_t *trapdoor /* 5 x 4 canonical limbs: t, alpha, beta, gamma, delta; NULL = random */) {
    return groth16_setup_impl(cs, trapdoor, nullptr, nullptr);
}

void zkg_keypair_free(zkg_keypair *kp) { delete kp; }
const zkg_pk *zkg_keypair_pk(const zkg_keypair *kp) { return kp && !kp->on_device ? &kp->pk_view : nullptr; }
int zkg_keypair_swapped(const zkg_keypair *kp) { return kp && kp->swapped ? 1 : 0; }

// operator<<(r1cs_gg_ppzksnark_proving_key), layout in codec.hip.  Built once per keypair (callers ask for the size first, then for
// the bytes); the fixed-size point records — 2.2 M of them at 20 payloads — are serialised on the host pool.
// the constraint rows of a pk blob (per row: a, b, c as #terms '\n' (index '\n' coefficient)*), in chunks written on the host pool
static void constraint_rows_text(const zkg_keypair *kp, std::vector<ser::Writer> &part) {
    const int chunks = (int)std::min<size_t>(64, ((size_t)kp->C + 4095) / 4096);
    part.assign(std::max(chunks, 1), ser::Writer());
    auto rows = [&](int ch) {
        ser::Writer &pw = part[ch];
        const uint32_t lo = (uint32_t)((size_t)kp->C * ch / std::max(chunks, 1)), hi = (uint32_t)((size_t)kp->C * (ch + 1) / std::max(chunks, 1));
        size_t terms = 0;
        for (int k = 0; k < 3; ++k) terms += kp->rp[k][hi] - kp->rp[k][lo];
        pw.buf.reserve(terms * 40 + (size_t)(hi - lo) * 8 + 64);
        for (uint32_t c = lo; c < hi; ++c)
            for (int k = 0; k < 3; ++k) {
                pw.dec(kp->rp[k][c + 1] - kp->rp[k][c]);
                for (uint32_t e = kp->rp[k][c]; e < kp->rp[k][c + 1]; ++e) { pw.dec(kp->col[k][e]); pw.raw(&kp->val[k][4 * (size_t)e], 32); }
            }
    };
    if (chunks <= 1) rows(0); else host_parallel_for(chunks, rows);
}
static void build_pk_blob(const zkg_keypair *kp, ser::Bytes &buf) {
    ser::Writer w;
    std::vector<size_t> idx;
    for (size_t i = 0; i < kp->B_g2.size(); ++i) if (!kp->B_g2[i].is_inf() || !kp->B_g1[i].is_inf()) idx.push_back(i);
    const size_t nterms = kp->col[0].size() + kp->col[1].size() + kp->col[2].size();
    w.buf.reserve((kp->A_query.size() + kp->H_query.size() + kp->L_query.size()) * 34 + idx.size() * 108 + nterms * 40 + (size_t)kp->C * 8 + 4096);
    // a run of fixed-size records: reserve the bytes, fill them in parallel
    auto records = [&](size_t count, size_t rec, const std::function<void(size_t, uint8_t *)> &put) {
        const size_t at = w.buf.size();
        w.buf.resize(at + count * rec);
        uint8_t *base = w.buf.data() + at;
        const int chunks = (int)std::min<size_t>(64, (count + 4095) / 4096);
        host_parallel_for(chunks, [&](int c) {
            size_t lo = count * (size_t)c / chunks, hi = count * (size_t)(c + 1) / chunks;
            for (size_t i = lo; i < hi; ++i) put(i, base + i * rec);
        });
    };
    w.g1(kp->alpha_g1); w.g1(kp->beta_g1); w.g2(kp->beta_g2); w.g1(kp->delta_g1); w.g2(kp->delta_g2);
    w.dec(kp->A_query.size()); records(kp->A_query.size(), 34, [&](size_t i, uint8_t *o) { ser::put_g1(o, kp->A_query[i]); });
    w.dec(kp->B_g2.size()); w.dec(idx.size()); for (size_t i : idx) w.dec(i);
    w.dec(idx.size()); records(idx.size(), 100, [&](size_t j, uint8_t *o) { ser::put_g2(o, kp->B_g2[idx[j]]); ser::put_g1(o + 66, kp->B_g1[idx[j]]); });
    w.dec(kp->H_query.size()); records(kp->H_query.size(), 34, [&](size_t i, uint8_t *o) { ser::put_g1(o, kp->H_query[i]); });
    w.dec(kp->L_query.size()); records(kp->L_query.size(), 34, [&](size_t i, uint8_t *o) { ser::put_g1(o, kp->L_query[i]); });
    w.dec(kp->l); w.dec(kp->n - kp->l); w.dec(kp->C);
    {   // the constraint system: variable-length records (decimal counts and indices), so each chunk of rows is written to a buffer of its
        // own on the pool and the buffers are then copied into place, also in parallel
        std::vector<ser::Writer> part;
        constraint_rows_text(kp, part);
        std::vector<size_t> at(part.size() + 1, w.buf.size());
        for (size_t i = 0; i < part.size(); ++i) at[i + 1] = at[i] + part[i].buf.size();
        w.buf.resize(at.back());
        host_parallel_for((int)part.size(), [&](int i) { if (!part[i].buf.empty()) memcpy(w.buf.data() + at[i], part[i].buf.data(), part[i].buf.size()); });
    }
    buf.swap(w.buf);
}
size_t zkg_keypair_pk_blob(const zkg_keypair *kp_, uint8_t *out, size_t cap) {
    zkg_keypair *kp = const_cast<zkg_keypair *>(kp_);
    if (!kp || kp->on_device) return 0;
    std::lock_guard<std::mutex> lk(kp->blob_mu);
    if (kp->pk_blob.empty()) build_pk_blob(kp, kp->pk_blob);
    if (out && cap >= kp->pk_blob.size()) {                                   // hundreds of MB into fresh pages: copy in parallel pieces
        const size_t len = kp->pk_blob.size(); const int chunks = (int)std::min<size_t>(32, (len >> 22) + 1);
        host_parallel_for(chunks, [&](int c) { size_t lo = len * (size_t)c / chunks, hi = len * (size_t)(c + 1) / chunks; memcpy(out + lo, kp->pk_blob.data() + lo, hi - lo); });
    }
    return kp->pk_blob.size();
}

// operator<<(r1cs_gg_ppzksnark_verification_key): alpha_g1_beta_g2 (GT, 384 B) | gamma_g2 | delta_g2 | gamma_ABC_g1 as an
// accumulation_vector: first (G1) then a sparse vector: domain '\n' #indices '\n' (index '\n')* #values '\n' (G1)*
size_t zkg_keypair_vk_blob(const zkg_keypair *kp, uint8_t *out, size_t cap) {
    if (!kp) return 0;
    ser::Writer w;
    uint8_t gt[384]; ser::put_fq12(gt, kp->alpha_beta); w.raw(gt, 384);
    w.g2(kp->gamma_g2); w.g2(kp->delta_g2);
    w.g1(kp->IC[0]);
    size_t rest = kp->IC.size() - 1;
    w.dec(rest); w.dec(rest); for (size_t i = 0; i < rest; ++i) w.dec(i);
    w.dec(rest); for (size_t i = 0; i < rest; ++i) w.g1(kp->IC[i + 1]);
    if (out && cap >= w.buf.size()) memcpy(out, w.buf.data(), w.buf.size());
    return w.buf.size();
}

// r1cs_gg_ppzksnark_verifier_strong_IC: 0 = proof valid, 1 = invalid (libsnark_verify returns !valid, libsnark_wrapper.cpp:269),
// 2 = malformed key / proof.  primary_input: n_inputs x 4 limbs, Montgomery Fr.
// What a verification key contributes to every verification, computed once per key: its points decompressed (one square root each —
// l + 3 of them) and every line of the Miller loops of gamma_g2 and delta_g2 (libsnark's r1cs_gg_ppzksnark_processed_verification_key;
// the reference's libsnark_verify re-parses and re-processes ctx->vk on every call, libsnark_wrapper.cpp:252-276).  Kept per vk blob —
// found by a 64-bit digest, confirmed byte for byte — for the last few keys.
struct PreparedVk {
    std::vector<uint8_t> blob;
    Fq12 alpha_beta; G2Affine gamma_g2, delta_g2; G1Affine ic0; size_t domain = 0;
    std::vector<size_t> idx; std::vector<G1Affine> ic;                          // gamma_ABC: indices and decompressed values
    std::vector<pairing::LineCoeff> gamma_lines, delta_lines;                   // empty when the point is infinity
};
static int prepare_vk(const uint8_t *vk_blob, size_t vk_len, PreparedVk &v) {   // 0, or 2 = malformed (message set)
    ser::Reader rd{vk_blob, vk_blob + vk_len};
    const uint8_t *gt = rd.take(384), *pg = rd.take(66), *pd = rd.take(66), *p0 = rd.take(34);
    if (!rd.ok) { set_error("vk blob truncated"); return 2; }
    ser::get_fq12(gt, v.alpha_beta);
    if (!ser::get_g2(pg, v.gamma_g2) || !ser::get_g2(pd, v.delta_g2) || !ser::get_g1(p0, v.ic0)) { set_error("vk blob: bad point"); return 2; }
    size_t domain = rd.dec(), nidx = rd.dec();
    // counts are bounded by the bytes that can still follow (an index takes >= 2 bytes, a value 34) before anything is sized by them
    if (!rd.ok || nidx > domain || nidx > (size_t)(rd.end - rd.p) / 2) { set_error("vk blob: bad gamma_ABC header"); return 2; }
    v.domain = domain; v.idx.resize(nidx);
    for (auto &i : v.idx) { i = rd.dec(); if (!rd.ok || i >= domain) { set_error("vk blob: bad index"); return 2; } }
    size_t nval = rd.dec();
    if (!rd.ok || nval != nidx || nval > (size_t)(rd.end - rd.p) / 34) { set_error("vk blob: bad gamma_ABC values"); return 2; }
    const uint8_t *vals = rd.take(nval * 34);
    if (!rd.ok) { set_error("vk blob: truncated gamma_ABC values"); return 2; }
    v.ic.resize(nidx);
    std::vector<char> bad(nidx, 0);
    host_parallel_for((int)nidx, [&](int k) { if (!ser::get_g1(vals + 34 * (size_t)k, v.ic[k])) bad[k] = 1; });
    for (char b : bad) if (b) { set_error("vk blob: bad gamma_ABC point"); return 2; }
    if (!v.gamma_g2.is_inf()) v.gamma_lines = pairing::miller_lines(v.gamma_g2);
    if (!v.delta_g2.is_inf()) v.delta_lines = pairing::miller_lines(v.delta_g2);
    v.blob.assign(vk_blob, vk_blob + vk_len);
    return 0;
}
static std::mutex g_vk_mu;
static std::map<uint64_t, std::shared_ptr<const PreparedVk>> g_vk_cache;
static uint64_t vk_digest(const uint8_t *p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    for (; i < n; ++i) { h = (h ^ p[i]) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    return h;
}
static std::shared_ptr<const PreparedVk> prepared_vk(const uint8_t *vk_blob, size_t vk_len, int &rc) {
    const uint64_t key = vk_digest(vk_blob, vk_len);
    {
        std::lock_guard<std::mutex> lk(g_vk_mu);
        auto it = g_vk_cache.find(key);
        if (it != g_vk_cache.end() && it->second->blob.size() == vk_len && memcmp(it->second->blob.data(), vk_blob, vk_len) == 0) { rc = 0; return it->second; }
    }
    auto v = std::make_shared<PreparedVk>();
    rc = prepare_vk(vk_blob, vk_len, *v);
    if (rc) return nullptr;
    std::lock_guard<std::mutex> lk(g_vk_mu);
    if (g_vk_cache.size() >= 8) g_vk_cache.clear();
    g_vk_cache[key] = v;
    return v;
}

static int groth16_verify_impl(const uint8_t *vk_blob, size_t vk_len, const uint64_t *primary_input, size_t n_inputs, const uint8_t *proof, size_t proof_len) {
    if (!vk_blob || !proof || (n_inputs && !primary_input)) { set_error("zkg_groth16_verify: null argument"); return 2; }
    int rc = 0;
    const std::shared_ptr<const PreparedVk> vk = prepared_vk(vk_blob, vk_len, rc);
    if (!vk) return rc;
    const size_t nidx = vk->idx.size();
    if (vk->domain != n_inputs) return 1;                                       // strong input consistency: sizes must agree
    if (proof_len != ZKG_PROOF_BYTES) return 1;
    G1Affine pA, pC; G2Affine pB;
    if (!ser::get_g1(proof, pA) || !ser::get_g2(proof + 34, pB) || !ser::get_g1(proof + 100, pC)) return 1;     // is_well_formed
    // acc = IC_0 + sum_i input_i * IC_{i+1}
    G1 acc = G1::from_affine(vk->ic0);
    {   // sum_i input_i * IC_{i+1} in up to 16 chunks on the host pool.  A chunk shares its doublings (Straus, one bit at a time: 254 doublings
        // and on average 127 mixed additions per point instead of a double-and-add per point): 0.67 -> 0.3 ms at 20 payloads' 102 inputs.
        const int chunks = (int)std::min<size_t>(16, nidx);
        std::vector<G1> part(std::max(chunks, 1), G1::inf());
        host_parallel_for(chunks, [&](int c) {
            const size_t lo = nidx * (size_t)c / chunks, hi = nidx * (size_t)(c + 1) / chunks, cnt = hi - lo;
            std::vector<uint32_t> e(8 * cnt);
            for (size_t k = lo; k < hi; ++k) { Fr x; memcpy(x.v, primary_input + 4 * vk->idx[k], 32); fr_limbs(x, &e[8 * (k - lo)]); }
            const G1Affine *pt = vk->ic.data() + lo;
            G1 a = G1::inf();
            for (int bit = 255; bit >= 0; --bit) {
                a = a.dbl();
                for (size_t j = 0; j < cnt; ++j) if ((e[8 * j + (bit >> 5)] >> (bit & 31)) & 1u) a.madd(pt[j]);
            }
            part[c] = a;
        });
        for (int c = 0; c < chunks; ++c) acc.add(part[c]);
    }
    // e(A, B) == e(alpha, beta) * e(acc, gamma) * e(C, delta)   <=>   FE( ML(A,B) * ML(-acc, gamma) * ML(-C, delta) ) == alpha_beta
    // (three loops in lock-step; the lines of gamma and delta come from the prepared key, only B's are computed here)
    G1Affine accA = acc.to_affine();
    std::vector<G1Affine> Ps; std::vector<G2Affine> Qs; std::vector<const std::vector<pairing::LineCoeff> *> prep;
    auto ml = [&](const G1Affine &P, const G2Affine &Q, const std::vector<pairing::LineCoeff> *lines) { if (!P.is_inf() && !Q.is_inf()) { Ps.push_back(P); Qs.push_back(Q); prep.push_back(lines); } };
    ml(pA, pB, nullptr); ml(accA.neg(), vk->gamma_g2, &vk->gamma_lines); ml(pC.neg(), vk->delta_g2, &vk->delta_lines);
    Fq12 f = Ps.empty() ? Fq12::one() : pairing::multi_miller_loop(Ps.data(), Qs.data(), (int)Ps.size(), prep.data());
    return pairing::final_exponentiation(f) == vk->alpha_beta ? 0 : 1;
}

int zkg_groth16_verify(const uint8_t *vk_blob, size_t vk_len, const uint64_t *primary_input, size_t n_inputs, const uint8_t *proof, size_t proof_len) {
    try { return groth16_verify_impl(vk_blob, vk_len, primary_input, n_inputs, proof, proof_len); }       // nothing propagates through the C boundary
    catch (const std::exception &e) { set_error(std::string("zkg_groth16_verify: ") + e.what()); return 2; }
    catch (...) { set_error("zkg_groth16_verify: unexpected exception"); return 2; }
}

// bilinearity probe for the tests: writes e(a*G1, b*G2) (384 B) for canonical scalars a, b
int zkg_pairing_probe(const uint64_t a[4], const uint64_t b[4], uint8_t out[384]) {
    uint32_t ea[8], eb[8]; memcpy(ea, a, 32); memcpy(eb, b, 32);
    G1Affine P = G1::from_affine(g1_generator()).mul(ea, 8).to_affine();
    G2Affine Q = G2::from_affine(g2_generator()).mul(eb, 8).to_affine();
    ser::put_fq12(out, pairing::reduced_pairing(P, Q));
    return 0;
}

// test hook: 0 when (1) x -> x^(q^k) by coefficient maps equals square-and-multiply by q^k (k = 1, 2, 3) and (2) the last chunk of the
// final exponentiation equals square-and-multiply by the integer `e` (nlimbs x u32, little-endian) and (3) the projective and the
// affine Miller loops give the same reduced pairing products and (4) prepared lines give the same Miller value; bit flags otherwise
int zkg_pairing_selfcheck(const uint32_t *e, int nlimbs) {
    uint32_t k3[8] = {3}, k5[8] = {5};
    G1Affine P = G1::from_affine(g1_generator()).mul(k3, 8).to_affine();
    G2Affine Q = G2::from_affine(g2_generator()).mul(k5, 8).to_affine();
    Fq12 f = pairing::miller_loop(P, Q);
    int bad = 0;
    Fq12 x = f;
    for (int k = 1; k <= 3; ++k) { x = x.pow(FqParams::P, 8); if (!(x == pairing::frobenius(f, k))) bad |= 1 << (k - 1); }
    Fq12 g = pairing::final_exponentiation_first_chunk(f);
    if (!(g.conjugate() * g == Fq12::one())) bad |= 8;                        // in the cyclotomic subgroup: g^(q^6) = g^-1
    if (e && nlimbs > 0 && !(pairing::final_exponentiation_last_chunk(g) == g.pow(e, nlimbs))) bad |= 16;
    {   // the inversion-free lock-step Miller loop against the affine one of the definition: equal after the final exponentiation
        uint32_t k[6][8] = {{7}, {11}, {0x9e3779b9u, 0x7f4a7c15u, 0xf39cc060u, 5}, {13}, {0xdeadbeefu, 0x12345678u, 0xcafef00du, 0x0badc0deu, 0x31415926u, 0x27182818u, 0x16180339u, 0x1}, {17}};
        G1Affine Ps[3]; G2Affine Qs[3];
        for (int j = 0; j < 3; ++j) { Ps[j] = G1::from_affine(g1_generator()).mul(k[j], 8).to_affine(); Qs[j] = G2::from_affine(g2_generator()).mul(k[3 + j], 8).to_affine(); }
        for (int n = 1; n <= 3; ++n)
            if (!(pairing::final_exponentiation(pairing::multi_miller_loop(Ps, Qs, n)) == pairing::final_exponentiation(pairing::multi_miller_loop_affine(Ps, Qs, n)))) bad |= 32;
        // lines prepared ahead for two of the three pairs (a verification key's gamma and delta): the very same Miller value
        std::vector<pairing::LineCoeff> l1 = pairing::miller_lines(Qs[1]), l2 = pairing::miller_lines(Qs[2]);
        const std::vector<pairing::LineCoeff> *prep[3] = {nullptr, &l1, &l2};
        G2Affine unused[3] = {Qs[0], G2Affine::inf(), G2Affine::inf()};
        if (!(pairing::multi_miller_loop(Ps, unused, 3, prep) == pairing::multi_miller_loop(Ps, Qs, 3))) bad |= 64;
    }
    return bad;
}

}  // extern "C"

// the seam's entry (compat.hip): sizes in `cs`, the CSR arrays given up in `owned`
zkg_keypair *groth16_setup_owned(const zkg_r1cs *cs, zk::OwnedCsr *owned, const std::function<void()> &under_gpu) {
    return groth16_setup_impl(cs, nullptr, owned, under_gpu);
}

zkg_crs *crs_upload_device_queries(const zkg_pk *pk, const std::function<bool()> &constraint_system_ready);   // prover.hip

// a finished generator's keypair is destroyed on a thread of its own (one at a time: the next one, and seam_keygen_quiesce, wait for it)
// (the thread object lives on the heap and is never destroyed: a C caller that exits without zkg_shutdown must not meet the destructor of
//  a joinable std::thread; an atexit handler joins it before the runtime goes away)
static std::mutex &discard_mu() { static std::mutex *m = new std::mutex(); return *m; }
static std::thread &discard_thread() { static std::thread *t = new std::thread(); return *t; }
void seam_keygen_quiesce() {
    std::lock_guard<std::mutex> lk(discard_mu());
    if (discard_thread().joinable()) discard_thread().join();
}
static void keypair_discard(zkg_keypair *kp, int device) {
    static const bool registered = [] { return std::atexit(seam_keygen_quiesce) == 0; }();
    (void)registered;
    std::lock_guard<std::mutex> lk(discard_mu());
    if (discard_thread().joinable()) discard_thread().join();
    discard_thread() = std::thread([kp, device] { (void)hipSetDevice(device); delete kp; });
}

// libsnark_trusted_setup's generator (zklaim/libsnark_wrapper.cpp:195-215) as the seam runs it.  The reference's protocol is setup -> ONE
// prove -> verify per key (src/main_benchmark.c:113-148), so the key this call generates is the key the next libsnark_prove needs:
// the ~4n + m query points are computed into device buffers and stay there as the resident key (zkg_crs, H table included); the pk blob
// the C caller receives is assembled from GPU-compressed records (34 / 100 bytes per point instead of 64 / 192 over PCIe, no host point
// vectors at all) around the constraint rows' text, which the host pool writes meanwhile.  Same bytes as zkg_keypair_pk_blob writes.
// pk / vk: malloc'd (the caller's to free).  on_blob runs on a thread of its own as soon as the pk blob is complete (the seam hashes it).
int seam_keygen(const zkg_r1cs *cs, zk::OwnedCsr *owned, const std::function<void()> &under_gpu, unsigned char **pk_out, size_t *pk_len,
                unsigned char **vk_out, size_t *vk_len, zkg_crs **crs_out, const std::function<void(const unsigned char *, size_t)> &on_blob) {
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) fprintf(stderr, "[zkg seam keygen] %-26s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    *pk_out = *vk_out = nullptr; *crs_out = nullptr;
    // ---- the constraint rows' text starts as soon as the generator has fixed the system it stores (a thread of its own that fans out over
    //      the host pool): it is the longest host-only part of the blob and needs nothing the GPU computes
    int device = 0; (void)hipGetDevice(&device);                               // threads started below bind to the caller's device
    std::unique_ptr<zkg_keypair> kp;                                            // (declared first: outlives every thread that reads it)
    std::vector<ser::Writer> rows_part; bool rows_ok = true;
    std::thread rows_thread;
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_rows{rows_thread};
    kp.reset(groth16_setup_impl(cs, nullptr, owned, under_gpu, true,
        [&](zkg_keypair *k) { rows_thread = std::thread([&rows_part, &rows_ok, k] { try { constraint_rows_text(k, rows_part); } catch (...) { rows_ok = false; } }); },
        [&] { if (rows_thread.joinable()) rows_thread.join(); }));
    if (!kp) return ZKG_ERROR;
    lap("points on the device");
    // ---- the resident key is built from the device queries on a thread of its own (null stream: H table, constraint system upload, comb
    //      tables, domain, prover slot) while this thread turns the same queries into the blob's records on a stream of its own
    zkg_crs *crs = nullptr;
    std::thread crs_thread([&] { try { (void)hipSetDevice(device); crs = crs_upload_device_queries(&kp->pk_view, nullptr); } catch (...) { crs = nullptr; } });
    struct JoinCrs { std::thread &t; zkg_crs *&c; bool keep = false; ~JoinCrs() { if (t.joinable()) t.join(); if (!keep && c) { zkg_crs_free(c); c = nullptr; } } } join_crs{crs_thread, crs};
    // ---- layout: everything before the constraint rows has a known size
    const size_t nA = (size_t)kp->n + 1, nidx = kp->b_idx.size(), nH = kp->m - 1, nL = (size_t)kp->n - kp->l;
    ser::Writer seg[5];
    seg[0].g1(kp->alpha_g1); seg[0].g1(kp->beta_g1); seg[0].g2(kp->beta_g2); seg[0].g1(kp->delta_g1); seg[0].g2(kp->delta_g2); seg[0].dec(nA);
    seg[1].buf.reserve(nidx * 8 + 64);
    seg[1].dec(nA); seg[1].dec(nidx); for (uint32_t i : kp->b_idx) seg[1].dec(i); seg[1].dec(nidx);
    seg[2].dec(nH); seg[3].dec(nL); seg[4].dec(kp->l); seg[4].dec(kp->n - kp->l); seg[4].dec(kp->C);
    const size_t run[4] = {nA * 34, nidx * 100, nH * 34, nL * 34};
    size_t at_seg[5], at_run[4], pos = 0;
    for (int i = 0; i < 5; ++i) { at_seg[i] = pos; pos += seg[i].buf.size(); if (i < 4) { at_run[i] = pos; pos += run[i]; } }
    const size_t rows_at = pos;
    size_t terms = 0; for (int k = 0; k < 3; ++k) terms += kp->col[k].size();
    const size_t rows_bound = terms * 44 + (size_t)kp->C * 3 * 12 + 64;        // an index is at most 10 digits + '\n', a count likewise
    unsigned char *pk = (unsigned char *)malloc(rows_at + rows_bound);          // (untouched pages of the bound cost nothing; shrunk below)
    if (!pk) { set_error("seam_keygen: out of memory"); return ZKG_ERROR; }
    struct FreeOnExit { unsigned char *&p; ~FreeOnExit() { free(p); } } free_pk{pk};
    for (int i = 0; i < 5; ++i) memcpy(pk + at_seg[i], seg[i].buf.data(), seg[i].buf.size());
    // ---- the GPU compresses the points into the blob's records; they come back, run by run, straight into the blob
    {
        ScopedDevBuf d_rec, d_idx;
        hipStream_t st = nullptr;
        if (!hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate", __FILE__, __LINE__)) return ZKG_ERROR;
        struct DropStream { hipStream_t s; ~DropStream() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } drop_stream{st};
        size_t off[4], total = 0;
        for (int i = 0; i < 4; ++i) { off[i] = total; total += (run[i] + 15) & ~(size_t)15; }
        if (d_rec.reserve(total + 16) || d_idx.reserve(nidx * 4 + 16) ||
            (nidx && !hip_ok(hipMemcpyAsync(d_idx.p, kp->b_idx.data(), nidx * 4, hipMemcpyHostToDevice, st), "H2D", __FILE__, __LINE__))) return ZKG_ERROR;
        uint8_t *r = d_rec.as<uint8_t>();
        if (compress_g1_records(kp->dA.as<G1Affine>(), nA, r + off[0], st) || compress_kc_records(kp->dB2.as<G2Affine>(), kp->dB1.as<G1Affine>(), d_idx.as<uint32_t>(), nidx, r + off[1], st) ||
            compress_g1_records(kp->dH.as<G1Affine>(), nH, r + off[2], st) || compress_g1_records(kp->dL.as<G1Affine>(), nL, r + off[3], st)) { set_error("seam_keygen: compression launch failed"); return ZKG_ERROR; }
        for (int i = 0; i < 4; ++i)
            if (run[i] && !hip_ok(hipMemcpyAsync(pk + at_run[i], r + off[i], run[i], hipMemcpyDeviceToHost, st), "D2H", __FILE__, __LINE__)) return ZKG_ERROR;
        if (!hip_ok(hipStreamSynchronize(st), "sync", __FILE__, __LINE__)) return ZKG_ERROR;
    }
    lap("records compressed + copied");
    rows_thread.join();
    size_t rows_len = 0;
    if (rows_ok) {
        std::vector<size_t> at(rows_part.size() + 1, rows_at);
        for (size_t i = 0; i < rows_part.size(); ++i) at[i + 1] = at[i] + rows_part[i].buf.size();
        rows_len = at.back() - rows_at;
        if (rows_len > rows_bound) rows_ok = false;
        else host_parallel_for((int)rows_part.size(), [&](int i) { if (!rows_part[i].buf.empty()) memcpy(pk + at[i], rows_part[i].buf.data(), rows_part[i].buf.size()); });
    }
    if (!rows_ok) { set_error("seam_keygen: constraint rows failed"); return ZKG_ERROR; }
    const size_t len = rows_at + rows_len;
    { unsigned char *shrunk = (unsigned char *)realloc(pk, len); if (shrunk) pk = shrunk; }
    lap("pk blob complete");
    std::thread blob_thread;
    if (on_blob) blob_thread = std::thread([&] { try { on_blob(pk, len); } catch (...) {} });
    struct Join2 { std::thread &t; ~Join2() { if (t.joinable()) t.join(); } } join_blob{blob_thread};
    // ---- vk (l + 1 points, host)
    const size_t vlen = zkg_keypair_vk_blob(kp.get(), nullptr, 0);
    unsigned char *vk = (unsigned char *)malloc(vlen);
    if (!vk || zkg_keypair_vk_blob(kp.get(), vk, vlen) != vlen) { free(vk); set_error("seam_keygen: vk blob failed"); return ZKG_ERROR; }
    crs_thread.join();
    lap("resident key built");
    if (!crs) { free(vk); return ZKG_ERROR; }
    if (blob_thread.joinable()) blob_thread.join();
    join_crs.keep = true;
    // the keypair's host vectors (the constraint system, 100 MB at 20 payloads) and device queries go back to their allocators on a thread
    // of their own: nothing below needs them
    keypair_discard(kp.release(), device);
    *pk_out = pk; *pk_len = len; *vk_out = vk; *vk_len = vlen; *crs_out = crs;
    pk = nullptr;                                                               // handed over
    return ZKG_OK;
}
