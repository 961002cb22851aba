// zklaim_circuit.hip — zklaim's credential circuit and public-input map on the host (no device code in this file).
//
// Replaces, for the prover path, what /root/reference/zklaim/snark.cpp:113-118 does with libsnark's protoboard:
//     zklaim_gadget<FieldT> g(pb, ctx); g.generate_r1cs_constraints(); g.generate_r1cs_witness(ctx);
// (zklaim/zklaim_gadget.cpp:153-784) and zklaim_input_map (zklaim_gadget.cpp:115-150).  The statement is the reference's:
// per payload, SHA-256(pre) == hash for the 48-byte pre-image, each of the five 64-bit attributes stands in the relation
// selected by the one-hot op to its reference value, exactly one op is selected per attribute, the compared values are
// tied to plvars.  One deliberate difference: the reference assigns the pack_PL / pack_REF / pack_OPS packings as witnesses
// (zklaim_gadget.cpp:768-769,780) but never generates their constraints (absent from :583-699), which leaves refvals /
// opsvals / plvars free: a prover who knows the pre-image could then satisfy ANY public predicate (set the op one-hot to noop).
// Keys made here are not interchangeable with libsnark-made keys anyway (gadgets.hpp), so the packings ARE enforced here
// (78 more constraints per payload, no extra variable, same witness).  ZKG_CIRCUIT_REFERENCE_QUIRK (include/zkg.h) asks
// for the reference's unbound shape; the choice lives in the constraint system inside the key, not in the process.
#include "common.hpp"
#include "../../include/zkg.h"
#include "../../include/zklaim_abi.h"
#include "host/gadgets.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>

using namespace zk;
using namespace zk::circuit;

namespace {

const size_t DIGEST = 256, FR_CAPACITY = 253;              // FieldT::capacity() of alt_bn128 Fr

// memtobv (libsnark_wrapper.cpp:65-74): most significant bit of each byte first
std::vector<bool> memtobv(const unsigned char *mem, size_t nbits) {
    std::vector<bool> r(nbits);
    for (size_t i = 0; i < nbits / 8; ++i) for (size_t k = 0; k < 8; ++k) r[i * 8 + k] = (mem[i] >> (7 - k)) & 1;
    return r;
}
// set_zklaim_ops (zklaim_gadget.cpp:71-104): one-hot byte per operation inside an 8-byte slot
void set_ops(unsigned char *buf, int op) {
    switch (op) {
    case zklaim_less: buf[0] = 1; break;          case zklaim_less_or_eq: buf[1] = 1; break;
    case zklaim_eq: buf[2] = 1; break;            case zklaim_greater_or_eq: buf[3] = 1; break;
    case zklaim_greater: buf[4] = 1; break;       case zklaim_not_eq: buf[5] = 1; break;
    case zklaim_noop: buf[6] = 1; break;          default: break;
    }
}
void payload_public_bytes(const zklaim_payload &pl, unsigned char refs[64], unsigned char ops[64]) {
    memset(refs, 0, 64); memset(ops, 0, 64);
    for (int j = 0; j < 5; ++j) { memcpy(refs + 8 * j, &pl.data_ref[j], 8); set_ops(ops + 8 * j, pl.data_op[j]); }
}
// the byte-reversed bit order zklaim_gadget uses for PL / REF / OPS (zklaim_gadget.cpp:447-469): within every byte the
// bits are taken least significant first, so that 64 of them pack to the little-endian u64 of 8 bytes
std::vector<Var> byte_lsb_first(const std::vector<Var> &bits) {
    std::vector<Var> o; o.reserve(bits.size());
    for (size_t l = 0; l < bits.size() / 8; ++l) for (int k = 7; k >= 0; --k) o.push_back(bits[l * 8 + k]);
    return o;
}

}  // namespace

struct zkg_circuit {
    Builder pb;
    Builder::Csr A, B, C;
    bool has_witness = false;
    bool sparse_built = false; std::vector<uint32_t> full_index; std::vector<uint64_t> full_values;
};

// The variable storage of a finished circuit (18 MB at 20 payloads) is handed to the next one: a prover calls this once per proof,
// and fresh pages cost more than the witness pass itself.
namespace {
std::mutex g_store_mu; std::vector<Fr> g_spare_val; std::vector<uint8_t> g_spare_nz;
void take_storage(Builder &pb) {
    std::lock_guard<std::mutex> lk(g_store_mu);
    if (g_spare_val.capacity()) { g_spare_val.clear(); g_spare_nz.clear(); pb.val.swap(g_spare_val); pb.nz.swap(g_spare_nz); pb.val.push_back(Fr::one()); pb.nz.push_back(1); }
}
void give_storage(Builder &pb) {
    std::lock_guard<std::mutex> lk(g_store_mu);
    if (pb.val.capacity() > g_spare_val.capacity()) { g_spare_val.swap(pb.val); g_spare_nz.swap(pb.nz); }
}
}  // namespace

static zkg_circuit *build_zklaim(const zklaim_ctx *ctx, bool with_witness, bool witness_only = false, bool reference_quirk = false) {
    zkg_circuit *ck = new zkg_circuit();
    Builder &pb = ck->pb;
    static const bool dbg = getenv("ZKG_DEBUG_TIMING") != nullptr;
    auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg) fprintf(stderr, "[zkg circuit] %-28s %8.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    take_storage(pb);
    pb.recording = !witness_only;                           // proving: the resident key already holds the constraint system
    pb.bits_lazy = witness_only;                            // ... and its witness goes up as tags: 0 / 1 variables need no field value yet
    pb.reserve(28000 * (ctx->num_of_payloads + 1));
    const size_t k = ctx->num_of_payloads;
    const bool bind_packings = !reference_quirk;
    std::vector<const zklaim_payload *> pls;
    for (const zklaim_wrap_payload_ctx *cur = ctx->pl_ctx_head; cur; cur = cur->next) pls.push_back(&cur->pl);
    if (pls.size() != k) { delete ck; set_error("zklaim circuit: num_of_payloads disagrees with the payload list"); return nullptr; }

    // ---- allocation, in the order of the reference's constructor (zklaim_gadget.cpp:348-540)
    const size_t input_bits = DIGEST * k * 5;
    const size_t n_inputs = (input_bits + FR_CAPACITY - 1) / FR_CAPACITY;
    std::vector<Var> input_fe = pb.alloc_n(n_inputs);
    pb.set_input_sizes((uint32_t)n_inputs);
    Var zero = pb.alloc();
    std::vector<std::array<Var, 5>> data(k), less(k), leq(k);
    for (size_t i = 0; i < k; ++i) for (int j = 0; j < 5; ++j) { data[i][j] = pb.alloc(); less[i][j] = pb.alloc(); leq[i][j] = pb.alloc(); }
    std::vector<Var> plvars = pb.alloc_n(6 * k), refvals = pb.alloc_n(8 * k), opsvals = pb.alloc_n(64 * k);
    std::vector<std::vector<Var>> h_bits(k), ref_bits(k), ops_bits(k), r_bits(k);
    std::vector<Var> input_as_bits;
    for (size_t i = 0; i < k; ++i) {
        h_bits[i] = pb.alloc_n(DIGEST); ref_bits[i] = pb.alloc_n(2 * DIGEST); ops_bits[i] = pb.alloc_n(2 * DIGEST);
        for (auto *v : {&h_bits[i], &ref_bits[i], &ops_bits[i]}) input_as_bits.insert(input_as_bits.end(), v->begin(), v->end());
    }
    for (size_t i = 0; i < k; ++i) r_bits[i] = pb.alloc_n(DIGEST + 128);
    std::vector<Var> PL, REF, OPS;
    for (size_t i = 0; i < k; ++i) { auto t = byte_lsb_first(r_bits[i]); PL.insert(PL.end(), t.begin(), t.end()); }
    for (size_t i = 0; i < k; ++i) { auto t = byte_lsb_first(ref_bits[i]); REF.insert(REF.end(), t.begin(), t.end()); }
    for (size_t i = 0; i < k; ++i) { auto t = byte_lsb_first(ops_bits[i]); OPS.insert(OPS.end(), t.begin(), t.end()); }

    // ---- witness values that do not depend on gadget internals (zklaim_gadget.cpp:705-783)
    std::vector<std::array<uint64_t, 5>> attr(k), refv(k);
    if (with_witness) {
        pb.set_bit(zero, false);
        host_parallel_for((int)k, [&](int ii) {                                  // payloads touch disjoint variables
            const size_t i = (size_t)ii;
            const zklaim_payload &pl = *pls[i];
            unsigned char refs[64], ops[64];
            payload_public_bytes(pl, refs, ops);
            auto set_bits = [&](const std::vector<Var> &vars, const std::vector<bool> &bv) { for (size_t b = 0; b < vars.size(); ++b) pb.set_bit(vars[b], bv[b]); };
            set_bits(r_bits[i], memtobv(pl.pre, 384));
            set_bits(h_bits[i], memtobv(pl.hash, 256));
            set_bits(ref_bits[i], memtobv(refs, 512));
            set_bits(ops_bits[i], memtobv(ops, 512));
            for (int j = 0; j < 5; ++j) {
                memcpy(&attr[i][j], pl.pre + 8 * j, 8);                        // extractFromBV: little-endian u64 of the slot
                refv[i][j] = pl.data_ref[j];
                pb.set(data[i][j], Fr::from_u64(attr[i][j]));
            }
            // the payload's own packings read the bits just set and write one variable each
            for (size_t c = 6 * i; c < 6 * (i + 1); ++c) assign_packing(pb, PL, c * 64, (c + 1) * 64, plvars[c]);
            for (size_t c = 8 * i; c < 8 * (i + 1); ++c) assign_packing(pb, REF, c * 64, (c + 1) * 64, refvals[c]);
            for (size_t c = 64 * i; c < 64 * (i + 1); ++c) assign_packing(pb, OPS, c * 8, (c + 1) * 8, opsvals[c]);
        });
        // the public-input packings span payloads (253 bits each, assembled natively): too little work for another trip through the pool
        for (size_t c = 0; c < n_inputs; ++c) assign_packing(pb, input_as_bits, c * FR_CAPACITY, std::min(input_bits, (c + 1) * FR_CAPACITY), input_fe[c]);
    }

    lap("allocation + public values");
    // ---- constraints (zklaim_gadget.cpp:583-699)
    for (size_t c = 0; c < n_inputs; ++c)                                       // unpack_inputs, with booleanity of every public bit
        enforce_packing(pb, input_as_bits, c * FR_CAPACITY, std::min(input_bits, (c + 1) * FR_CAPACITY), input_fe[c], true);
    pb.enforce(LC::constant(1), LC(zero), LC());                               // zero == 0
    if (bind_packings) {
        for (size_t c = 0; c < 6 * k; ++c) enforce_packing(pb, PL, c * 64, (c + 1) * 64, plvars[c], false);
        for (size_t c = 0; c < 8 * k; ++c) enforce_packing(pb, REF, c * 64, (c + 1) * 64, refvals[c], false);
        for (size_t c = 0; c < 64 * k; ++c) enforce_packing(pb, OPS, c * 8, (c + 1) * 8, opsvals[c], false);
    }
    auto payload_gadgets = [&](Builder &pb, size_t i) {                           // the per-payload sub-circuit; `pb` may be a view
        for (Var v : r_bits[i]) pb.enforce_boolean(v);                          // digest_variable r: booleanity
        for (int j = 0; j < 5; ++j) {                                           // comparison_gadget(64, data_j, refvals[j + 8 i], less, less_or_eq)
            Comparison cmp = comparison_alloc(pb, 64, leq[i][j]);
            comparison_constraints(pb, cmp, 64, data[i][j], refvals[j + 8 * i], less[i][j], leq[i][j]);
            if (with_witness) comparison_witness(pb, cmp, 64, attr[i][j], refv[i][j], less[i][j], leq[i][j]);
        }
        if (pb.recording) for (int j = 0; j < 5; ++j) {                                           // op selection: op * relation = op
            auto op = [&](int o) { return LC(opsvals[o + j * 8 + i * 64]); };
            LC L(less[i][j]), E(leq[i][j]), one = LC::constant(1);
            pb.enforce(op(0), L, op(0));                                        // <
            pb.enforce(op(1), E, op(1));                                        // <=
            pb.enforce(op(2), E, op(2));                                        // == : <= ...
            pb.enforce(op(2), L, LC());                                         //      ... and not <
            pb.enforce(op(3), one - L, op(3));                                  // >=
            pb.enforce(op(4), one - E, op(4));                                  // >
            pb.enforce(op(5), L + (one - E), op(5));                            // !=
            pb.enforce(op(6), one, op(6));                                      // noop
        }
        if (pb.recording) for (int j = 0; j < 5; ++j) {                                           // exactly one op per attribute
            LC s; for (int o = 0; o < 7; ++o) s = s + LC(opsvals[o + j * 8 + i * 64]);
            pb.enforce(LC::constant(1), s, LC::constant(1));
        }
        for (int j = 0; j < 5; ++j) pb.enforce(LC::constant(1), LC(data[i][j]) - LC(plvars[j + i * 6]), LC());   // input validation
        // SHA-256(pre || padding) == hash: block = 384 pre-image bits + the fixed padding of a 48-byte message
        std::vector<Bit> block(512);
        for (size_t b = 0; b < 384; ++b) block[b] = Bit::var(r_bits[i][b]);
        for (size_t b = 384; b < 512; ++b) block[b] = Bit::zero();
        block[384] = Bit::one();                                                // 0x80
        block[512 - 9] = Bit::one(); block[512 - 8] = Bit::one();              // length 384 = 0x0180, big-endian in the last 64 bits
        sha256_compress_from_iv(pb, block, h_bits[i]);
    };
    if (k > 1 && !getenv("ZKG_SERIAL_CIRCUIT")) {                                  // (the env switch keeps the serial pass available to the tests)
        // The k sub-circuits allocate the same number of variables each and touch disjoint ranges: they run on the host thread pool
        // through views of the pre-sized storage; constraints recorded by the views are appended in payload order, which makes the
        // system identical to the one a serial pass records.  The per-payload sizes are a property of the circuit, measured once per
        // process (payload 0 on this thread, the rest in parallel) and remembered, so that later passes start all k at once.
        static std::atomic<uint32_t> known_vars{0}, known_cons{0}, known_terms{0};
        uint32_t per = known_vars.load();
        size_t cons_per = known_cons.load(), terms_per = known_terms.load();
        size_t first_parallel = 0;
        if (!per || (pb.recording && !cons_per)) {                                  // sizes not measured yet (for this kind of pass)
            const uint32_t before = pb.num_variables();
            const size_t cons_before = pb.cons.size(), terms_before = pb.arena.size();
            payload_gadgets(pb, 0);
            per = pb.num_variables() - before;
            if (pb.recording) { cons_per = pb.cons.size() - cons_before; terms_per = pb.arena.size() - terms_before; }
            first_parallel = 1;
        }
        const size_t np = k - first_parallel;                                       // payloads first_parallel .. k-1 go to the pool
        if (pb.recording) { pb.cons.reserve(pb.cons.size() + cons_per * np); pb.arena.reserve(pb.arena.size() + terms_per * np); }
        const uint32_t first = pb.extend((size_t)per * np);
        std::vector<char> bad(np, 0);
        std::vector<Builder> views(np);
        // (with constraints recorded this is the heavy loop of a first libsnark_prove, beside the key loader: threads of its own if the pool is taken)
        (pb.recording ? host_parallel_for_spawn : host_parallel_for)((int)np, [&](int t) {
            Builder &v = views[t];
            v = Builder::view_of(pb, first + (uint32_t)t * per, first + (uint32_t)(t + 1) * per);
            if (v.recording) { v.cons.reserve(cons_per); v.arena.reserve(terms_per); }
            payload_gadgets(v, (size_t)t + first_parallel);
            bad[t] = v.overrun || v.cursor != v.cursor_end;
        });
        for (char b : bad) if (b) { known_vars = 0; delete ck; set_error("zklaim circuit: payload sub-circuits differ in size"); return nullptr; }
        if (pb.recording) for (const Builder &v : views) pb.absorb(v);
        known_vars = per;
        if (pb.recording && cons_per) { known_cons = (uint32_t)cons_per; known_terms = (uint32_t)terms_per; }
    } else {
        for (size_t i = 0; i < k; ++i) payload_gadgets(pb, i);
    }
    lap("payload sub-circuits");
    if (pb.recording) {
        Builder::Csr *ms[3] = {&ck->A, &ck->B, &ck->C};
        const int chunks = getenv("ZKG_SERIAL_CIRCUIT") ? 1 : (int)std::min<size_t>(16, pb.cons.size() / 8192 + 1);
        pb.export_csr_chunked(ms, chunks, [](int tasks, const std::function<void(int)> &f) { host_parallel_for_spawn(tasks, f); });
    }
    lap("CSR exported");
    ck->has_witness = with_witness;             // the witness is pb.val[1..] itself: Fr is the ABI's 4 x u64 Montgomery element
    return ck;
}

extern "C" {

zkg_circuit *zkg_zklaim_circuit_new(const zklaim_ctx *ctx, int flags) {
    if (!ctx) { set_error("zkg_zklaim_circuit_new: null ctx"); return nullptr; }
    return build_zklaim(ctx, (flags & ZKG_CIRCUIT_WITH_WITNESS) != 0, false, (flags & ZKG_CIRCUIT_REFERENCE_QUIRK) != 0);
}
// witness only (generate_r1cs_witness without re-deriving the constraints): what a prover holding a resident key needs
zkg_circuit *zkg_zklaim_witness_new(const zklaim_ctx *ctx) {
    if (!ctx) { set_error("zkg_zklaim_witness_new: null ctx"); return nullptr; }
    return build_zklaim(ctx, true, true);
}
void zkg_circuit_free(zkg_circuit *c) { if (c) give_storage(c->pb); delete c; }

int zkg_circuit_r1cs(const zkg_circuit *c, zkg_r1cs *out) {
    if (!c || !out) return ZKG_ERROR;
    memset(out, 0, sizeof(*out));
    out->num_variables = c->pb.num_variables(); out->num_inputs = c->pb.num_inputs; out->num_constraints = (uint32_t)c->pb.cons.size();
    out->a_rowptr = c->A.rowptr.data(); out->a_col = c->A.col.data(); out->a_val = c->A.val.data();
    out->b_rowptr = c->B.rowptr.data(); out->b_col = c->B.col.data(); out->b_val = c->B.val.data();
    out->c_rowptr = c->C.rowptr.data(); out->c_col = c->C.col.data(); out->c_val = c->C.val.data();
    return ZKG_OK;
}
const uint64_t *zkg_circuit_witness(const zkg_circuit *c) {
    static_assert(sizeof(Fr) == 32, "Fr must be the ABI's 32-byte element");
    if (c && c->has_witness) const_cast<zkg_circuit *>(c)->pb.materialize_bits();             // a witness-only pass wrote tags only for its bits
    return (c && c->has_witness) ? reinterpret_cast<const uint64_t *>(c->pb.val.data() + 1) : nullptr;
}
// sparse form of the same witness for zkg_groth16_prove_sparse: one tag per variable (0 zero, 1 one, 2 listed) and the listed variables.
// The lists live in the circuit object and are built on the first call.
int zkg_circuit_sparse_witness(const zkg_circuit *c_, const uint8_t **tags, const uint32_t **full_index, const uint64_t **full_values, size_t *count) {
    zkg_circuit *c = const_cast<zkg_circuit *>(c_);
    if (!c || !c->has_witness || !tags || !full_index || !full_values || !count) return ZKG_ERROR;
    const uint32_t n = c->pb.num_variables();
    if (!c->sparse_built) {
        // the tag scan (219 K bytes at 8 payloads, 0.2 ms on one thread) runs in chunks on the host pool; chunk lists are joined in index order
        const uint8_t *t = c->pb.nz.data() + 1;
        const int chunks = (int)std::min<uint32_t>(16, n / 16384 + 1);
        std::vector<std::vector<uint32_t>> found((size_t)chunks);
        host_parallel_for(chunks, [&](int ch) {
            const uint32_t lo = (uint32_t)((uint64_t)n * ch / chunks), hi = (uint32_t)((uint64_t)n * (ch + 1) / chunks);
            for (uint32_t v = lo; v < hi; ++v) if (t[v] == 2) found[(size_t)ch].push_back(v);
        });
        size_t total = 0;
        for (auto &f : found) total += f.size();
        c->full_index.clear(); c->full_index.reserve(total); c->full_values.resize(4 * total);
        for (auto &f : found) c->full_index.insert(c->full_index.end(), f.begin(), f.end());
        for (size_t i = 0; i < total; ++i) memcpy(&c->full_values[4 * i], c->pb.val[c->full_index[i] + 1].v, 32);
        c->sparse_built = true;
    }
    *tags = c->pb.nz.data() + 1; *full_index = c->full_index.data(); *full_values = c->full_values.data(); *count = c->full_index.size();
    return ZKG_OK;
}
int zkg_circuit_is_satisfied(const zkg_circuit *c) { return c && c->has_witness && c->pb.recording && c->pb.is_satisfied() ? 1 : 0; }
uint32_t zkg_circuit_num_variables(const zkg_circuit *c) { return c ? c->pb.num_variables() : 0; }
long zkg_circuit_first_unsatisfied(const zkg_circuit *c) { return c ? (long)c->pb.first_unsatisfied() : -2; }

// zklaim_input_map (zklaim_gadget.cpp:115-150): hash || refs(512 b) || ops(512 b) per payload, 253 bits per field element
size_t zkg_zklaim_input_map(const zklaim_ctx *ctx, uint64_t *out, size_t cap_elems) {
    if (!ctx) return 0;
    std::vector<bool> bits;
    for (const zklaim_wrap_payload_ctx *cur = ctx->pl_ctx_head; cur; cur = cur->next) {
        unsigned char refs[64], ops[64];
        payload_public_bytes(cur->pl, refs, ops);
        for (auto &v : {memtobv(cur->pl.hash, 256), memtobv(refs, 512), memtobv(ops, 512)}) bits.insert(bits.end(), v.begin(), v.end());
    }
    size_t n = (bits.size() + FR_CAPACITY - 1) / FR_CAPACITY;
    if (!out || cap_elems < n) return n;
    for (size_t c = 0; c < n; ++c) {
        Fr s = Fr::zero(), w = Fr::one();
        for (size_t b = c * FR_CAPACITY; b < std::min(bits.size(), (c + 1) * FR_CAPACITY); ++b) { if (bits[b]) s += w; w = w.dbl(); }
        memcpy(out + 4 * c, s.v, 32);
    }
    return n;
}

}  // extern "C"

// the circuit gives up its CSR matrices (the seam's key generation: no copy of 100 MB); zkg_circuit_r1cs is empty afterwards
void circuit_release_csr(zkg_circuit *c, zk::OwnedCsr &out) {
    Builder::Csr *ms[3] = {&c->A, &c->B, &c->C};
    for (int k = 0; k < 3; ++k) { out.rp[k].swap(ms[k]->rowptr); out.col[k].swap(ms[k]->col); out.val[k].swap(ms[k]->val); }
}
