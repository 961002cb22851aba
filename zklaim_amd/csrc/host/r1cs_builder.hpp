// r1cs_builder.hpp — host-side constraint-system builder (what zklaim uses libsnark's protoboard / pb_variable /
// linear_combination / r1cs_constraint for: /root/reference/zklaim/snark.cpp:113-118, zklaim_gadget.cpp:18-20).
// Variable 0 is the constant ONE; variables 1..num_inputs are the primary input; the rest is the auxiliary input.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>
#include "../fp.hip.hpp"

namespace zk { namespace circuit {

typedef uint32_t Var;                    // 0 == ONE

struct LC {                              // sparse linear combination sum coeff_i * x_i
    std::vector<std::pair<Var, Fr>> t;
    LC() {}
    LC(Var v) { t.push_back({v, Fr::one()}); }
    static LC constant(const Fr &c) { LC r; if (!c.is_zero()) r.t.push_back({0, c}); return r; }
    static LC constant(uint64_t c) { return constant(Fr::from_u64(c)); }
    LC &add(Var v, const Fr &c) { if (!c.is_zero()) t.push_back({v, c}); return *this; }
    LC operator+(const LC &o) const { LC r = *this; r.t.insert(r.t.end(), o.t.begin(), o.t.end()); return r; }
    LC operator-(const LC &o) const { LC r = *this; for (auto &e : o.t) r.t.push_back({e.first, e.second.neg()}); return r; }
    LC operator*(const Fr &c) const { LC r; if (c.is_zero()) return r; for (auto &e : t) r.t.push_back({e.first, e.second * c}); return r; }
    LC operator*(uint64_t c) const { return (*this) * Fr::from_u64(c); }
    bool is_constant() const { for (auto &e : t) if (e.first != 0) return false; return true; }
};

// a recorded constraint: three runs of terms in the builder's arena (one allocation for the whole system instead of three
// small vectors per constraint: building is faster and freeing half a million constraints is two deallocations)
struct Span { uint32_t off, len; };
struct Constraint { Span a, b, c; };

struct Builder {
    std::vector<Fr> val;                 // val[0] == 1
    std::vector<uint8_t> nz;             // nz[v]: 0 = zero, 1 = exactly one, 2 = any other value.  The boolean gadgets read it as
                                         // "non-zero" on the witness path; the prover's sparse upload reads it as a tag.
    std::vector<Constraint> cons;
    std::vector<std::pair<Var, Fr>> arena;
    uint32_t num_inputs = 0;
    bool recording = true;               // false: witness-only pass (the proving key already holds the constraint system)
    bool bits_lazy = false;              // set by the owner before a witness-only pass: boolean variables carry their tag only until materialize_bits()
    // A view allocates from a pre-sized range [cursor, cursor_end) of its root's storage and reads / writes values there: the
    // per-payload sub-circuits of a witness-only pass are independent and run on separate threads through views.
    Builder *root = nullptr; uint32_t cursor = 0, cursor_end = 0, cursor_begin = 0; bool overrun = false;
    Builder() { val.push_back(Fr::one()); nz.push_back(1); }
    static Builder view_of(Builder &r, uint32_t begin, uint32_t end) { Builder v; v.root = &r; v.cursor = v.cursor_begin = begin; v.cursor_end = end; v.recording = r.recording; return v; }
    // append the constraints a view recorded (views keep their own cons / arena; merged in payload order the result is the serial one)
    void absorb(const Builder &v) {
        const uint32_t base = (uint32_t)arena.size();
        arena.insert(arena.end(), v.arena.begin(), v.arena.end());
        cons.reserve(cons.size() + v.cons.size());
        for (Constraint c : v.cons) { c.a.off += base; c.b.off += base; c.c.off += base; cons.push_back(c); }
    }
    void reserve(size_t nvars) { val.reserve(nvars + 1); nz.reserve(nvars + 1); }
    uint32_t extend(size_t count) { uint32_t first = (uint32_t)val.size(); val.resize(val.size() + count, Fr::zero()); nz.resize(nz.size() + count, 0); return first; }
    Var alloc() {
        if (root) {                                                          // a view: its own pre-sized range only
            if (cursor < cursor_end) return cursor++;
            overrun = true;                                                  // reported by the caller; the stand-in is one of the view's OWN variables
            return cursor_end > cursor_begin ? cursor_end - 1 : (Var)(root->val.size() - 1);     // (never the constant ONE, never another view's)
        }
        val.push_back(Fr::zero()); nz.push_back(0); return (Var)(val.size() - 1);
    }
    // n consecutive fresh variables, first index returned (values zero).  Callers write all n of them as a block (the word-level gadgets
    // store 32 / 64 tags at once), so a view that cannot supply them must still hand back n indices it owns: the LAST n of its own range
    // (stale stand-ins, like alloc()'s), with the overrun recorded for the caller to report.  A view narrower than n has no such block:
    // it returns the largest in-range start and the caller's check of `overrun` refuses the circuit (views are sized in whole payloads,
    // thousands of variables; the widest block is ~100).
    Var alloc_block(uint32_t n) {
        if (root) {
            if (cursor + n <= cursor_end) { Var f = cursor; cursor += n; return f; }
            overrun = true; cursor = cursor_end;
            if (cursor_end - cursor_begin >= n) return cursor_end - n;
            return root->val.size() > n ? (Var)(root->val.size() - n) : (Var)0;          // in bounds of the root's storage whatever happens
        }
        return (Var)extend(n);
    }
    void set(Var v, const Fr &x) { Builder &r = root ? *root : *this; r.val[v] = x; r.nz[v] = x.is_zero() ? 0 : (x == Fr::one() ? 1 : 2); }
    // A boolean variable.  In a witness-only pass only its tag is written: the tags (one byte per variable) are what the prover's sparse
    // upload and the gadgets read, and the 32-byte field value of a 0 / 1 variable is implied by it (materialize_bits() fills the values
    // in for callers that ask for the dense vector) — 27 K variables per payload, 33 bytes each otherwise.
    void set_bit(Var v, bool b) { Builder &r = root ? *root : *this; r.nz[v] = b; if (!r.bits_lazy) r.val[v] = b ? Fr::one() : Fr::zero(); }
    void materialize_bits() {
        if (!bits_lazy) return;
        const Fr one = Fr::one(), zero = Fr::zero();
        for (size_t v = 0; v < val.size(); ++v) if (nz[v] < 2) val[v] = nz[v] ? one : zero;
        bits_lazy = false;
    }
    bool is_nonzero(Var v) const { return (root ? root : this)->nz[v] != 0; }
    uint8_t *tag_data() { return (root ? root : this)->nz.data(); }              // one byte per variable (see nz); stable while no variable is appended
    const uint8_t *tag_data() const { return (root ? root : this)->nz.data(); }
    std::vector<Var> alloc_n(size_t n) {
        std::vector<Var> v(n);
        if (root) { for (auto &x : v) x = alloc(); return v; }
        const Var first = (Var)extend(n);                                 // one resize instead of n push_backs
        for (size_t i = 0; i < n; ++i) v[i] = first + (Var)i;
        return v;
    }
    void set_input_sizes(uint32_t n) { num_inputs = n; }
    uint32_t num_variables() const { return (uint32_t)val.size() - 1; }
    Span put(const LC &l) { Span sp{(uint32_t)arena.size(), (uint32_t)l.t.size()}; arena.insert(arena.end(), l.t.begin(), l.t.end()); return sp; }
    void enforce(const LC &a, const LC &b, const LC &c) { if (recording) { Span sa = put(a), sb = put(b), sc = put(c); cons.push_back({sa, sb, sc}); } }
    void enforce_boolean(Var v) { if (recording) enforce(LC(v), LC::constant(1) - LC(v), LC()); }          // v (1 - v) = 0
    Fr eval(const LC &l) const { Fr s = Fr::zero(); for (auto &e : l.t) s += e.second * val[e.first]; return s; }
    Fr eval(const Span &sp) const { Fr s = Fr::zero(); for (uint32_t i = sp.off; i < sp.off + sp.len; ++i) s += arena[i].second * val[arena[i].first]; return s; }
    bool is_satisfied() const { for (auto &c : cons) if (eval(c.a) * eval(c.b) != eval(c.c)) return false; return true; }
    size_t first_unsatisfied() const { for (size_t i = 0; i < cons.size(); ++i) if (eval(cons[i].a) * eval(cons[i].b) != eval(cons[i].c)) return i; return (size_t)-1; }

    // CSR export (terms on the same variable merged, zero coefficients dropped): the zkg_r1cs layout
    struct Csr { std::vector<uint32_t> rowptr, col; std::vector<uint64_t> val; };
    void push_row(Csr &m, const Span &sp, std::vector<std::pair<Var, Fr>> &s) const {      // s: scratch, reused from row to row
        if (sp.len == 1) {                                                                  // the common row: one term
            const auto &e = arena[sp.off];
            if (!e.second.is_zero()) { m.col.push_back(e.first); uint64_t l4[4]; memcpy(l4, e.second.v, 32); m.val.insert(m.val.end(), l4, l4 + 4); }
            m.rowptr.push_back((uint32_t)m.col.size());
            return;
        }
        s.assign(arena.begin() + sp.off, arena.begin() + sp.off + sp.len);
        std::stable_sort(s.begin(), s.end(), [](const std::pair<Var, Fr> &x, const std::pair<Var, Fr> &y) { return x.first < y.first; });
        for (size_t i = 0; i < s.size();) {
            Fr acc = Fr::zero(); size_t j = i;
            for (; j < s.size() && s[j].first == s[i].first; ++j) acc += s[j].second;
            if (!acc.is_zero()) { m.col.push_back(s[i].first); uint64_t l4[4]; memcpy(l4, acc.v, 32); m.val.insert(m.val.end(), l4, l4 + 4); }
            i = j;
        }
        m.rowptr.push_back((uint32_t)m.col.size());
    }
    void export_matrix(int which, Csr &m) const {                                           // 0 = A, 1 = B, 2 = C; independent of each other
        m.rowptr.assign(1, 0); m.col.clear(); m.val.clear();
        m.rowptr.reserve(cons.size() + 1);
        size_t terms = 0;
        for (auto &c : cons) terms += (which == 0 ? c.a : which == 1 ? c.b : c.c).len;
        m.col.reserve(terms); m.val.reserve(4 * terms);
        std::vector<std::pair<Var, Fr>> scratch;
        for (auto &c : cons) push_row(m, which == 0 ? c.a : which == 1 ? c.b : c.c, scratch);
    }
    void export_csr(Csr &A, Csr &B, Csr &C) const { export_matrix(0, A); export_matrix(1, B); export_matrix(2, C); }
    // The same three matrices, rows cut into `chunks` ranges per matrix: run(tasks, f) executes f(0..tasks-1) (a thread pool's parallel-for).
    // A range is exported into a matrix of its own and the pieces are then joined — rows are independent, the result is export_matrix's.
    template <class Run> void export_csr_chunked(Csr *out[3], int chunks, Run run) const {
        const size_t rows = cons.size();
        if (chunks < 1) chunks = 1;
        std::vector<Csr> piece((size_t)3 * chunks);
        run(3 * chunks, [&](int task) {
            const int which = task / chunks, ch = task % chunks;
            const size_t lo = rows * (size_t)ch / chunks, hi = rows * (size_t)(ch + 1) / chunks;
            Csr &m = piece[task];
            m.rowptr.assign(1, 0); m.rowptr.reserve(hi - lo + 1);
            size_t terms = 0;
            for (size_t r = lo; r < hi; ++r) terms += (which == 0 ? cons[r].a : which == 1 ? cons[r].b : cons[r].c).len;
            m.col.reserve(terms); m.val.reserve(4 * terms);
            std::vector<std::pair<Var, Fr>> scratch;
            for (size_t r = lo; r < hi; ++r) push_row(m, which == 0 ? cons[r].a : which == 1 ? cons[r].b : cons[r].c, scratch);
        });
        run(3, [&](int which) {
            Csr &m = *out[which];
            size_t terms = 0;
            for (int ch = 0; ch < chunks; ++ch) terms += piece[(size_t)which * chunks + ch].col.size();
            m.rowptr.assign(1, 0); m.rowptr.reserve(rows + 1); m.col.clear(); m.col.reserve(terms); m.val.clear(); m.val.reserve(4 * terms);
            for (int ch = 0; ch < chunks; ++ch) {
                const Csr &pc = piece[(size_t)which * chunks + ch];
                const uint32_t base = (uint32_t)m.col.size();
                for (size_t r = 1; r < pc.rowptr.size(); ++r) m.rowptr.push_back(base + pc.rowptr[r]);
                m.col.insert(m.col.end(), pc.col.begin(), pc.col.end());
                m.val.insert(m.val.end(), pc.val.begin(), pc.val.end());
            }
        });
    }
};

}}  // namespace zk::circuit
