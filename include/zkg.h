/*
 * zkg.h — C ABI of the MI355X-native Groth16 prover hot path (alt_bn128).
 *
 * This is the drop-in boundary below zklaim's prover call
 *     r1cs_gg_ppzksnark_prover<ppT>(proving_key, primary_input, auxiliary_input)
 * at /root/reference/zklaim/snark.cpp:126 (reached from libsnark_prove,
 * zklaim/libsnark_wrapper.cpp:218-249, reached from zklaim_proof_generate,
 * zklaim/zklaim.c:77-80).  Everything above that call stays host C/C++; everything
 * below it is hand-written HIP for gfx950 behind the functions declared here.
 * Each entry point names the libsnark / libff / libfqfft function it replaces; those
 * live in the un-vendored submodule lib/libsnark (.gitmodules:1-6) and are cited by
 * the reference call site that reaches them.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; never throws; 0 == success
 *     (ZKLAIM_OK, zklaim/zklaim.h:38), non-zero == failure (ZKLAIM_ERROR semantics).
 *   - Field element (Fq or Fr): 4 x uint64_t little-endian limbs, MONTGOMERY form with
 *     R = 2^256 — the in-memory form of libff's Fp_model<4,...> — unless a parameter
 *     says "canonical".
 *   - G1 affine: 8 limbs  X||Y.   G2 affine: 16 limbs  X.c0||X.c1||Y.c0||Y.c1.
 *     The point at infinity is encoded as all-zero limbs ((0,0) is not on either curve).
 *   - "jac" outputs: X||Y||Z, NORMALISED (Z == Montgomery one) or infinity == (0, one, 0),
 *     i.e. what libff's to_affine_coordinates() leaves behind.  Normalised output is what
 *     makes results byte-comparable between the HIP path and the CPU oracle.
 *   - *_dev entry points take DEVICE pointers (hipMalloc'd, or torch tensor data_ptr())
 *     and a hipStream_t passed as void*; all others take HOST pointers and stage
 *     internally.
 */
#ifndef ZKG_H
#define ZKG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKG_OK 0
#define ZKG_ERROR 1          /* generic failure (bad argument, HIP error)            */
#define ZKG_UNSATISFIED 2    /* zkg_groth16_prove*: the witness violates the constraint
                                system and no proof was made.  A code of its own, so that
                                callers never have to tell it from ZKG_ERROR by the error
                                text; the seam maps it to the reference's return value 1
                                (libsnark_prove, zklaim/libsnark_wrapper.cpp:233-240)   */
#define ZKG_PROOF_BYTES 134  /* G1(34) || G2(66) || G1(34), see zkg_groth16_prove    */

/* ---- R1CS in CSR form (three matrices).  Column 0 is the constant ONE, columns
 *      1..num_inputs the primary input, the rest the auxiliary input: the layout of
 *      libsnark's r1cs_constraint_system + variable indices (used through
 *      pb.get_constraint_system(), snark.cpp:87).                                    */
typedef struct zkg_r1cs {
    uint32_t num_variables;    /* n, excluding the constant                           */
    uint32_t num_inputs;       /* l                                                   */
    uint32_t num_constraints;  /* C                                                   */
    uint32_t reserved;
    const uint32_t *a_rowptr, *a_col; const uint64_t *a_val;   /* rowptr[C+1], col[nnz], val[nnz*4] (Fr) */
    const uint32_t *b_rowptr, *b_col; const uint64_t *b_val;
    const uint32_t *c_rowptr, *c_col; const uint64_t *c_val;
} zkg_r1cs;

/* ---- Proving key material as flat host arrays: the fields of
 *      r1cs_gg_ppzksnark_proving_key<ppT> (imported at libsnark_wrapper.cpp:160-168).
 *      B_query is libsnark's sparse knowledge_commitment_vector<G2,G1> densified:
 *      absent entries are infinity.                                                   */
typedef struct zkg_pk {
    zkg_r1cs cs;               /* the (possibly A/B-swapped) system stored in the pk   */
    uint32_t log_m;            /* ceil(log2 m) of the evaluation domain size m         */
    uint32_t domain_size;      /* m as libfqfft's get_evaluation_domain(C+l+1) picks it:
                                  2^log_m (basic_radix2_domain; 0 means the same) or
                                  2^(log_m-1) + 2^b, b < log_m-1 (step_radix2_domain)   */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1;   /* 8 limbs each                   */
    const uint64_t *beta_g2, *delta_g2;              /* 16 limbs each                  */
    const uint64_t *A_query;   /* (n+1) x 8                                            */
    const uint64_t *B_g1;      /* (n+1) x 8                                            */
    const uint64_t *B_g2;      /* (n+1) x 16                                           */
    const uint64_t *H_query;   /* (m-1) x 8                                            */
    const uint64_t *L_query;   /* (n-l) x 8                                            */
} zkg_pk;

typedef struct zkg_crs zkg_crs;   /* opaque: device-resident proving key + domain tables */

/* ---- lifecycle -------------------------------------------------------------------- */
/* Replaces ppT::init_public_params() (libsnark_wrapper.cpp:204,227,259).  Selects HIP
 * device `device` (>=0) for the calling process (one process per GPU), uploads constant
 * tables.  Re-entrant; fails (non-zero) if no HIP device is usable — there is no CPU
 * fallback behind this ABI.                                                            */
int  zkg_init(int device);
void zkg_shutdown(void);
const char *zkg_last_error(void);
/* number of compute units / device name of the device zkg_init selected (diagnostics) */
int  zkg_device_info(char *name, size_t name_len, int *compute_units);

/* ---- NTT: libfqfft basic_radix2_domain<Fr>::FFT / iFFT / cosetFFT / icosetFFT
 *      (reached from r1cs_to_qap_witness_map inside the call at snark.cpp:126).
 *      a: N = 2^logN Fr elements, Montgomery, natural order in and out, in place.
 *      inverse: 0 forward, 1 inverse (includes the 1/N scaling).
 *      coset:   0 plain, 1 coset with g = Fr::multiplicative_generator (= 5).         */
int zkg_ntt(uint64_t *a, unsigned logN, int inverse, int coset);
int zkg_ntt_dev(void *d_a, unsigned logN, int inverse, int coset, void *stream);

/* ---- Domain choice and the non-power-of-two case.
 *      zkg_evaluation_domain_size: libfqfft::get_evaluation_domain(min_size) as
 *      r1cs_to_qap_instance_map / _witness_map call it with min_size = C + l + 1
 *      (inside snark.cpp:91 and :126): *m = the domain size, *is_step = 1 when it is a
 *      step_radix2_domain (m = 2^a + 2^b, b < a) instead of a basic_radix2_domain.
 *      zklaim's circuit lands on a step domain for 11 of its 20 payload counts
 *      (e.g. 3 payloads: m = 2^16 + 2^15).
 *      zkg_ntt_domain: the same four transforms as zkg_ntt on the domain of size m,
 *      m = 2^k or 2^a + 2^b: libfqfft step_radix2_domain<Fr>::FFT / iFFT / cosetFFT /
 *      icosetFFT for the latter.  a: m Fr elements, Montgomery, in place.             */
int zkg_evaluation_domain_size(size_t min_size, size_t *m, int *is_step);
int zkg_ntt_domain(uint64_t *a, size_t m, int inverse, int coset);
int zkg_ntt_domain_dev(void *d_a, size_t m, int inverse, int coset, void *stream);

/* ---- MSM: libff::multi_exp<G1,Fr,multi_exp_method_BDLO12> and
 *      multi_exp_with_mixed_addition (A/H/L queries), and the G2 half of
 *      kc_multi_exp_with_mixed_addition (B query); reached from snark.cpp:126.
 *      bases: affine Montgomery; scalars: N x 4 limbs, CANONICAL (as_bigint()) values
 *      in [0, r).  out: normalised jac.                                               */
int zkg_msm_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_jac[12]);
int zkg_msm_g2(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_jac[24]);
/* device-resident inputs.  scalars_mont is a set of flags:
 *   ZKG_SCALARS_MONT         the scalars are Montgomery Fr (a witness vector), converted on the fly;
 *   ZKG_SCALARS_MOSTLY_BITS  the caller knows most scalars are 0 or 1 — libff's
 *                            multi_exp_with_mixed_addition case (A / B / L queries over a witness), as opposed to plain multi_exp
 *                            (H query): the digits are then sorted in one pass; the two-pass sort that is faster for uniformly
 *                            random scalars degrades when half of them share a digit.  The result never depends on the flag.
 * The result is written to HOST memory (out_jac) after the stream is synchronised.     */
#define ZKG_SCALARS_MONT 1
#define ZKG_SCALARS_MOSTLY_BITS 2
int zkg_msm_g1_dev(const void *d_bases, const void *d_scalars, size_t n, int scalars_mont,
                   uint64_t out_jac[12], void *stream);
int zkg_msm_g2_dev(const void *d_bases, const void *d_scalars, size_t n, int scalars_mont,
                   uint64_t out_jac[24], void *stream);
/* Bases resident on the device, scalars in HOST memory: the step SURVEY.md section 8(d) times ("wall clock around the call incl. H2D of
 * scalars; bases resident") — what a prover with a resident key does per proof when the scalars come from the host.  `scalars`: n x 4
 * canonical limbs (or Montgomery with ZKG_SCALARS_MONT) in host memory; page-locked memory (hipHostMalloc / hipHostRegister) lets the upload
 * run under the work: from 2^19 points on the job is cut by points into four pieces whose uploads overlap the sort and accumulation of the
 * pieces before them, all accumulating into one set of buckets (pageable memory works, without the overlap).  Same point as zkg_msm_g1_dev on
 * the uploaded vector, bit for bit.  `stream` as in zkg_msm_g1_dev; the result is in out_jac when the call returns.                      */
int zkg_msm_g1_host_scalars(const void *d_bases, const uint64_t *scalars, size_t n, int scalars_mont, uint64_t out_jac[12], void *stream);
/* Fixed bases kept resident WITH their per-window tables (level w = 2^(c w) P_i; c = 16 from 2^15 points on: 16 x the bases' memory plus the
 * same again as 29-bit records) — what the prover builds for a key's H query, offered for any fixed G1 base set (libff has no counterpart; its
 * multi_exp takes the bases as they are).  Every window's digit then weighs the same: one bucket set, one reduction, no doublings on the
 * host.  zkg_msm_g1_bases_upload reads n affine points from DEVICE memory (the layout zkg_msm_g1_dev takes) and builds the tables once;
 * zkg_msm_g1_resident computes sum scalars[i] * P_i for n DEVICE scalars (n = the handle's point count) — the same point as
 * zkg_msm_g1_dev, bit for bit.  Calls on one handle take turns.  Ordering: the job runs on a stream of the handle's own, behind everything
 * queued on `stream` (NULL: the null stream) at the time of the call — the work that wrote d_scalars — and the call returns when the result
 * is in out_jac.  The calling thread's current device must be the one the handle was made on (otherwise ZKG_ERROR); zkg_msm_g1_bases_free
 * may be called from any device.  bench.py reports it as a SECOND figure (extras.msm_resident_tables); the headline stays the plain path.                                                                                                  */
typedef struct zkg_msm_bases zkg_msm_bases;
zkg_msm_bases *zkg_msm_g1_bases_upload(const void *d_bases, size_t n);
int zkg_msm_g1_resident(zkg_msm_bases *bases, const void *d_scalars, size_t n, int scalars_mont, uint64_t out_jac[12], void *stream);
void zkg_msm_g1_bases_free(zkg_msm_bases *bases);
/* Window-sharded variant for multi-GPU runs where every GPU holds every base: the partial
 * sum over the Pippenger windows first_window, first_window + window_stride, ... only, each
 * already weighted by 2^(c w) — the partials of ranks g = 0..G-1 (first_window = g,
 * window_stride = G) add up to zkg_msm_g1_dev's result (zkg_g1_sum).                      */
int zkg_msm_g1_windows_dev(const void *d_bases, const void *d_scalars, size_t n, int scalars_mont,
                           unsigned first_window, unsigned window_stride, uint64_t out_jac[12], void *stream);
/* Sum of `count` normalised-jac G1 (G2) points held in HOST memory: the combine step after
 * the per-GPU partial MSMs have been all-gathered (RCCL has no elliptic-curve reduce op). */
int zkg_g1_sum(const uint64_t *points_jac, size_t count, uint64_t out_jac[12]);
int zkg_g2_sum(const uint64_t *points_jac, size_t count, uint64_t out_jac[24]);

/* ---- several GPUs in ONE process: the G1 multi-exponentiation sharded by points.
 *      zkg_init_multi: zkg_init(devices[0]) plus the per-device kernel setup of the other devices.  A device may be listed more
 *      than once (two shards on one GPU: how a single-GPU box rehearses the path).
 *      zkg_msm_g1_shards_upload: splits n affine bases (HOST pointer) into ndev contiguous shards, shard i resident on devices[i]
 *      with its own stream and workspace.
 *      zkg_msm_g1_multi: scalars n x 4 canonical limbs (HOST pointer).  One host thread per shard uploads that shard's scalar
 *      slice and runs the complete single-GPU Pippenger; the ndev partial points are added on the host (RCCL has no elliptic-curve
 *      reduction; the exchange is 96 bytes per GPU).  partials_jac (optional, ndev x 12 limbs) receives the per-shard results.
 *      Result == zkg_msm_g1 on the same inputs, bit for bit.  Calls on one handle take turns (a mutex inside the handle: a call owns
 *      every shard's scalar buffer, workspace and stream); different handles run side by side.                                */
typedef struct zkg_msm_shards zkg_msm_shards;
int zkg_init_multi(const int *devices, int ndev);
zkg_msm_shards *zkg_msm_g1_shards_upload(const uint64_t *bases, size_t n, const int *devices, int ndev);
void zkg_msm_g1_shards_free(zkg_msm_shards *shards);
size_t zkg_msm_g1_shards_count(const zkg_msm_shards *shards, size_t *points);
int zkg_msm_g1_multi(const zkg_msm_shards *shards, const uint64_t *scalars, uint64_t out_jac[12], uint64_t *partials_jac);
/* With ZKG_MULTI_RCCL=1 in the environment zkg_msm_g1_multi exchanges the shards' partial points with an RCCL all-gather (one
 * communicator per shard from ncclCommInitAll, one group call per multi-exponentiation, 96 bytes per GPU over xGMI) before the sum; the
 * library is opened at run time.  Needs every shard on its own device (RCCL refuses a device listed twice: such a handle keeps the host
 * exchange).  zkg_multi_rccl_calls: how many calls of this process went through the collective.                                      */
unsigned zkg_multi_rccl_calls(void);

/* ---- fixed-base batch: out[i] = scalars[i] * base (affine out).  The batch_exp of
 *      libsnark's generator (snark.cpp:91); used here to build synthetic bases on device. */
int zkg_g1_fixed_base_dev(const uint64_t base[8], const void *d_scalars, size_t n, void *d_out_affine, void *stream);
int zkg_g2_fixed_base_dev(const uint64_t base[16], const void *d_scalars, size_t n, void *d_out_affine, void *stream);

/* ---- CRS residency: parse once, keep on device (removes the per-call pk re-parse of
 *      libsnark_wrapper.cpp:230 and the by-value pk copy of snark.cpp:107-109).        */
zkg_crs *zkg_crs_upload(const zkg_pk *pk);
/* Same, from the byte blob zklaim keeps in ctx->pk (written by libsnark_export_pk, libsnark_wrapper.cpp:146-157, i.e.
 * operator<<(r1cs_gg_ppzksnark_proving_key) under libsnark's default flags: binary, Montgomery, compressed points).
 * Replaces libsnark_import_pk (libsnark_wrapper.cpp:160-168); the ~4n+m point decompressions (one square root each) run
 * on the GPU.  The domain is recovered from |H_query| + 1 (2^k, or a step_radix2 size 2^a + 2^b).  Counts inside the blob are
 * bounded by the bytes that follow them before anything is sized by them: a malformed blob is an error return, never a fault. */
zkg_crs *zkg_crs_upload_blob(const void *pk_blob, size_t len);
/* The host half of that loader alone (no GPU, no zkg_init): walks the sections, the sparse B index list and every constraint's terms of
 * a pk blob and reports its sizes — out[8] (optional): A_query entries, B_query values, H_query entries, L_query entries, public inputs,
 * constraints, terms in A + B + C, evaluation domain size.  ZKG_OK / ZKG_ERROR; what the reference's libsnark_import_pk would have
 * thrown on (libsnark_wrapper.cpp:160-168) is an error return here. */
int zkg_pk_blob_inspect(const void *pk_blob, size_t len, uint64_t out[8]);
void     zkg_crs_free(zkg_crs *crs);
/* One proof over several GPUs of ONE process: shards the H query of a resident key — the largest of the prover's
 * four multi-exponentiations, m - 1 uniformly random scalars — by points over `ndev` devices (call zkg_init_multi first; a device may be
 * listed more than once, which is how a one-GPU box rehearses the path).  Shard i keeps the per-window table of its slice on devices[i];
 * every later proof copies that slice of coefficients_for_H there (32 bytes per point, peer-to-peer), runs the shards side by side and adds
 * the partial points on the host.  Everything else of the proof stays on the key's own device.  Proof bytes are unchanged.
 * No proof of this key may be in flight during the call.                                                                         */
int zkg_crs_shard_h(zkg_crs *crs, const int *devices, int ndev);
uint32_t zkg_crs_num_variables(const zkg_crs *crs);   /* n of the resident key (the witness length zkg_groth16_prove expects) */

/* ---- Groth16 prove: r1cs_gg_ppzksnark_prover (snark.cpp:126) with the prover
 *      randomness (r, s) as explicit inputs (libsnark draws them internally).
 *      witness: n x 4 limbs Montgomery Fr = primary_input || auxiliary_input.
 *      r, s: Montgomery Fr.  proof_out: >= ZKG_PROOF_BYTES; layout = libsnark
 *      operator<<(proof) under its default flags (binary, Montgomery, compressed):
 *      g_A (34 B) || g_B (66 B) || g_C (34 B) (exported at libsnark_wrapper.cpp:170-181).
 *      check_satisfied != 0 reproduces the gate of snark.cpp:121-124 and returns
 *      ZKG_UNSATISFIED without proving.                                                */
int zkg_groth16_prove(const zkg_crs *crs, const uint64_t *witness, const uint64_t r[4],
                      const uint64_t s[4], int check_satisfied, uint8_t *proof_out, size_t *proof_len);
/* The same proof from a sparse description of the witness: tags[n] (0 = zero, 1 = one, 2 = listed) and `count` listed variables
 * as (index in 0..n-1, value as 4 Montgomery limbs).  For witness generators that know their bits (zkg_circuit_sparse_witness): the
 * host-to-device upload shrinks ~30x.  Proof bytes are identical to zkg_groth16_prove on the expanded vector.
 * Every listed index must be in range, tagged 2 and listed once (a listed VALUE may be anything, 0 and 1 included); a tag-2 variable that
 * is not listed counts as zero.  An index out of range, not tagged 2 or listed twice is ZKG_ERROR (detected on the device: each listed
 * variable's tag byte is claimed once; no proof is written). */
int zkg_groth16_prove_sparse(const zkg_crs *crs, const uint8_t *tags, const uint32_t *full_index, const uint64_t *full_values, size_t count,
                             const uint64_t r[4], const uint64_t s[4], int check_satisfied, uint8_t *proof_out, size_t *proof_len);
/* coefficients_for_H (m+1 Fr, Montgomery) of r1cs_to_qap_witness_map, for parity tests */
int zkg_qap_witness_h(const zkg_crs *crs, const uint64_t *witness, uint64_t *h_out);
/* per-stage device milliseconds of the last zkg_groth16_prove on this crs (the stages run on their own streams, so the entries
 * overlap and do not add up to the total): [0] R1CS mat-vec, [1] 7 NTTs + pointwise, [2] A / B(G1) / L over the non-bit witness
 * elements (one batched job), [3] unused, [4] B(G2) over the same elements, [5] H, [6] unused, [7] wall-clock total incl. host
 * assembly.  The flat sums over the witness elements equal to one run beside [2] and [4] on a stream of their own.               */
int zkg_prove_stage_ms(const zkg_crs *crs, float ms[8]);
/* the largest number of proofs one resident key has had in flight at once (callers on several threads share a key's prover slots: up
 * to three below m = 2^18, two at 2^18, one above) since the last call with reset != 0.  A counter, not a clock: what the concurrency
 * tests assert instead of wall-clock ratios. */
int zkg_prover_peak_in_flight(int reset);

/* ---- zklaim's credential circuit on the host: replaces protoboard + zklaim_gadget construction,
 *      generate_r1cs_constraints and generate_r1cs_witness (snark.cpp:113-118, zklaim_gadget.cpp:153-784) and
 *      zklaim_input_map (zklaim_gadget.cpp:115-150).  `ctx` is zklaim's own zklaim_ctx (include/zklaim_abi.h).           */
struct zklaim_ctx;
typedef struct zkg_circuit zkg_circuit;
/* flags: ZKG_CIRCUIT_WITH_WITNESS assigns the variables from ctx (generate_r1cs_witness);
 *        ZKG_CIRCUIT_REFERENCE_QUIRK leaves the pack_PL / pack_REF / pack_OPS packings unconstrained as the reference does
 *        (zklaim_gadget.cpp:583-699 never generates them): refvals / opsvals / plvars are then free witness variables and the
 *        proof only binds SHA256(pre) == hash.  Default (flag clear): the packings are enforced (78 constraints per payload). */
#define ZKG_CIRCUIT_WITH_WITNESS 1
#define ZKG_CIRCUIT_REFERENCE_QUIRK 2
zkg_circuit *zkg_zklaim_circuit_new(const struct zklaim_ctx *ctx, int flags);
zkg_circuit *zkg_zklaim_witness_new(const struct zklaim_ctx *ctx);   /* witness only: no constraints, no CSR (prover with a resident key) */
uint32_t zkg_circuit_num_variables(const zkg_circuit *c);
void zkg_circuit_free(zkg_circuit *c);
int zkg_circuit_r1cs(const zkg_circuit *c, zkg_r1cs *out);        /* pointers stay valid until zkg_circuit_free            */
const uint64_t *zkg_circuit_witness(const zkg_circuit *c);        /* num_variables x 4 limbs (NULL without witness)        */
int zkg_circuit_sparse_witness(const zkg_circuit *c, const uint8_t **tags, const uint32_t **full_index, const uint64_t **full_values, size_t *count);
int zkg_circuit_is_satisfied(const zkg_circuit *c);               /* pb.is_satisfied() (snark.cpp:121)                     */
long zkg_circuit_first_unsatisfied(const zkg_circuit *c);         /* index of the first violated constraint, -1 if none    */
size_t zkg_zklaim_input_map(const struct zklaim_ctx *ctx, uint64_t *out, size_t cap_elems);   /* returns the element count */

/* ---- key generation and verification.
 *      zkg_groth16_setup replaces r1cs_gg_ppzksnark_generator (snark.cpp:91): trapdoor = 5 x 4 canonical limbs
 *      (t, alpha, beta, gamma, delta) or NULL for fresh randomness; the fixed-base exponentiations run on the GPU.
 *      The blobs follow libsnark's operator<< layout for pk / vk (exported at libsnark_wrapper.cpp:122-157).
 *      zkg_groth16_verify replaces r1cs_gg_ppzksnark_verifier_strong_IC (snark.cpp:62): 0 valid, 1 invalid, 2 malformed.   */
typedef struct zkg_keypair zkg_keypair;
zkg_keypair *zkg_groth16_setup(const zkg_r1cs *cs, const uint64_t *trapdoor);
void zkg_keypair_free(zkg_keypair *kp);
const zkg_pk *zkg_keypair_pk(const zkg_keypair *kp);               /* flat arrays, valid until zkg_keypair_free              */
int zkg_keypair_swapped(const zkg_keypair *kp);                    /* 1 if swap_AB_if_beneficial exchanged A and B          */
size_t zkg_keypair_pk_blob(const zkg_keypair *kp, uint8_t *out, size_t cap);   /* returns the size needed / written       */
size_t zkg_keypair_vk_blob(const zkg_keypair *kp, uint8_t *out, size_t cap);
int zkg_groth16_verify(const uint8_t *vk_blob, size_t vk_len, const uint64_t *primary_input, size_t n_inputs,
                       const uint8_t *proof, size_t proof_len);
int zkg_pairing_probe(const uint64_t a[4], const uint64_t b[4], uint8_t out[384]);   /* e(a*G1, b*G2), for bilinearity tests */
/* test hook: Frobenius maps and the last chunk of the final exponentiation against plain square-and-multiply by q^k and by
 * the integer e (nlimbs x u32, little-endian); 0 = all agree */
int zkg_pairing_selfcheck(const uint32_t *e, int nlimbs);

/* ---- the reference's own seam (zklaim.h:257-259, libsnark_wrapper.cpp:195-276), same names and return codes, on
 *      zklaim's zklaim_ctx (include/zklaim_abi.h): the three functions zklaim.c:77-91 calls.                               */
int libsnark_trusted_setup(struct zklaim_ctx *ctx);
int libsnark_prove(struct zklaim_ctx *ctx);
int libsnark_verify(struct zklaim_ctx *ctx);
void zkg_compat_reset(void);

/* ---- known-answer hook for the device arithmetic (libff Fp_model<4,...>::mul_reduce, Fp2_model —
 *      here the generated v_mad_u64_u32 streams of csrc/mont_asm.inc).  Element-wise ON THE GPU, host pointers:
 *      field 0 = Fq, 1 = Fr (4 limbs per element), 2 = Fq2 (8 limbs);  op 0 mul, 1 add, 2 sub, 3 inverse, 4 to Montgomery form,
 *      5 from Montgomery form, 6 negate, 7 square (4 and 5: Fq / Fr only).  Inputs and outputs are Montgomery limbs except op 4's
 *      input and op 5's output (canonical); outputs are fully reduced.  b is read by ops 0-2 only.
 *      Fq only, ops 10-14: the same arithmetic on the 9 x 29-bit representation of the bucket-accumulation kernel (csrc/fq29.hip.hpp),
 *      entered and left through its conversions: 10 mul, 11 add, 12 sub, 13 a if a != b else 0 (its zero test), 14 the composite
 *      (b-a)(a-b) - (b-a)^2 - 2ab with unnormalised intermediate sums, as the mixed addition chains them, 15 the inverse of 3a
 *      (safegcd divsteps on 30-bit limbs, f29::inverse: what the batched-affine accumulation shares across a workgroup; 0 -> 0).     */
int zkg_field_op(int field, int op, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out);

/* known-answer hook for the 29-bit group law of the bucket-reduction kernels (csrc/fq29.hip.hpp, xyzz29_add_quad): on the GPU,
 * out[i] = a[i] + b[i], then `chain` rounds of x <- 2x + b[i]; points as normalised Jacobian (12 limbs), host pointers.          */
int zkg_g1_add_quad29(const uint64_t *a_jac, const uint64_t *b_jac, size_t n, int chain, uint64_t *out_jac);
/* the same for the pair form the bucket reduction uses since round 4 (xyzz29_add_pair: lane 0 of a pair holds (X, ZZ), lane 1 (Y, ZZZ)) */
int zkg_g1_add_pair29(const uint64_t *a_jac, const uint64_t *b_jac, size_t n, int chain, uint64_t *out_jac);

/* kernel-only timing hooks for bench.py (HIP events on the stream the kernels run on): average device ms per launch of the dominant kernel
 * since the last reset.  Every fourth call of an MSM entry point is timed, with all of its launches (the event records cost the stream they
 * sit on: 2 % of a 2^20-point step when every launch carries them; ZKG_KERNEL_TIMER_STRIDE=1 times every call); *launches is the launch
 * count of ALL calls since the reset, so that average x launches / calls is the kernel's time per call. */
void  zkg_timing_reset(void);
float zkg_timing_dominant_ms(int *launches);

#ifdef __cplusplus
}
#endif
#endif /* ZKG_H */
