/* sharded_proof_demo.c — ONE credential proof with its H query sharded over several devices, from plain C (SURVEY.md section 8e;
 * r1cs_gg_ppzksnark_prover at /root/reference/zklaim/snark.cpp:126 is the call this stands behind):
 *   zkg_zklaim_circuit_new -> zkg_groth16_setup -> zkg_crs_upload -> zkg_groth16_prove                       (one device)
 *   zkg_init_multi -> zkg_crs_shard_h(devices) -> zkg_groth16_prove with the same (r, s)                      (H over argv[1] shards)
 * and the two 134-byte proofs are compared byte for byte, then verified (zkg_groth16_verify).  argv[1] = number of shards: on devices
 * 0..n-1 when the box has that many GPUs, otherwise all on device 0 (the single-GPU rehearsal).  argv[2] = payloads (default 2).
 * Exit code 0 iff the proofs are identical and valid. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <openssl/sha.h>
#include "zklaim_abi.h"
#include "zkg.h"

static void add_payload(zklaim_ctx *ctx, const uint64_t attr[5], const uint64_t ref[5], const enum zklaim_op op[5], uint64_t salt) {
    zklaim_wrap_payload_ctx *n = calloc(1, sizeof *n), *cur = ctx->pl_ctx_head;
    for (int j = 0; j < 5; ++j) { memcpy(n->pl.pre + 8 * j, &attr[j], 8); n->pl.data_ref[j] = ref[j]; n->pl.data_op[j] = op[j]; }
    n->pl.salt = salt; memcpy(n->pl.pre + 40, &salt, 8);
    SHA256(n->pl.pre, sizeof n->pl.pre, n->pl.hash);
    if (!cur) ctx->pl_ctx_head = n; else { while (cur->next) cur = cur->next; cur->next = n; }
    ctx->num_of_payloads += 1;
}

int main(int argc, char **argv) {
    const int ndev = argc > 1 ? atoi(argv[1]) : 2, payloads = argc > 2 ? atoi(argv[2]) : 2;
    if (ndev < 1 || ndev > 16 || payloads < 1 || payloads > 20) return 10;
    if (zkg_init(0)) { fprintf(stderr, "zkg_init: %s\n", zkg_last_error()); return 1; }
    zklaim_ctx *ctx = calloc(1, sizeof *ctx);
    for (int i = 0; i < payloads; ++i) {
        const uint64_t attr[5] = {1990 + (uint64_t)i, 7, 42, (uint64_t)i, 5}, ref[5] = {2100, 7, 41, 0, 5};
        const enum zklaim_op op[5] = {zklaim_less, zklaim_eq, zklaim_greater, zklaim_noop, zklaim_greater_or_eq};
        add_payload(ctx, attr, ref, op, 0x5A4B0000ull + (uint64_t)i);
    }
    zkg_circuit *ck = zkg_zklaim_circuit_new(ctx, ZKG_CIRCUIT_WITH_WITNESS);
    if (!ck || !zkg_circuit_is_satisfied(ck)) { fprintf(stderr, "circuit: %s\n", zkg_last_error()); return 2; }
    zkg_r1cs cs;
    if (zkg_circuit_r1cs(ck, &cs)) return 3;
    const uint64_t trapdoor[20] = {11, 0, 0, 0, 22, 0, 0, 0, 33, 0, 0, 0, 44, 0, 0, 0, 55, 0, 0, 0};     /* fixed (t, alpha, beta, gamma, delta): a test key */
    zkg_keypair *kp = zkg_groth16_setup(&cs, trapdoor);
    if (!kp) { fprintf(stderr, "setup: %s\n", zkg_last_error()); return 4; }
    zkg_crs *crs = zkg_crs_upload(zkg_keypair_pk(kp));
    if (!crs) { fprintf(stderr, "crs_upload: %s\n", zkg_last_error()); return 5; }
    /* r, s: Montgomery Fr limbs of two fixed field elements (any values below r in Montgomery form are valid randomness for a test) */
    const uint64_t r[4] = {0x1111111111111111ull, 0x2222222222222222ull, 0x3333333333333333ull, 0x0123456789abcdefull};
    const uint64_t s[4] = {0x9999999999999999ull, 0x8888888888888888ull, 0x7777777777777777ull, 0x0fedcba987654321ull};
    const uint64_t *w = zkg_circuit_witness(ck);
    unsigned char plain[ZKG_PROOF_BYTES], sharded[ZKG_PROOF_BYTES]; size_t len1 = 0, len2 = 0;
    if (zkg_groth16_prove(crs, w, r, s, 1, plain, &len1) || len1 != ZKG_PROOF_BYTES) { fprintf(stderr, "prove: %s\n", zkg_last_error()); return 6; }
    int devs[16];
    for (int i = 0; i < ndev; ++i) devs[i] = i;
    if (zkg_init_multi(devs, ndev)) { for (int i = 0; i < ndev; ++i) devs[i] = 0; if (zkg_init_multi(devs, ndev)) { fprintf(stderr, "init_multi: %s\n", zkg_last_error()); return 7; } }
    if (zkg_crs_shard_h(crs, devs, ndev)) { fprintf(stderr, "shard_h: %s\n", zkg_last_error()); return 8; }
    if (zkg_groth16_prove(crs, w, r, s, 1, sharded, &len2) || len2 != ZKG_PROOF_BYTES) { fprintf(stderr, "sharded prove: %s\n", zkg_last_error()); return 9; }
    unsigned char *vk = NULL; size_t vk_len = zkg_keypair_vk_blob(kp, NULL, 0);
    vk = malloc(vk_len);
    if (zkg_keypair_vk_blob(kp, vk, vk_len) != vk_len) return 11;
    const int same = memcmp(plain, sharded, ZKG_PROOF_BYTES) == 0;
    const int valid = zkg_groth16_verify(vk, vk_len, w, cs.num_inputs, sharded, len2) == 0;
    printf("sharded proof demo: %d payload(s), %u constraints, H over %d shard(s) on device(s) %d..%d: proofs %s, verification %s\n", payloads, cs.num_constraints,
           ndev, devs[0], devs[ndev - 1], same ? "identical" : "DIFFER", valid ? "ok" : "FAILED");
    free(vk);
    zkg_crs_free(crs); zkg_keypair_free(kp); zkg_circuit_free(ck);
    for (zklaim_wrap_payload_ctx *p = ctx->pl_ctx_head; p;) { zklaim_wrap_payload_ctx *nx = p->next; free(p); p = nx; }
    free(ctx);
    zkg_shutdown();
    return same && valid ? 0 : 12;
}
