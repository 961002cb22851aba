/* seam_demo.c — a C caller of the reference's seam, linked against libzkg.so: the issuer -> prover -> verifier call sequence
 * of /root/reference/zklaim/main.c:31-252 without the Ed25519 signature steps (libgcrypt, outside the path):
 *   zklaim_add_pl + zklaim_hash_ctx   -> payload list with SHA-256(pre)      (zklaim.c:93-131)
 *   zklaim_trusted_setup              -> libsnark_trusted_setup(ctx)         (zklaim.c:89-91)
 *   zklaim_proof_generate             -> libsnark_prove(ctx)                 (zklaim.c:77-80)
 *   zklaim_proof_verify               -> libsnark_verify(ctx)                (zklaim.c:82-87)
 * Prints one line per step; exit code 0 iff every step behaved as the reference's tests expect. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <openssl/sha.h>
#include "zklaim_abi.h"

int libsnark_trusted_setup(zklaim_ctx *);
int libsnark_prove(zklaim_ctx *);
int libsnark_verify(zklaim_ctx *);

static void add_payload(zklaim_ctx *ctx, const uint64_t attr[5], const uint64_t ref[5], const enum zklaim_op op[5], uint64_t salt) {
    zklaim_wrap_payload_ctx *n = calloc(1, sizeof *n), *cur = ctx->pl_ctx_head;
    for (int j = 0; j < 5; ++j) { memcpy(n->pl.pre + 8 * j, &attr[j], 8); n->pl.data_ref[j] = ref[j]; n->pl.data_op[j] = op[j]; }
    n->pl.salt = salt; memcpy(n->pl.pre + 40, &salt, 8);
    SHA256(n->pl.pre, sizeof n->pl.pre, n->pl.hash);
    if (!cur) ctx->pl_ctx_head = n; else { while (cur->next) cur = cur->next; cur->next = n; }
    ctx->num_of_payloads += 1;
}

int main(void) {
    zklaim_ctx *ctx = calloc(1, sizeof *ctx);
    const uint64_t attr[5] = {1994, 7, 42, 0, 5}, ref[5] = {2000, 7, 41, 0, 5};
    const enum zklaim_op op[5] = {zklaim_less, zklaim_eq, zklaim_greater, zklaim_noop, zklaim_greater_or_eq};
    add_payload(ctx, attr, ref, op, 0x1122334455667788ull);
    int rc = libsnark_trusted_setup(ctx);
    printf("trusted_setup rc=%d pk=%zu B vk=%zu B\n", rc, ctx->pk_size, ctx->vk_size);
    if (rc) return 1;
    rc = libsnark_prove(ctx);
    printf("prove rc=%d proof=%zu B\n", rc, ctx->proof_size);
    if (rc || ctx->proof_size != 134) return 2;
    rc = libsnark_verify(ctx);
    printf("verify rc=%d\n", rc);
    if (rc) return 3;
    memset(ctx->pl_ctx_head->pl.pre, 0, 48); ctx->pl_ctx_head->pl.salt = 0;                 /* zklaim_clear_pres: the verifier never sees pre */
    if (libsnark_verify(ctx)) return 4;
    ctx->pl_ctx_head->pl.data_ref[0] = 1990;                                                   /* forged claim: 1994 < 1990 */
    rc = libsnark_verify(ctx);
    printf("verify forged reference rc=%d (must be non-zero)\n", rc);
    if (!rc) return 5;
    free(ctx->pk); free(ctx->vk); free(ctx->proof); free(ctx->pl_ctx_head); free(ctx);
    puts("seam demo ok");
    return 0;
}
