/*
 * zklaim_abi.h — the C structures zklaim's front-end hands across the seam
 *     int libsnark_trusted_setup(zklaim_ctx*), libsnark_prove(zklaim_ctx*), libsnark_verify(zklaim_ctx*)
 * (declared /root/reference/zklaim/zklaim.h:257-259, defined zklaim/libsnark_wrapper.cpp:195,218,252).
 * Layouts follow zklaim.h:38-107 field for field (LP64: sizeof(zklaim_payload) == 160, sizeof(zklaim_ctx) == 160) so that a
 * zklaim_ctx built by the reference's zklaim.c can be passed to libzkg.so unchanged.  Only the data layout is restated here;
 * the reference header also pulls in libgcrypt / OpenSSL types that this path never touches.
 */
#ifndef ZKLAIM_ABI_H
#define ZKLAIM_ABI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef ZKLAIM_OK
#define ZKLAIM_OK 0                      /* zklaim.h:38-42 */
#define ZKLAIM_ERROR 1
#define ZKLAIM_INVALID_SIGNATURE 2
#define ZKLAIM_INVALID_PROOF 3
#define ZKLAIM_MAX_PAYLOAD_ATTRIBUTES 5
#endif

enum zklaim_op {                         /* zklaim.h:45-53 */
    zklaim_less = 1, zklaim_less_or_eq = 3, zklaim_eq = 2, zklaim_greater_or_eq = 10,
    zklaim_greater = 8, zklaim_not_eq = 9, zklaim_noop = 99
};

typedef struct zklaim_payload {          /* zklaim.h:64-71 */
    uint64_t data_ref[ZKLAIM_MAX_PAYLOAD_ATTRIBUTES];   /* public reference values the attributes are compared with */
    enum zklaim_op data_op[ZKLAIM_MAX_PAYLOAD_ATTRIBUTES];
    uint64_t salt;
    unsigned char hash[32];              /* SHA-256 of pre (zklaim.c:114-121) */
    uint8_t priv;
    unsigned char pre[48];               /* 5 x u64 attributes || u64 salt; zeroed when public */
} zklaim_payload;

typedef struct zklaim_wrap_payload_ctx { /* zklaim.h:82-85 */
    struct zklaim_wrap_payload_ctx *next;
    zklaim_payload pl;
} zklaim_wrap_payload_ctx;

typedef struct zklaim_ctx {              /* zklaim.h:96-107 */
    size_t num_of_payloads;
    zklaim_wrap_payload_ctx *pl_ctx_head;
    size_t pk_size;  unsigned char *pk;      /* malloc'd by the callee, freed by zklaim_ctx_free (zklaim.c:57-72) */
    size_t vk_size;  unsigned char *vk;
    size_t proof_size; unsigned char *proof;
    unsigned char pub_key[32];
    unsigned char signature[64];
} zklaim_ctx;

#ifdef __cplusplus
}
#endif
#endif
