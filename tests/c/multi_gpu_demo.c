/* multi_gpu_demo.c — a plain C caller of the several-GPUs-in-one-process entry points of include/zkg.h:
 *   zkg_init_multi -> zkg_msm_g1_shards_upload -> zkg_msm_g1_multi -> zkg_g1_sum of the partials == the result == zkg_msm_g1.
 * argv[1] = number of shards (devices 0..ndev-1 when that many GPUs are visible, otherwise every shard on device 0 — the single-GPU
 * rehearsal).  Bases are multiples of the generator made with zkg_g1_fixed_base_dev-free host arithmetic: the generator itself repeated,
 * scalars i + 1, so the expected point is (sum (i+1)) * G — checked against zkg_msm_g1 on the same inputs.  Exit code 0 iff all agree. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "zkg.h"

int main(int argc, char **argv) {
    int ndev = argc > 1 ? atoi(argv[1]) : 3;
    if (ndev < 1 || ndev > 16) return 10;
    int devs[16];
    for (int i = 0; i < ndev; ++i) devs[i] = 0;
    if (zkg_init(0)) { fprintf(stderr, "zkg_init: %s\n", zkg_last_error()); return 1; }
    {   /* use distinct devices when the box has them */
        int ok = 1; int distinct[16];
        for (int i = 0; i < ndev; ++i) distinct[i] = i;
        if (zkg_init_multi(distinct, ndev) == 0) memcpy(devs, distinct, sizeof(int) * (size_t)ndev); else ok = 0;
        if (!ok && zkg_init_multi(devs, ndev)) { fprintf(stderr, "zkg_init_multi: %s\n", zkg_last_error()); return 2; }
    }
    const size_t n = 5000;
    /* G1 generator (1, 2) in Montgomery limbs (include/zkg.h conventions) */
    const uint64_t gen[8] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full,
                             0xa6ba871b8b1e1b3aull, 0x14f1d651eb8e167bull, 0xccdd46def0f28c58ull, 0x1c14ef83340fbe5eull};
    uint64_t *bases = malloc(n * 64), *scalars = calloc(n, 32);
    for (size_t i = 0; i < n; ++i) { memcpy(bases + 8 * i, gen, 64); scalars[4 * i] = i + 1; }
    zkg_msm_shards *sh = zkg_msm_g1_shards_upload(bases, n, devs, ndev);
    if (!sh) { fprintf(stderr, "shards_upload: %s\n", zkg_last_error()); return 3; }
    size_t pts = 0;
    if (zkg_msm_g1_shards_count(sh, &pts) != (size_t)ndev || pts != n) return 4;
    uint64_t out[12], single[12], sum[12], *parts = malloc((size_t)ndev * 96);
    if (zkg_msm_g1_multi(sh, scalars, out, parts)) { fprintf(stderr, "msm_multi: %s\n", zkg_last_error()); return 5; }
    if (zkg_msm_g1(bases, scalars, n, single)) return 6;
    if (zkg_g1_sum(parts, (size_t)ndev, sum)) return 7;
    /* (sum_{i=1..n} i) * G through a one-point MSM */
    uint64_t tot[4] = {(uint64_t)n * (n + 1) / 2, 0, 0, 0}, expect[12];
    if (zkg_msm_g1(gen, tot, 1, expect)) return 8;
    int ok = !memcmp(out, single, 96) && !memcmp(out, sum, 96) && !memcmp(out, expect, 96);
    printf("multi-GPU C demo: %d shard(s) on device(s) %d..%d, %zu points: %s\n", ndev, devs[0], devs[ndev - 1], n, ok ? "ok" : "MISMATCH");
    zkg_msm_g1_shards_free(sh);
    free(bases); free(scalars); free(parts);
    zkg_shutdown();
    return ok ? 0 : 9;
}
