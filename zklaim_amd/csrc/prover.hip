// prover.hip — placeholder (Groth16 pipeline lands next)
#include "common.hpp"
#include "../../include/zkg.h"
using namespace zk;
extern "C" {
zkg_crs *zkg_crs_upload(const zkg_pk *) { set_error("not implemented"); return nullptr; }
void zkg_crs_free(zkg_crs *) {}
int zkg_groth16_prove(const zkg_crs *, const uint64_t *, const uint64_t *, const uint64_t *, int, uint8_t *, size_t *) { return ZKG_ERROR; }
int zkg_qap_witness_h(const zkg_crs *, const uint64_t *, uint64_t *) { return ZKG_ERROR; }
int zkg_prove_stage_ms(const zkg_crs *, float *) { return ZKG_ERROR; }
}
