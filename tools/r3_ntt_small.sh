#!/bin/bash
# lone transforms 2^15..2^19 under the geometry knobs
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt_small.txt
: > $OUT
run() { echo "== $*" >> $OUT; for lg in 15 16 17 18 19; do env "$@" timeout -k 10 100 python3 tools/ntt_profile.py $lg 300 2>&1 | tail -1 >> $OUT || exit 1; done; }
run SHIPPED=1
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=9 ZKG_NTT_RADIX2=0
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=9 ZKG_NTT_RADIX2=1
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=10 ZKG_NTT_RADIX2=1
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=10 ZKG_NTT_RADIX2=0
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=8 ZKG_NTT_RADIX2=1
cat $OUT
