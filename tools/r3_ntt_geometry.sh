#!/bin/bash
# lone transforms 2^12..2^24: shipped geometry against round 2's (and the steps between them)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt_geometry.txt
: > $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_step_domain.py -x -q 2>&1 | tail -1 | tee -a $OUT
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
run() { echo "== $*" >> $OUT; for lg in 12 14 16 18 20 22 24; do env "$@" timeout -k 10 100 python3 tools/ntt_profile.py $lg 200 2>&1 | tail -1 >> $OUT || exit 1; done; }
run SHIPPED=1
run ZKG_NTT_MAX_R=8 ZKG_NTT_TILE_LOG=9 ZKG_NTT_RADIX2=1 ZKG_NTT_XCD=0
run ZKG_NTT_RADIX2=1
run ZKG_NTT_XCD=0
run ZKG_NTT_MAX_R=8
cat $OUT
