#!/bin/bash
# lone 2^20 transform in two passes of ten stages (bigger tiles) against the shipped three passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt_2pass.txt
: > $OUT
run() { echo "== $*" >> $OUT; for lg in 20; do env "$@" timeout -k 10 100 python3 tools/ntt_profile.py $lg 200 2>&1 | tail -1 >> $OUT || exit 1; done; }
run A=0
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=11
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=11 ZKG_NTT_XCD=1
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=10
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=10 ZKG_NTT_XCD=1
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=12
run ZKG_NTT_TILE_LOG=10
run ZKG_NTT_TILE_LOG=10 ZKG_NTT_XCD=1
run ZKG_NTT_TILE_LOG=8
cat $OUT
ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=11 ZKG_NTT_XCD=1 timeout -k 10 600 python -m pytest tests/test_gpu_ntt.py -x -q 2>&1 | tail -3
