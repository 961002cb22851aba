#!/bin/bash
# prover timing (8 and 37 payloads) under the NTT variants
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_prove_ntt_ab.txt
: > $OUT
run() { echo "== $*" >> $OUT; for k in 8 37; do env "$@" REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>&1 | tail -1 >> $OUT || exit 1; done; }
run A=0
run ZKG_NTT_RADIX2=1
run ZKG_NTT_XCD=1
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=10 ZKG_NTT_XCD=1
run ZKG_NTT_RADIX2=1 ZKG_NTT_XCD=1
cat $OUT
