#!/bin/bash
# radix-4 steps (k_ntt_pass29_r4) against the radix-2 pass: parity tests, then lone transforms
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt_r4.txt
: > $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_step_domain.py -x -q 2>&1 | tail -3 | tee -a $OUT
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
run() { echo "== $*" >> $OUT; for lg in ${LGS:-18 20 22}; do env "$@" timeout -k 10 100 python3 tools/ntt_profile.py $lg 200 2>&1 | tail -1 >> $OUT || exit 1; done; }
run A=0
run ZKG_NTT_RADIX2=1
run ZKG_NTT_XCD=1
run ZKG_NTT_TILE_LOG=10
run ZKG_NTT_MAX_R=10 ZKG_NTT_TILE_LOG=10 ZKG_NTT_XCD=1
cat $OUT
