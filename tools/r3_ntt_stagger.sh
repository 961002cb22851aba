#!/bin/bash
# lone 2^20 transform: cohorts of workgroups started late (ZKG_NTT_STAGGER_*) and the XCD-contiguous tile map (ZKG_NTT_XCD)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt_stagger.txt
: > $OUT
run() { echo "== $*" >> $OUT; for lg in 20; do env "$@" timeout -k 10 100 python3 tools/ntt_profile.py $lg 200 2>&1 | tail -1 >> $OUT || exit 1; done; }
run A=0
run ZKG_NTT_XCD=1
for sh in 8 3 9; do for ns in 3000 6000 10000 15000; do run ZKG_NTT_STAGGER_NS=$ns ZKG_NTT_STAGGER_SHIFT=$sh; done; done
for ns in 2000 4000 7000; do run ZKG_NTT_STAGGER_NS=$ns ZKG_NTT_STAGGER_SHIFT=8 ZKG_NTT_STAGGER_COHORTS=4; done
for ns in 2000 4000; do run ZKG_NTT_STAGGER_NS=$ns ZKG_NTT_STAGGER_SHIFT=8 ZKG_NTT_STAGGER_COHORTS=8; done
run ZKG_NTT_STAGGER_NS=6000 ZKG_NTT_STAGGER_SHIFT=8 ZKG_NTT_XCD=1
cat $OUT
ZKG_NTT_XCD=1 ZKG_NTT_STAGGER_NS=6000 timeout -k 10 600 python -m pytest tests/test_gpu_ntt.py -x -q 2>&1 | tail -3
