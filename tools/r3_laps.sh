#!/bin/bash
# host laps of the prover (ZKG_DEBUG_TIMING=1) at several payload counts
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_laps
mkdir -p $OUT
for k in ${KS:-1 8 37}; do
  GC=0 ZKG_DEBUG_TIMING=1 timeout -k 10 200 python3 tools/prove_outliers.py $k 200 > $OUT/out_k$k.txt 2> $OUT/err_k$k.txt || exit 1
  tail -n +1 $OUT/out_k$k.txt | grep -v "proof " 
  python3 tools/prove_outliers_laps.py $OUT/err_k$k.txt | sed -n 1,2p
done
