#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_timeline
mkdir -p $OUT
for k in ${KS:-1 8}; do
  REPS=6 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/log_k$k.txt 2>&1 || exit 1
  python3 tools/proof_timeline.py $OUT/k$k/p_kernel_trace.csv > $OUT/timeline_k$k.txt
  tail -1 $OUT/timeline_k$k.txt
done
