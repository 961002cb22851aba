#!/bin/bash
# kernel timeline of one sparse-witness proof at $1 payloads (default 8), rocprofv3 --kernel-trace over tools/zklaim_prove_profile.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-8}
OUT=gpurun_out/r4_ptl
mkdir -p $OUT
REPS=6 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/k$K -o t -- python3 tools/zklaim_prove_profile.py $K > $OUT/prof_k$K.txt 2> $OUT/err_k$K.log || { tail -5 $OUT/err_k$K.log; exit 1; }
python3 tools/proof_timeline.py $(ls $OUT/k$K/*kernel_trace.csv | head -1) > $OUT/timeline_k$K.txt
tail -1 $OUT/prof_k$K.txt; tail -1 $OUT/timeline_k$K.txt
