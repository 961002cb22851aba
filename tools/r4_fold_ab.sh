#!/bin/bash
# a lone table set (the prover's H query, the fixed-bases handle) folded on the 29-bit records (default) against the 32-bit fold (ZKG_FOLD_32=1), same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_fold
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_baseline_sizes.py tests/test_gpu_step_domain.py tests/test_gpu_zklaim_flow.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for tag in fold29 fold32 fold29 fold32; do
  if [ $tag = fold32 ]; then export ZKG_FOLD_32=1; else unset ZKG_FOLD_32; fi
  for k in 1 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
