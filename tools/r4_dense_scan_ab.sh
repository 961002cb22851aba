#!/bin/bash
# zkg_groth16_prove on a dense witness: host scan + sparse upload (default) against the dense upload (ZKG_DENSE_UPLOAD=1), same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_dense
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_baseline_sizes.py tests/test_gpu_step_domain.py tests/test_gpu_zklaim_flow.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for tag in scan dense scan dense; do
  if [ $tag = dense ]; then export ZKG_DENSE_UPLOAD=1; else unset ZKG_DENSE_UPLOAD; fi
  for k in 2 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
