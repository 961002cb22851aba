#!/bin/bash
# parity tests of the prover and the MSM, then prover timing at 1, 8 and 37 payloads
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_prove_ab2
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_zklaim_flow.py tests/test_gpu_msm.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
for k in 1 8 37; do
  REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k | tail -1 | tee $OUT/timing_k$k.txt || exit 1
done
