#!/bin/bash
# radix-4 NTT steps: carry propagation on the two added-to loads (default) against on the four stores (ZKG_NTT_NORM_STORES=1), alternating on one box;
# the transform's parity tests (2^0 ... 2^21, four variants, against the oracle) and the prover's first
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_nttnorm
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_step_domain.py tests/test_gpu_groth16.py tests/test_gpu_baseline_sizes.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for tag in loads stores loads stores; do
  if [ $tag = stores ]; then export ZKG_NTT_NORM_STORES=1; else unset ZKG_NTT_NORM_STORES; fi
  timeout -k 10 200 python3 tools/ntt_profile.py 20 200 2>/dev/null | tail -1 | sed "s/^/$tag 2^20 /"
  timeout -k 10 200 python3 tools/ntt_profile.py 18 200 2>/dev/null | tail -1 | sed "s/^/$tag 2^18 /"
  for k in 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
