#!/bin/bash
# raised wave priority (s_setprio 2) for the proof's critical-path kernels — matrix-vector products, transforms, pointwise step, the H query's
# accumulate / fold / reduce — against ZKG_CRIT_PRIO=0, alternating on one box; parity of the proofs first
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_crit
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_ntt.py tests/test_gpu_baseline_sizes.py tests/test_gpu_step_domain.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for tag in on off on off; do
  if [ $tag = off ]; then export ZKG_CRIT_PRIO=0; else unset ZKG_CRIT_PRIO; fi
  for k in 2 8 20 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
