#!/bin/bash
# the satisfiability check inside k_r1cs_eval (default) against the check as a kernel of its own in front of the transforms (ZKG_CHECK_KERNEL=1),
# alternating on one box; the prover's parity and refusal tests first
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_fcheck
mkdir -p $OUT
timeout -k 10 800 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_baseline_sizes.py tests/test_gpu_step_domain.py tests/test_gpu_zklaim_flow.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
ZKG_CHECK_KERNEL=1 timeout -k 10 300 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q -k "unsatisfied or real_zklaim or sparse_witness" > $OUT/tests_old.log 2>&1 || { tail -30 $OUT/tests_old.log; exit 1; }
tail -1 $OUT/tests_old.log
for tag in fused kernel fused kernel; do
  if [ $tag = kernel ]; then export ZKG_CHECK_KERNEL=1; else unset ZKG_CHECK_KERNEL; fi
  for k in 2 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
