#!/bin/bash
# the matrix-vector stage three ways, alternating on one box: "merged" (ZKG_LONG_MERGED=1: one launch — the long rows' workgroups lead the grid, the check inside),
# "fused" (default: the check inside k_r1cs_eval, the long rows a launch of their own), "kernels" (ZKG_CHECK_KERNEL=1: eval, long, check);
# the prover's parity and refusal tests first ($SKIP_TESTS=1 skips them)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_fcheck
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 800 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_baseline_sizes.py tests/test_gpu_step_domain.py tests/test_gpu_zklaim_flow.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for e in ZKG_CHECK_KERNEL ZKG_LONG_MERGED; do
  env $e=1 timeout -k 10 300 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q -k "unsatisfied or real_zklaim or sparse_witness" > $OUT/tests_$e.log 2>&1 || { tail -30 $OUT/tests_$e.log; exit 1; }
  tail -1 $OUT/tests_$e.log
done
fi
for round in 1 2 3; do
for tag in merged fused kernels; do
  unset ZKG_CHECK_KERNEL ZKG_LONG_MERGED
  [ $tag = merged ] && export ZKG_LONG_MERGED=1
  [ $tag = kernels ] && export ZKG_CHECK_KERNEL=1
  for k in ${KS:-8 37}; do REPS=${REPS:-50} timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
done
