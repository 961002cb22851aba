#!/bin/bash
# prover A/B: parity tests of the prover, then timing at 8 and 37 payloads and a serial-mode kernel profile
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_prove_ab
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_zklaim_flow.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
for k in 8 37; do
  REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k | tail -1 | tee $OUT/timing_k$k.txt || exit 1
done
for k in ${PROFILE_KS:-37}; do
  ZKG_SERIAL_MSM=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_serial_k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/serial_k$k.log 2>&1 || exit 1
  python3 tools/kstats.py $OUT/ks_serial_k$k/p_kernel_stats.csv 24 2>/dev/null || true
done
