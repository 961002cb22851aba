#!/bin/bash
# 64-byte packed base records: parity (MSM, batched-affine variant, prover), then the bench's MSM legs, the H query's accumulation inside a 37-payload proof
# (kernel averages under rocprofv3) and the proofs' medians
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_rec64
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_field.py tests/test_gpu_groth16.py tests/test_gpu_baseline_sizes.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 -c "
import json; j=json.load(open('$OUT/bench.json')); r=j['scalars_resident']; print('value', j['value'], 'median', j['ms_per_step_stats']['median'], 'accum in step', j['roofline']['kernel_ms'], '| resident median', r['ms_per_step']['median'], 'accum alone', r['accumulation_kernel']['kernel_ms'])"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k37 -o p -- python3 tools/zklaim_prove_profile.py 37 > $OUT/prove37_rocprof.log 2>&1 || exit 1
python3 tools/kstats.py $OUT/k37/p_kernel_stats.csv | grep -E "accum29|k_bases_to29"
for k in 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1; done
