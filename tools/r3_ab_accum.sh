#!/bin/bash
# A/B of the bucket accumulation on the GPU box: parity tests first, then bench.py's headline leg with the 29-bit kernel (two register
# budgets, window sizes) and with the 8 x 32-bit one (ZKG_ACCUM_32=1); then the seam: flow tests and a short sweep of the reference's protocol.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ab
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_field.py tests/test_gpu_msm.py tests/test_gpu_baseline_sizes.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -6 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }; }
run 29 X=1; run 29w3 ZKG_ACC29_WAVES=3; run 32 ZKG_ACCUM_32=1; run 29c15 ZKG_MSM_C=15; run 29c14 ZKG_MSM_C=14
python3 - <<'PY'
import json
for tag in ("29", "29w3", "32", "29c15", "29c14"):
    j = json.load(open(f"gpurun_out/r3_ab/bench_{tag}.json"))
    print(tag, "value", j["value"], "GB/s  median", j["ms_per_step_stats"]["median"], " kernel_ms", j["roofline"]["kernel_ms"], " frac", j["roofline"]["frac"])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_zklaim_flow.py tests/test_gpu_groth16.py -x -q -s > $OUT/pytest_flow.log 2>&1; rc=$?
tail -12 $OUT/pytest_flow.log
[ $rc -eq 0 ] || exit 1
ZKG_DEBUG_TIMING=1 timeout -k 10 300 python tools/zklaim_benchmark.py 20 --runs 2 > $OUT/seam_k20_dbg.csv 2> $OUT/seam_k20_dbg.err; grep -E "seam|key upload|setup\]" $OUT/seam_k20_dbg.err | tail -22
timeout -k 10 600 python tools/zklaim_benchmark.py 1 8 20 --runs 4 > $OUT/seam_sweep.csv 2> $OUT/seam_sweep.err || { tail -5 $OUT/seam_sweep.err; exit 1; }
cat $OUT/seam_sweep.csv
REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py 8 | tail -1 | cut -c1-600
REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py 37 | tail -1 | cut -c1-600
