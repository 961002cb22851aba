#!/bin/bash
# MSM parity tests, then the bench's MSM line (no extras) and a kernel profile of it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_msm_ab
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_field.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_stats'], d['roofline']['kernel_ms'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $OUT/prof.log 2>&1 || exit 1
python3 tools/kstats.py $OUT/ks/p_kernel_stats.csv | head -12
