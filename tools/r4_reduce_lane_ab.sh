#!/bin/bash
# the large bucket reduction with its first phase on single lanes (k_bucket_reduce29l, default) against the pair kernel (ZKG_REDUCE_PAIR=1), same box:
# parity first, then kernel durations under rocprofv3 and the bench's headline / resident medians
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_redlane
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_baseline_sizes.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for tag in lane pair lane pair; do
  if [ $tag = pair ]; then export ZKG_REDUCE_PAIR=1; else unset ZKG_REDUCE_PAIR; fi
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/err_$tag.log || { tail -5 $OUT/err_$tag.log; exit 1; }
  python3 -c "
import json; j=json.load(open('$OUT/bench_$tag.json')); print('$tag: value', j['value'], 'median', j['ms_per_step_stats']['median'], 'resident median', j['scalars_resident']['ms_per_step']['median'])"
done
for tag in lane pair; do
  if [ $tag = pair ]; then export ZKG_REDUCE_PAIR=1; else unset ZKG_REDUCE_PAIR; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$tag -o b -- python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 10 --warmup 3 > $OUT/benchp_$tag.json 2> $OUT/errp_$tag.log || { tail -5 $OUT/errp_$tag.log; exit 1; }
  python3 tools/kstats.py $OUT/k_$tag/b_kernel_stats.csv | grep -E "reduce29"
done
