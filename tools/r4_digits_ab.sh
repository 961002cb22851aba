#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_digits
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for tag in new generic; do
  [ $tag = generic ] && export ZKG_DIGITS_GENERIC=1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$tag -o b -- python3 bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_$tag.json 2> $OUT/err_$tag.log || { tail -5 $OUT/err_$tag.log; exit 1; }
  echo "== $tag"; python3 tools/kstats.py $OUT/k_$tag/b_kernel_stats.csv > $OUT/kstats_$tag.txt; grep -E "k_digits|k_rx|k_class|k_order|reduce29|accum29" $OUT/kstats_$tag.txt
  python3 -c "
import json; j=json.load(open('$OUT/bench_$tag.json')); print('value', j['value'], 'median', j['ms_per_step_stats']['median'], 'resident median', j['scalars_resident']['ms_per_step']['median'])"
done
