#!/bin/bash
# the bucket reduction with additions shared by pairs of lanes (default) against round 3's quad kernel (ZKG_REDUCE_QUAD=1), same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_reduce
mkdir -p $OUT
for tag in pair quad; do
  if [ $tag = quad ]; then export ZKG_REDUCE_QUAD=1; else unset ZKG_REDUCE_QUAD; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k_$tag -o b -- python3 bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_$tag.json 2> $OUT/err_$tag.log || { tail -5 $OUT/err_$tag.log; exit 1; }
  python3 tools/kstats.py $OUT/k_$tag/b_kernel_stats.csv > $OUT/kstats_$tag.txt; grep -E "reduce29" $OUT/kstats_$tag.txt
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/bench2_$tag.json 2> $OUT/err2_$tag.log || exit 1
  python3 -c "
import json; j=json.load(open('$OUT/bench2_$tag.json')); print('$tag: value', j['value'], 'median', j['ms_per_step_stats']['median'], 'resident median', j['scalars_resident']['ms_per_step']['median'])"
  REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py 8 | tail -1
done
