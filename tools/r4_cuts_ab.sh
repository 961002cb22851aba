#!/bin/bash
# piece boundaries of the headline step (ZKG_MSM_CUTS, 64ths of n), same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_cuts
mkdir -p $OUT
for cuts in default 8,32 8,24 12,36 16,40 8,24,44 6,20,40 10,28; do
  tag=$(echo $cuts | tr ',' '_')
  if [ $cuts = default ]; then unset ZKG_MSM_CUTS; else export ZKG_MSM_CUTS=$cuts; fi
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 30 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }
  python3 -c "
import json; j=json.load(open('$OUT/bench_$tag.json')); print('cuts $cuts: value', j['value'], 'mean', j['ms_per_step'], 'median', j['ms_per_step_stats']['median'], 'min', j['ms_per_step_stats']['min'], 'accum', j['roofline']['kernel_ms'])"
done
