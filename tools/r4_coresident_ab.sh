#!/bin/bash
# accumulation capped at 192 registers + 512-thread sort workgroups (+ s_setprio in the sort kernels): do the next piece's sorts run beside
# the accumulation?  parity first, then the headline step over sort workgroup size x piece boundaries (ZKG_MSM_CUTS, 64ths of n), then a timeline
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_cores
mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_msm.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for wg in ${WGS:-512 1024}; do
for cuts in ${CUTS:-default 8,32 8,24,44 8,16,32 6,14,26,42 4,12,24,40}; do
  tag=${wg}_$(echo $cuts | tr ',' '_')
  export ZKG_SORT_WG=$wg
  if [ $cuts = default ]; then unset ZKG_MSM_CUTS; else export ZKG_MSM_CUTS=$cuts; fi
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 30 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }
  python3 -c "
import json; j=json.load(open('$OUT/bench_$tag.json')); print('wg $wg cuts $cuts: value', j['value'], 'mean', j['ms_per_step'], 'median', j['ms_per_step_stats']['median'], 'min', j['ms_per_step_stats']['min'], 'accum', j['roofline']['kernel_ms'])"
done
done
unset ZKG_SORT_WG ZKG_MSM_CUTS
[ -n "$TIMELINE_CUTS" ] && export ZKG_MSM_CUTS=$TIMELINE_CUTS
bash tools/r4_msm_timeline.sh 3 > $OUT/timeline.txt 2>&1; tail -60 $OUT/timeline.txt
