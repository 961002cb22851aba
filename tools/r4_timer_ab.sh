#!/bin/bash
# what the kernel timer's two event records per accumulation launch cost the headline step (ZKG_KERNEL_TIMER=0 drops them), alternating on one box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_timer
mkdir -p $OUT
# fourth: every fourth call timed (default); off: none (ZKG_KERNEL_TIMER=0); all: every call (ZKG_KERNEL_TIMER_STRIDE=1)
for tag in fourth all off fourth all off; do
  unset ZKG_KERNEL_TIMER ZKG_KERNEL_TIMER_STRIDE
  [ $tag = off ] && export ZKG_KERNEL_TIMER=0
  [ $tag = all ] && export ZKG_KERNEL_TIMER_STRIDE=1
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 40 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }
  python3 -c "
import json; j=json.load(open('$OUT/bench_$tag.json')); print('timer $tag: value', j['value'], 'mean', j['ms_per_step'], 'median', j['ms_per_step_stats']['median'], 'min', j['ms_per_step_stats']['min'], 'accum', j['roofline']['kernel_ms'])"
done
