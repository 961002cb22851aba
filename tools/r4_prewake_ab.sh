#!/bin/bash
# the host pool woken when the last accumulation has finished (ZKG_POOL_PREWAKE=1) against woken with the work (default), alternating on one box:
# the pool's microbenchmark, the headline / resident steps with the host tail's time, and proofs at 8 and 37 payloads
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_prewake
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pool_wake_bench.hip -o /tmp/pool_wake_bench -Lzklaim_amd -lzkg -Wl,-rpath,$PWD/zklaim_amd 2> $OUT/build.err && ZKG_POOL_PREWAKE=1 /tmp/pool_wake_bench
timeout -k 10 300 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "host_scalars or vs_oracle" > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
for tag in pre none pre none; do
  if [ $tag = pre ]; then export ZKG_POOL_PREWAKE=1; else unset ZKG_POOL_PREWAKE; fi
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/err_$tag.log || { tail -5 $OUT/err_$tag.log; exit 1; }
  python3 -c "
import json; j=json.load(open('$OUT/bench_$tag.json')); print('$tag: value', j['value'], 'median', j['ms_per_step_stats']['median'], 'min', j['ms_per_step_stats']['min'], 'resident median', j['scalars_resident']['ms_per_step']['median'])"
  ZKG_DEBUG_TIMING=1 timeout -k 10 200 python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 6 --warmup 3 2>&1 | grep -i "job finish" | tail -2 | sed "s/^/$tag /"
  for k in 8 37; do REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k 2>/dev/null | tail -1 | sed "s/^/$tag /"; done
done
