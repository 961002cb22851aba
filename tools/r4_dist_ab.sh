#!/bin/bash
# the headline step as a plain process and as one rank under torch.distributed.run, with more hardware queues, same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_dist
mkdir -p $OUT
show() { python3 -c "
import json; j=json.load(open('$1')); print('$2: value', j['value'], 'mean', j['ms_per_step'], 'median', j['ms_per_step_stats']['median'], 'p95', j['ms_per_step_stats']['p95'], 'accum', j['roofline']['kernel_ms'])"; }
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 20 --warmup 5 > $OUT/plain_$q.json 2> $OUT/plain.err || exit 1
  show $OUT/plain_$q.json "plain, queues $q"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --no-extras --no-cpu-baseline --headline-only --steps 20 --warmup 5 > $OUT/torchrun_$q.json 2> $OUT/torchrun.err || { tail -5 $OUT/torchrun.err; exit 1; }
  show $OUT/torchrun_$q.json "torchrun, queues $q"
done
