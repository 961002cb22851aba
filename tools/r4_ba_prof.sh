#!/bin/bash
# kernel trace of the headline MSM under the batched-affine levels (ZKG_ACCUM_BA = $1, default 3)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=${1:-3}
OUT=gpurun_out/r4_ba
mkdir -p $OUT
export ZKG_ACCUM_BA=$L
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ba$L -o b -- python3 bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_ba${L}_prof.json 2> $OUT/prof.err || { tail -5 $OUT/prof.err; exit 1; }
cp $OUT/prof_ba$L/b_kernel_stats.csv $OUT/ba${L}_kernel_stats.csv
python3 tools/kstats.py $OUT/ba${L}_kernel_stats.csv | head -16
