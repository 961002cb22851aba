#!/bin/bash
# Collects HBM traffic counters for bench.py's kernels on the GPU box (run through gpurun from the repo root):
#   separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE (TCC slots do not fit both; --pmc never combined with other trace domains),
#   plus the calibration kernels of tools/pmc_calib.hip in the same passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o $OUT/pmc_calib 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- ./$OUT/pmc_calib > $OUT/calib_$c.txt 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/bench_$c -- python bench.py --no-cpu-baseline --no-extras --headline-only --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err || exit 1
done
python tools/pmc_summary.py $OUT profiles/pmc_traffic.json
