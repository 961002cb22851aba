#!/bin/bash
# Regenerates the artefacts under profiles/ on the GPU box (run through gpurun from the repo root); results land in gpurun_out/refresh/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/refresh
mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/r1_bench.json 2> $OUT/bench.err || exit 1
echo "bench done" && tail -c 600 $OUT/r1_bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
cp $OUT/kstats/b_kernel_stats.csv $OUT/r1_bench_kernel_stats.csv
python3 tools/kstats.py $OUT/r1_bench_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_prove -o p -- python3 tools/zklaim_prove_profile.py 8 > $OUT/prove_under_rocprof.log 2>&1 || exit 1
cp $OUT/kstats_prove/p_kernel_stats.csv $OUT/r1_prove_k8_kernel_stats.csv
bash tools/pmc_collect.sh || exit 1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout -k 10 300 python tools/zklaim_benchmark.py > $OUT/r1_zklaim_benchmark_seam_k1_20.csv 2> $OUT/seam.err || exit 1
tail -3 $OUT/r1_zklaim_benchmark_seam_k1_20.csv
