#!/bin/bash
# Regenerates the round-3 artefacts under profiles/ on the GPU box (run through gpurun from the repo root); results land in gpurun_out/refresh/
# and are copied into profiles/ by hand afterwards.  Every rocprofv3 command has the program itself directly after `--`; --pmc passes are
# separate from --kernel-trace --stats passes and from each other (tools/pmc_collect.sh).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/refresh
mkdir -p $OUT
# counters first: bench.py reports roofline.traffic only from a pmc_traffic.json collected on the kernel sources it runs
bash tools/pmc_collect.sh > $OUT/pmc_collect.log 2>&1 || { tail -5 $OUT/pmc_collect.log; exit 1; }
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
echo "pmc done"
timeout -k 10 700 python bench.py > $OUT/r3_bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
# headline MSM: per-kernel durations of the same command (no CPU legs, no extras)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > $OUT/r3_bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
cp $OUT/kstats/b_kernel_stats.csv $OUT/r3_bench_kernel_stats.csv
# NTT 2^20 alone (BASELINE configs[2])
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_ntt -o n -- python3 tools/ntt_profile.py 20 50 > $OUT/ntt_under_rocprof.log 2>&1 || exit 1
cp $OUT/kstats_ntt/n_kernel_stats.csv $OUT/r3_ntt_2p20_kernel_stats.csv
echo "kernel stats done"
# the prover, 8 payloads (m = 2^18) and 37 payloads (m = 2^20): concurrent (as shipped) and one multi-exponentiation at a time
for k in 8 37; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_prove_k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/prove_k${k}_under_rocprof.log 2>&1 || exit 1
  cp $OUT/kstats_prove_k$k/p_kernel_stats.csv $OUT/r3_prove_k${k}_kernel_stats.csv
  ZKG_SERIAL_MSM=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_prove_serial_k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/prove_k${k}_serial_under_rocprof.log 2>&1 || exit 1
  cp $OUT/kstats_prove_serial_k$k/p_kernel_stats.csv $OUT/r3_prove_k${k}_serial_kernel_stats.csv
  REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k | tail -1 > $OUT/r3_prove_k${k}_timing.txt
  GC=0 timeout -k 10 300 python3 tools/prove_outliers.py $k 1000 > $OUT/r3_prove_k${k}_latency_tail.txt 2>&1
done
echo "prove profiles done"
# the reference's benchmark protocol through the seam: k = 1..20 payloads, RUNS = 30 (main_benchmark.c:175-182)
timeout -k 10 1000 python tools/zklaim_benchmark.py --runs 30 > $OUT/r3_zklaim_benchmark_seam_k1_20_runs30.csv 2> $OUT/seam.err || exit 1
tail -3 $OUT/r3_zklaim_benchmark_seam_k1_20_runs30.csv
