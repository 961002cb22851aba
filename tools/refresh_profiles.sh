#!/bin/bash
# Regenerates the round's artefacts under profiles/ on the GPU box (run through gpurun from the repo root): `bash tools/refresh_profiles.sh [a|b]`.
# Part a: counters, bench line, kernel stats, SQ counters, NTT, prove profiles.  Part b: the reference's benchmark protocol through the seam
# (k = 1..20, RUNS = 30) — a call of its own because of its length.  Results land in gpurun_out/refresh/ and are copied into profiles/ by hand.
# Every rocprofv3 command has the program itself directly after `--`; --pmc passes are separate from --kernel-trace --stats passes and from
# each other (tools/pmc_collect.sh).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=r4
OUT=gpurun_out/refresh
mkdir -p $OUT
if [ "${1:-a}" = "b" ]; then
  timeout -k 10 1100 python tools/zklaim_benchmark.py --runs 30 > $OUT/${R}_zklaim_benchmark_seam_k1_20_runs30.csv 2> $OUT/seam.err || { tail -5 $OUT/seam.err; exit 1; }
  tail -3 $OUT/${R}_zklaim_benchmark_seam_k1_20_runs30.csv
  exit 0
fi
# counters first: bench.py reports roofline.traffic only from a pmc_traffic.json collected on the kernel sources it runs
bash tools/pmc_collect.sh > $OUT/pmc_collect.log 2>&1 || { tail -5 $OUT/pmc_collect.log; exit 1; }
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
echo "pmc done"
timeout -k 10 700 python bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
# headline MSM: per-kernel durations of the same command (no CPU legs, no extras)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > $OUT/${R}_bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
cp $OUT/kstats/b_kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
# ... and of the headline step alone (no resident-scalars legs: every launch belongs to a piece-wise step)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_h -o b -- python3 bench.py --no-cpu-baseline --no-extras --headline-only --steps 10 > $OUT/${R}_bench_headline_only_under_rocprof.json 2> $OUT/rocprof_h.err || exit 1
cp $OUT/kstats_h/b_kernel_stats.csv $OUT/${R}_bench_headline_only_kernel_stats.csv
# SQ counters of the accumulation and the reduction (own --pmc pass)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/sq_bench.json 2> $OUT/sq.err || { tail -5 $OUT/sq.err; exit 1; }
python3 - $OUT/sq <<'PY' > $OUT/${R}_accum_reduce_sq_counters.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ counters per launch (averages over the launches of `bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1`: the piece-wise headline steps — one accumulation launch per piece (four), beside the next piece's sort — and the resident-scalars legs — one launch, alone)")
for k, v in sorted(agg.items()):
    if "accum29" in k or "reduce29" in k or "k_rx" in k or "k_digits" in k:
        wc = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"])
        print(f"{k:34s} launches {len(v['SQ_WAVES']):3d}  waves {sum(v['SQ_WAVES']) / len(v['SQ_WAVES']):8.0f}  VALU insts {sum(v['SQ_INSTS_VALU']) / len(v['SQ_INSTS_VALU']) / 1e6:8.2f} M  "
              f"of wave cycles: issuing VALU {100 * sum(v['SQ_ACTIVE_INST_VALU']) / len(v['SQ_ACTIVE_INST_VALU']) / wc:5.1f} %  waiting (any) {100 * sum(v['SQ_WAIT_ANY']) / len(v['SQ_WAIT_ANY']) / wc:5.1f} %  "
              f"ready-not-issued {100 * sum(v['SQ_WAIT_INST_ANY']) / len(v['SQ_WAIT_INST_ANY']) / wc:5.1f} %")
PY
cat $OUT/${R}_accum_reduce_sq_counters.txt
# NTT 2^20 alone (BASELINE configs[2])
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_ntt -o n -- python3 tools/ntt_profile.py 20 50 > $OUT/ntt_under_rocprof.log 2>&1 || exit 1
cp $OUT/kstats_ntt/n_kernel_stats.csv $OUT/${R}_ntt_2p20_kernel_stats.csv
echo "kernel stats done"
# the prover, 8 payloads (m = 2^18) and 37 payloads (m = 2^20): concurrent (as shipped) and one multi-exponentiation at a time
for k in 8 37; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_prove_k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/prove_k${k}_under_rocprof.log 2>&1 || exit 1
  cp $OUT/kstats_prove_k$k/p_kernel_stats.csv $OUT/${R}_prove_k${k}_kernel_stats.csv
  ZKG_SERIAL_MSM=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_prove_serial_k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/prove_k${k}_serial_under_rocprof.log 2>&1 || exit 1
  cp $OUT/kstats_prove_serial_k$k/p_kernel_stats.csv $OUT/${R}_prove_k${k}_serial_kernel_stats.csv
  REPS=30 timeout -k 10 200 python3 tools/zklaim_prove_profile.py $k | tail -1 > $OUT/${R}_prove_k${k}_timing.txt
  GC=0 timeout -k 10 300 python3 tools/prove_outliers.py $k 1000 > $OUT/${R}_prove_k${k}_latency_tail.txt 2>&1
done
echo "prove profiles done"
