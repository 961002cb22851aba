#!/bin/bash
# rocprofv3 kernel stats of the headline bench leg and of the 8- and 37-payload proofs (round 3)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_prof
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
python3 tools/kstats.py $OUT/kstats/b_kernel_stats.csv | head -24
for k in 8 37; do
  ZKG_SERIAL_MSM=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_prove_serial_k$k -o p -- python3 tools/zklaim_prove_profile.py $k > $OUT/prove_k${k}_serial.log 2>&1 || exit 1
  echo "== prove k=$k serial"; python3 tools/kstats.py $OUT/kstats_prove_serial_k$k/p_kernel_stats.csv | head -22
done
