#!/bin/bash
# A/B on the GPU box: the headline MSM with and without the batched-affine levels (ZKG_ACCUM_BA), then a kernel trace of the best variant
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_ba
mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }; }
run base X=1
for l in 1 2 3 4; do run ba$l ZKG_ACCUM_BA=$l; done
run ba3_k8 ZKG_ACCUM_BA=3 ZKG_BA_K=8; run ba3_k32 ZKG_ACCUM_BA=3 ZKG_BA_K=32; run ba2_k32 ZKG_ACCUM_BA=2 ZKG_BA_K=32
python3 - <<'PY'
import json
for tag in ("base", "ba1", "ba2", "ba3", "ba4", "ba3_k8", "ba3_k32", "ba2_k32"):
    j = json.load(open(f"gpurun_out/r4_ba/bench_{tag}.json"))
    print(f"{tag:10s} value {j['value']:7.3f} GB/s  median {j['ms_per_step_stats']['median']:.4f}  accumulation {j['roofline']['kernel_ms']:.4f} ms x {j['roofline']['launches']}")
PY
export ZKG_ACCUM_BA=3
rocprofv3 --kernel-trace --stats -d $OUT/prof_ba3 -o ba3 -- python3 bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_ba3_prof.json 2> $OUT/prof.err || tail -5 $OUT/prof.err
f=$(find $OUT/prof_ba3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/ba3_kernel_stats.csv && head -14 $OUT/ba3_kernel_stats.csv | cut -c1-200
