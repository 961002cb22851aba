#!/bin/bash
# A/B on the GPU box: parity tests first, then bench.py's headline leg in the variants named below
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ab
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_msm.py tests/test_gpu_baseline_sizes.py::test_msm_2p20_all_points_vs_oracle tests/test_gpu_config5.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -6 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }; }
run split X=1; run nosplit ZKG_MSM_NO_SPLIT=1; run split_w3 ZKG_ACC29_WAVES=3; run split_red32 ZKG_REDUCE_32=1; run all32 ZKG_ACCUM_32=1
python3 - <<'PY'
import json
for tag in ("split", "nosplit", "split_w3", "split_red32", "all32"):
    j = json.load(open(f"gpurun_out/r3_ab/bench_{tag}.json"))
    print(f"{tag:12s} value {j['value']:7.3f} GB/s  median {j['ms_per_step_stats']['median']:.4f}  kernel_ms {j['roofline']['kernel_ms']:.4f} x {j['roofline']['launches']}  frac {j['roofline']['frac']}")
PY
