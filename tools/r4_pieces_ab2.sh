#!/bin/bash
# piece-wise headline step: parity first, then one / two streams x piece counts, same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_pieces2
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "host_scalars or vs_oracle or adversarial" > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail -5 $OUT/bench_$tag.err; exit 1; }; }
run one_p3 ZKG_MSM_PIECES_ONE_STREAM=1 ZKG_MSM_PIECES=3
for p in 2 3 4 5; do run two_p$p ZKG_MSM_PIECES=$p; done
python3 - <<'PY'
import json
for tag in ("one_p3", "two_p2", "two_p3", "two_p4", "two_p5"):
    j = json.load(open(f"gpurun_out/r4_pieces2/bench_{tag}.json"))
    r = j["scalars_resident"]
    print(f"{tag}: value {j['value']:7.3f} GB/s  step mean {j['ms_per_step']:.4f} median {j['ms_per_step_stats']['median']:.4f} ms   accumulation {j['roofline']['kernel_ms']:.4f} ms/step"
          f"   | resident {r['ms_per_step']['median']:.4f} ms   one upload + resident call {r['one_upload_then_resident_call']['ms_per_step']['median']:.4f} ms  same {r['same_result']}")
PY
