#!/bin/bash
# the headline step (scalars uploaded inside the call) with 1..6 pieces, same box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_pieces
mkdir -p $OUT
for p in 1 2 3 4 5 6; do
  ZKG_MSM_PIECES=$p timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_p$p.json 2> $OUT/bench_p$p.err || { tail -5 $OUT/bench_p$p.err; exit 1; }
done
python3 - <<'PY'
import json
for p in range(1, 7):
    j = json.load(open(f"gpurun_out/r4_pieces/bench_p{p}.json"))
    r = j["scalars_resident"]
    print(f"pieces {p}: value {j['value']:7.3f} GB/s  step mean {j['ms_per_step']:.4f} median {j['ms_per_step_stats']['median']:.4f} ms   accumulation {j['roofline']['kernel_ms']:.4f} ms/step"
          f"   | resident {r['ms_per_step']['median']:.4f} ms   one upload + resident call {r['one_upload_then_resident_call']['ms_per_step']['median']:.4f} ms  same {r['same_result']}")
PY
