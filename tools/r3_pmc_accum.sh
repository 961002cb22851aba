#!/bin/bash
# SQ counters of the bucket accumulation (own --pmc pass, kernel-trace only beside it)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_pmc
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - $OUT/sq <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "accum29" in k or "reduce29" in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
