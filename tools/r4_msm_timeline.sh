#!/bin/bash
# kernel + copy timeline of one headline step (ZKG_MSM_PIECES = $1, default 3)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=${1:-3}
OUT=gpurun_out/r4_timeline
mkdir -p $OUT
export ZKG_MSM_PIECES=$P
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t$P -o t -- python3 bench.py --no-extras --no-cpu-baseline --headline-only --steps 4 --warmup 3 > $OUT/bench_p$P.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - $OUT/t$P <<'PY'
import csv, sys, glob
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], "q%s" % r.get("Queue_Id", "")))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
ev.sort()
# the last step: from the last H2D copy group backwards — find the last k_bucket_reduce29 and the step's first event after the previous reduce
reds = [i for i, e in enumerate(ev) if "k_bucket_reduce29" in e[2]]
lo = reds[-2] + 1 if len(reds) > 1 else 0
hi = reds[-1]
t0 = ev[lo][0]
prev_end = t0
for s, e, n, q in ev[lo:hi + 3]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {n} {q}")
    prev_end = max(prev_end, e)
PY
