#!/bin/bash
# kernel + copy trace of a cold libsnark_prove (key from its blob, then the first proof): the 30 events in front of the proof's first kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/coldtrace; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o t -- python3 tools/zklaim_benchmark.py 8 --runs 2 > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
tail -2 $OUT/out.txt
python3 - $OUT/t <<'PY'
import csv, sys, glob
d=sys.argv[1]; ev=[]
for r in csv.DictReader(open(glob.glob(d+"/*kernel_trace.csv")[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ","")[:40], "q"+r.get("Queue_Id","")))
for f in glob.glob(d+"/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")[-14:], ""))
ev.sort()
# the last cold call: find last k_expand_tags whose preceding 30 ms contain k_decompress
idx=[i for i,e in enumerate(ev) if "k_expand_tags" in e[2]]
# pick the expand_tags that follows a decompress within 60 ms
cands=[i for i in idx if any("k_decompress" in ev[j][2] and ev[i][0]-ev[j][0] < 80e6 for j in range(max(0,i-400), i))]
i0=cands[-1]
t0=ev[i0][0]
for s,e,n,q in ev[max(0,i0-25):i0+6]:
    print(f"{(s-t0)/1e6:9.3f} ms  +{(e-s)/1e3:9.1f} us  {n} {q}")
PY
