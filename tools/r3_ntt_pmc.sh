#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_ntt_pmc
mkdir -p $OUT
for v in 29 32; do
  if [ $v = 32 ]; then export ZKG_NTT_32=1; else unset ZKG_NTT_32; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq$v -o p -- python3 tools/ntt_profile.py 20 5 > $OUT/log$v.txt 2>&1 || { tail -3 $OUT/log$v.txt; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks$v -o k -- python3 tools/ntt_profile.py 20 20 > $OUT/logk$v.txt 2>&1 || exit 1
  python3 - $OUT/sq$v $OUT/ks$v/k_kernel_stats.csv <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "ntt_pass" in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "VGPR?", flush=True)
for r in csv.DictReader(open(sys.argv[2])):
    if "ntt_pass" in r["Name"]: print(r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
PY
done
