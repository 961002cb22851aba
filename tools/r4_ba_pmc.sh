#!/bin/bash
# Counters of the batched-affine variant (ZKG_ACCUM_BA = $1, default 3) and of the default accumulation beside it: one --pmc pass per
# counter group (FETCH_SIZE, WRITE_SIZE, SQ), kernel trace only beside each; summary -> gpurun_out/r4_ba/pmc_summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=${1:-3}
OUT=gpurun_out/r4_ba
mkdir -p $OUT
pass() { tag=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$tag -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || { tail -5 $OUT/pmc_$tag.err; exit 1; }; }
export ZKG_ACCUM_BA=$L
pass ba_fetch FETCH_SIZE
pass ba_write WRITE_SIZE
pass ba_sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
export ZKG_ACCUM_BA=0
pass base_sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
python3 - $OUT <<'PY' | tee $OUT/pmc_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
def load(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}
want = ("k_ba_level", "k_bucket_accum")
f, w = load("ba_fetch"), load("ba_write")
print("HBM-side counters per launch (raw counter x 1024 B; FETCH_SIZE under-reports wide coalesced streams by up to 2x, see profiles/pmc_traffic.json)")
for k in sorted(f):
    if any(x in k for x in want):
        print(f"  {k:34s} FETCH_SIZE {f[k].get('FETCH_SIZE', 0) * 1024 / 1e6:9.1f} MB   WRITE_SIZE {w.get(k, {}).get('WRITE_SIZE', 0) * 1024 / 1e6:9.1f} MB")
for tag in ("ba_sq", "base_sq"):
    print(f"SQ counters per launch, {tag}")
    for k, v in sorted(load(tag).items()):
        if any(x in k for x in want):
            wc = v.get("SQ_WAVE_CYCLES", 1)
            print(f"  {k:34s} waves {v.get('SQ_WAVES', 0):9.0f}  VALU insts {v.get('SQ_INSTS_VALU', 0) / 1e6:8.1f} M  of wave cycles: issuing VALU {100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.1f} %  "
                  f"waiting (any) {100 * v.get('SQ_WAIT_ANY', 0) / wc:5.1f} %  ready-not-issued {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f} %")
PY
