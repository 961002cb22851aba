#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_quad
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o q -- python3 tools/r4_quad_chain.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/r4_quad/t/*kernel_trace.csv")[0])) if "k_add_quad29" in r["Kernel_Name"]]
for r, w in zip(rows, (256, 1024, 2048, 4096)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{w:5d} wavefronts of 16 quads, 129 dependent quad additions: {us:8.1f} us = {us / 129:6.3f} us per addition")
PY
