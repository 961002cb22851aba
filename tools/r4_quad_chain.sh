#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_quad
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o q -- python3 tools/r4_quad_chain.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob
allrows = list(csv.DictReader(open(glob.glob("gpurun_out/r4_quad/t/*kernel_trace.csv")[0])))
for name, per, what in (("k_add_quad29", 16, "quads"), ("k_add_pair29", 32, "pairs")):
    rows = [r for r in allrows if name in r["Kernel_Name"]]
    for r, w in zip(rows, (256, 1024, 2048, 4096)):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{w:5d} wavefronts of {per} {what}, 129 dependent additions: {us:8.1f} us = {us / 129:6.3f} us per addition = {us / 129 / per * 1e3:6.1f} ns of a wavefront per addition")
PY
