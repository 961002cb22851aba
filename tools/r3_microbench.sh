#!/bin/bash
# Round-3 measurement batch on the GPU box (run through gpurun from the repo root):
#   tools/mul_variants.hip (built here into build_variants/, which travels), its 29-bit product vectors checked with Python integers,
#   and the effective clock under bench.py's kernels (GRBM_GUI_ACTIVE in its own --pmc pass, kernel-trace only beside it).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_micro
mkdir -p $OUT
[ -x build_variants/mul_variants ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mul_variants.hip -o build_variants/mul_variants 2>/dev/null || exit 1
timeout -k 10 300 ./build_variants/mul_variants > $OUT/mul_variants.txt 2>&1 || { echo "mul_variants failed"; tail -5 $OUT/mul_variants.txt; exit 1; }
python3 - $OUT/mul_variants.txt <<'PY' || exit 1
import re, sys
q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
rinv = pow(1 << 261, -1, q); n = 0
for line in open(sys.argv[1]):
    m = re.match(r"VEC29 a=(\w+) b=(\w+) r=(\w+)", line)
    if m:
        a, b, r = (int(x, 16) for x in m.groups())
        assert r % q == a * b * rinv % q, line
        n += 1
assert n >= 4
print(f"29-bit product: {n} vectors equal a*b*2^-261 mod q")
PY
cat $OUT/mul_variants.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o g -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $OUT/bench_grbm.json 2> $OUT/bench_grbm.err || { echo "grbm failed"; tail -5 $OUT/bench_grbm.err; exit 1; }
python3 - $OUT/grbm <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
kt = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in kt:
        ns = kt[r["Dispatch_Id"]]
        agg[r["Kernel_Name"][:60]].append((float(r["Counter_Value"]) / 8.0 / ns, ns))
for k, v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:8]:
    print(f"{k:60s} launches {len(v):3d}  avg {sum(x[1] for x in v) / len(v) / 1e3:9.1f} us  effective clock {sum(x[0] for x in v) / len(v):.3f} GHz")
PY
