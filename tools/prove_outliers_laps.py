"""Parses the [zkg] laps a ZKG_DEBUG_TIMING=1 run of tools/prove_outliers.py wrote to stderr: prints the laps of every proof whose last
lap exceeds 1.8 x the median.  Usage: python tools/prove_outliers_laps.py stderr.log"""
import re, sys
import numpy as np
proofs, cur = [], []
for line in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[zkg\]\s+(.+?)\s+([0-9.]+) ms", line)
    if not m:
        continue
    cur.append((m.group(1).strip(), float(m.group(2))))
    if m.group(1).strip().startswith("assembled+serialised"):
        proofs.append(cur); cur = []
tot = np.array([p[-1][1] for p in proofs]); med = float(np.median(tot))
print(len(proofs), "proofs, median of the last lap", round(med, 3), "ms")
typical = proofs[len(proofs) // 2]
print("typical:", [(n, t) for n, t in typical if not n.startswith(("sort", "accum"))])
for i in np.nonzero(tot > 1.8 * med)[0][:12]:
    print(i, [(n, t) for n, t in proofs[i] if not n.startswith(("sort", "accum"))])
