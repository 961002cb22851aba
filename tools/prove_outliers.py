"""Latency tail of the prover: N sparse-witness proofs back to back on a resident key; every proof slower than 1.8 x the median is printed
with its per-stage device times (zkg_prove_stage_ms) and, under ZKG_DEBUG_TIMING=1, the host laps the library prints.
Usage: python tools/prove_outliers.py [payloads] [proofs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import zklaim_amd as zkg
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
zkg.init(0)
keep = []
pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i) for i in range(k)]
ck = zkg.ZklaimCircuit(zkg.make_ctx(pls, keep))
kp = zkg.Keypair(ck.r1cs, bench.splitmix_fr(5, 77))
crs = zkg.Crs(kp.pk)
rs = bench.splitmix_fr(2, 9)
tags, fidx, fvals = ck.sparse_witness()
for _ in range(5):
    crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
import gc
if os.environ.get("GC") == "0":
    gc.disable()
ts, st = [], []
for i in range(n):
    t = time.perf_counter(); crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1]); ts.append((time.perf_counter() - t) * 1e3)
    st.append([round(x, 3) for x in crs.stage_ms()])
ts = np.array(ts); med = float(np.median(ts))
print(f"k={k}: {n} proofs, median {med:.3f} p95 {np.percentile(ts, 95):.3f} p99 {np.percentile(ts, 99):.3f} max {ts.max():.3f} ms; {int((ts > 1.8 * med).sum())} above 1.8 x median")
for i in np.nonzero(ts > 1.8 * med)[0][:25]:
    print(f"  proof {i}: {ts[i]:.3f} ms, stages {st[i]}")
