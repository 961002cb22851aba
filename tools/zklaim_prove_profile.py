"""Runs a few Groth16 proofs of a real k-payload zklaim credential (for rocprofv3).  ZKG_SERIAL_MSM=1 makes the prover wait for
each multi-exponentiation before starting the next, so that per-kernel durations are those of the kernel alone on the chip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import zklaim_amd as zkg
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
zkg.init(0)
keep = []
pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i) for i in range(k)]
ck = zkg.ZklaimCircuit(zkg.make_ctx(pls, keep))
w = ck.witness()
kp = zkg.Keypair(ck.r1cs, bench.splitmix_fr(5, 77))
crs = zkg.Crs(kp.pk)
rs = bench.splitmix_fr(2, 9)
import numpy as np
reps = int(os.environ.get("REPS", "10"))
ts = []
for _ in range(reps):
    t = time.perf_counter(); rc, proof = crs.prove(w, rs[0], rs[1]); ts.append((time.perf_counter() - t) * 1e3)
    if reps <= 10:
        print(rc, round(ts[-1], 3), "ms", [round(x, 3) for x in crs.stage_ms()])
tags, fidx, fvals = ck.sparse_witness()
tsp = []
for _ in range(reps):
    t = time.perf_counter(); rc2, proof2 = crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1]); tsp.append((time.perf_counter() - t) * 1e3)
assert rc == 0 and rc2 == 0 and proof2 == proof
print(f"k={k} ZKG_PRIO={os.environ.get('ZKG_PRIO')} dense median {np.median(ts[2:]):.3f} min {min(ts):.3f} | sparse median {np.median(tsp[2:]):.3f} min {min(tsp):.3f} ms | stages", [round(x, 3) for x in crs.stage_ms()])
