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
for _ in range(10):
    t = time.perf_counter(); rc, proof = crs.prove(w, rs[0], rs[1]); print(rc, round((time.perf_counter() - t) * 1e3, 3), "ms", [round(x, 3) for x in crs.stage_ms()])
