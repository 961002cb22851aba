"""Host-only timing of the zklaim circuit passes for k payloads (no GPU work): witness-only (what libsnark_prove runs per proof) and
recording (what libsnark_trusted_setup runs once).  ZKG_DEBUG_TIMING=1 prints the phases."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zklaim_amd as zkg
for k in [int(x) for x in sys.argv[1:]] or [1, 8, 20]:
    keep = []
    pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i) for i in range(k)]
    ctx = zkg.make_ctx(pls, keep)
    zkg.ZklaimCircuit(ctx, witness_only=True).free()
    t = time.perf_counter()
    for _ in range(10):
        zkg.ZklaimCircuit(ctx, witness_only=True).free()
    t1 = (time.perf_counter() - t) / 10
    t = time.perf_counter(); ck = zkg.ZklaimCircuit(ctx, with_witness=False); t2 = time.perf_counter() - t
    print(f"k={k}: witness-only pass {t1*1e3:.2f} ms (incl. the Python wrapper's witness copy); recording pass {t2*1e3:.1f} ms; variables {ck.r1cs.num_variables}", flush=True)
    ck.free()
