import sys, time
sys.path.insert(0, '/root/repo')
import zklaim_amd as zkg
pass  # host-only timing: no device needed
k = int(sys.argv[1])
keep = []
pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i) for i in range(k)]
ctx = zkg.make_ctx(pls, keep)
t = time.perf_counter(); ck = zkg.ZklaimCircuit(ctx, witness_only=True); t1 = time.perf_counter() - t
t = time.perf_counter(); ck2 = zkg.ZklaimCircuit(ctx); t2 = time.perf_counter() - t
print(f"k={k} witness-only pass {t1*1e3:.2f} ms; full circuit (constraints + CSR) {t2*1e3:.1f} ms; vars {ck.r1cs.num_variables}")
