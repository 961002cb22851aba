"""Proofs per second with several callers: T host threads prove the same k-payload credential on ONE resident key (the key's prover
slots; SHARED_KEY=0: each on a resident copy of the key) (argv: k, threads, proofs per thread).  One caller = the latency figure bench.py reports; more callers show what the chip still
has to give when a proof's latency-bound phases run under another proof's kernels."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import zklaim_amd as zkg
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
zkg.init(0)
keep = []
pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i) for i in range(k)]
ck = zkg.ZklaimCircuit(zkg.make_ctx(pls, keep))
tags, fidx, fvals = ck.sparse_witness()
kp = zkg.Keypair(ck.r1cs, bench.splitmix_fr(5, 77))
rs = bench.splitmix_fr(2, 9)
shared = os.environ.get("SHARED_KEY", "1") != "0"        # one resident key for all callers (its prover slots) or a copy per caller
for nthreads in range(1, T + 1):
    crss = [zkg.Crs(kp.pk)] * nthreads if shared else [zkg.Crs(kp.pk) for _ in range(nthreads)]
    expect = crss[0].prove_sparse(tags, fidx, fvals, rs[0], rs[1])
    for c in crss:
        assert c.prove_sparse(tags, fidx, fvals, rs[0], rs[1]) == expect
    bad = []
    def run(c):
        for _ in range(reps):
            if c.prove_sparse(tags, fidx, fvals, rs[0], rs[1]) != expect:
                bad.append(1)
    th = [threading.Thread(target=run, args=(c,)) for c in crss]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    assert not bad
    print(f"k={k} callers={nthreads}: {nthreads * reps / dt:.1f} proofs/s ({dt / reps * 1e3:.3f} ms per proof per caller)", flush=True)
    for c in set(crss): c.free()
