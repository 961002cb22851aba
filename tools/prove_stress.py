"""Stress: many proofs on one resident key, dense and sparse witness alternating, every proof's bytes compared with the first; device
memory before / after (no growth: workspaces are grow-only and reused).  Usage: python tools/prove_stress.py [k] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import zklaim_amd as zkg
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
zkg.init(0)
keep = []
ck = zkg.ZklaimCircuit(zkg.make_ctx([dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i)
                                     for i in range(k)], keep))
w = ck.witness(); tags, fidx, fvals = ck.sparse_witness()
kp = zkg.Keypair(ck.r1cs, bench.splitmix_fr(5, 77))
crs = zkg.Crs(kp.pk)
rs = bench.splitmix_fr(2, 9)
rc, first = crs.prove(w, rs[0], rs[1])
assert rc == 0 and zkg.groth16_verify(kp.vk_blob(), w[:ck.r1cs.num_inputs], first) == 0
for _ in range(5):
    crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
free0, _ = torch.cuda.mem_get_info()
t = time.perf_counter(); worst = 0.0
for i in range(count):
    t1 = time.perf_counter()
    rc, p = crs.prove(w, rs[0], rs[1]) if i % 2 else crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
    worst = max(worst, time.perf_counter() - t1)
    assert rc == 0 and p == first, i
dt = time.perf_counter() - t
free1, _ = torch.cuda.mem_get_info()
print(f"k={k}: {count} proofs, all bytes identical, {dt / count * 1e3:.3f} ms mean, worst {worst * 1e3:.2f} ms, device memory delta {(free0 - free1) / 2**20:.1f} MiB")
