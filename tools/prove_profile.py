"""Runs a few Groth16 proofs at m = 2^logm on a synthetic zklaim-shaped system (for rocprofv3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import zklaim_amd as zkg
from zklaim_amd import synth
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 18
zkg.init(0)
nv, l, A, B, C, w = synth.zklaim_shaped(logm, num_inputs=41, seed=4)
m = 1 << logm
def pts(cnt, seed, g2=False):
    ks = bench.splitmix_fr(cnt, seed); d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    o = torch.empty((cnt, 16 if g2 else 8), dtype=torch.int64, device="cuda")
    (zkg.fixed_base_g2_dev if g2 else zkg.fixed_base_g1_dev)(bench.G2_GEN_MONT if g2 else bench.G1_GEN_MONT, d_k.data_ptr(), cnt, o.data_ptr()); torch.cuda.synchronize()
    return o.cpu().numpy().view(np.uint64)
s1 = pts(3, 1); s2 = pts(2, 2, True)
arrays = dict(alpha_g1=s1[0], beta_g1=s1[1], delta_g1=s1[2], beta_g2=s2[0], delta_g2=s2[1], A_query=pts(nv + 1, 3), B_g1=pts(nv + 1, 4),
              B_g2=pts(nv + 1, 5, True), H_query=pts(m - 1, 6), L_query=pts(nv - l, 7))
keep = []
crs = zkg.Crs(zkg.make_pk(zkg.make_r1cs(nv, l, A, B, C, keep), arrays, logm, keep))
rs = bench.splitmix_fr(2, 9)
for _ in range(14):
    t = time.perf_counter(); rc, proof = crs.prove(w, rs[0], rs[1]); print(rc, (time.perf_counter() - t) * 1e3, "ms", crs.stage_ms())
