"""Runs a few G1 MSMs of 2^logn random points (for rocprofv3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
import zklaim_amd as zkg
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << logn
zkg.init(0)
ks = bench.splitmix_fr(n, 1); d_k = torch.from_numpy(ks.view(np.int64)).cuda()
d_b = torch.empty((n, 8), dtype=torch.int64, device="cuda")
zkg.fixed_base_g1_dev(bench.G1_GEN_MONT, d_k.data_ptr(), n, d_b.data_ptr())
sc = torch.from_numpy(bench.splitmix_fr(n, 2).view(np.int64)).cuda()
for _ in range(4):
    t = time.perf_counter(); zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n); print((time.perf_counter() - t) * 1e3, "ms")
