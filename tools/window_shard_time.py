"""One rank's share of an 8-GPU job under the two partitions, timed on one GPU: points (2^20 own points, all 16 windows) against
windows (all 2^23 points resident, windows g and g + 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import zklaim_amd as zkg
zkg.init(0)
G = 8
n = 1 << 23
ks = bench.splitmix_fr(n, 1); d_k = torch.from_numpy(ks.view(np.int64)).cuda()
d_b = torch.empty((n, 8), dtype=torch.int64, device="cuda")
zkg.fixed_base_g1_dev(bench.G1_GEN_MONT, d_k.data_ptr(), n, d_b.data_ptr())
sc = torch.from_numpy(bench.splitmix_fr(n, 2).view(np.int64)).cuda()
def timed(fn, reps=5):
    for _ in range(2): fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e3
n1 = n // G
print(f"points: rank share 2^20 points x 16 windows: {timed(lambda: zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n1)):.2f} ms")
for g in (0, 7):
    print(f"windows: rank {g} share 2^23 points x windows {g},{g+8}: {timed(lambda: zkg.msm_g1_windows_dev(d_b.data_ptr(), sc.data_ptr(), n, g, G)):.2f} ms")
print(f"single GPU, all 2^23 points: {timed(lambda: zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n), 3):.2f} ms")
