"""Tuning aid: wall time of one G1 MSM of n points for the window size given in ZKG_MSM_C (one process per setting)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
import zklaim_amd as zkg
zkg.init(0)
for n in [int(x) for x in sys.argv[1:]]:
    ks = bench.splitmix_fr(n, 1); d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_b = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zkg.fixed_base_g1_dev(bench.G1_GEN_MONT, d_k.data_ptr(), n, d_b.data_ptr())
    sc = torch.from_numpy(bench.splitmix_fr(n, 2).view(np.int64)).cuda()
    for _ in range(3): zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n)
    zkg.timing_reset()
    t = time.perf_counter()
    for _ in range(10): out = zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n)
    dt = (time.perf_counter() - t) / 10
    print(f"c={os.environ.get('ZKG_MSM_C','default')} n={n} wall {dt*1e3:.3f} ms accum {zkg.timing_dominant_ms()}", flush=True)
