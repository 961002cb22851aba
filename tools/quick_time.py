"""Rough device timings of the hot path at BASELINE sizes (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import zklaim_amd as zkg
from util import random_fr_canonical
from gpu_util import dev_bases_g1
zkg.init(0)
print(zkg.device_info())
for logn in (16, 18, 20, 22):
    n = 1 << logn
    a = torch.from_numpy(random_fr_canonical(n, 3).view(np.int64)).cuda()
    zkg.ntt_dev(a.data_ptr(), logn); torch.cuda.synchronize()
    t = time.time()
    for _ in range(10): zkg.ntt_dev(a.data_ptr(), logn)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(f"ntt 2^{logn}: {dt*1e3:.3f} ms  {64*n/dt/1e9:.1f} GB/s algorithmic")
for logn in (16, 18, 20):
    n = 1 << logn
    d_b, _, _ = dev_bases_g1(zkg, n, 1)
    sc = torch.from_numpy(random_fr_canonical(n, 2).view(np.int64)).cuda()
    zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n)
    zkg.timing_reset()
    t = time.time()
    for _ in range(3): zkg.msm_g1_dev(d_b.data_ptr(), sc.data_ptr(), n)
    dt = (time.time() - t) / 3
    print(f"msm g1 2^{logn}: {dt*1e3:.2f} ms wall  accum kernel {zkg.timing_dominant_ms()}  {96*n/dt/1e9:.2f} GB/s algorithmic")
