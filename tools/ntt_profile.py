"""Runs the 2^logn radix-2 NTT (forward, resident) REPS times: the target of the rocprofv3 kernel-trace / --pmc passes for k_ntt_pass.
Usage: python tools/ntt_profile.py [logn] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import zklaim_amd as zkg
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
zkg.init(0)
n = 1 << logn
a = torch.from_numpy(bench.splitmix_fr(n, bench.SEED + 3).view(np.int64)).cuda()
for _ in range(3):
    zkg.ntt_dev(a.data_ptr(), logn)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    zkg.ntt_dev(a.data_ptr(), logn)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"ntt 2^{logn}: {dt * 1e3:.4f} ms per transform, {64 * n / dt / 1e9:.1f} GB/s algorithmic")
