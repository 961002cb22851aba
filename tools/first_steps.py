"""Per-call wall time of the first calls of the two MSM entry points in a fresh process (the first allocates the work buffers)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zklaim_amd as zkg
from gpu_util import dev_bases_g1
from util import random_fr_canonical
zkg.init(0)
n = 1 << 20
d_bases, bases, _ = dev_bases_g1(zkg, n, 7)
sc = random_fr_canonical(n, 8)
h = torch.from_numpy(sc.view(np.int64)).pin_memory()
ts = []
for i in range(12):
    t0 = time.perf_counter(); zkg.msm_g1_host_scalars(d_bases.data_ptr(), h.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
print("host_scalars steps ms:", " ".join(f"{t:.2f}" for t in ts))
d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
ts = []
for i in range(8):
    t0 = time.perf_counter(); zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
print("resident steps ms:", " ".join(f"{t:.2f}" for t in ts))
