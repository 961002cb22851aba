"""Is one 2^logn NTT throughput-bound?  Runs K independent transforms (own data, own HIP stream each) concurrently and reports the time
per transform: equal to the single-stream time means the multiplier is saturated; lower means a single transform leaves the chip idle in
its load / store phases.  Usage: python tools/ntt_concurrency.py [logn]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import zklaim_amd as zkg
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
zkg.init(0)
n = 1 << logn
bufs = [torch.from_numpy(bench.splitmix_fr(n, bench.SEED + 30 + i).view(np.int64)).cuda() for i in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
for b, s in zip(bufs, streams):
    zkg.ntt_dev(b.data_ptr(), logn, stream=s.cuda_stream)
torch.cuda.synchronize()
for k in (1, 2, 3, 4):
    reps = 30
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for i in range(k):
            zkg.ntt_dev(bufs[i].data_ptr(), logn, stream=streams[i].cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / (reps * k)
    print(f"2^{logn}: {k} concurrent stream(s): {dt * 1e3:.4f} ms per transform ({64 * n / dt / 1e9:.0f} GB/s algorithmic)")
