"""Time of the partial-point exchange of bench.py --gpus N (zklaim_amd/dist.py combine_partials_g1: RCCL all-gather of 96 B per rank + EC sum),
one rank rehearsing it (ZKG_DIST_FORCE_EXCHANGE=1).  Run under torch.distributed.run --nproc-per-node 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import zklaim_amd as zkg
from zklaim_amd import dist as zdist
os.environ["ZKG_DIST_FORCE_EXCHANGE"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=torch.device("cuda", 0))
zkg.init(0)
part = np.zeros(12, np.uint64); part[4] = 1
for _ in range(20):
    zdist.combine_partials_g1(part, device="cuda")
ts = []
for _ in range(200):
    t = time.perf_counter(); zdist.combine_partials_g1(part, device="cuda"); ts.append(time.perf_counter() - t)
ts.sort()
print(f"exchange + sum: median {ts[100] * 1e6:.1f} us, p95 {ts[190] * 1e6:.1f} us, min {ts[0] * 1e6:.1f} us", file=sys.stderr)
dist.destroy_process_group()
