"""Key life cycle stress through the seam: N times (trusted setup, prove, verify, drop the resident key) on a k-payload credential;
device memory before / after (a leak in the key load, the witness tables or the prover slot would show as growth).
Usage: python tools/key_cycle_stress.py [k] [cycles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import zklaim_amd as zkg
from gpu_util import credential_payloads
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 40
zkg.init(0)
free0 = None
for c in range(cycles):
    keep = []
    ctx = zkg.make_ctx(credential_payloads(k), keep)
    assert zkg.libsnark_trusted_setup(ctx) == 0 and zkg.libsnark_prove(ctx) == 0 and zkg.libsnark_verify(ctx) == 0
    assert zkg.libsnark_prove(ctx) == 0 and zkg.libsnark_verify(ctx) == 0
    zkg.lib().zkg_compat_reset()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    if c == 4:
        free0 = free                      # after the runtime's pools have warmed up
print(f"k={k}: {cycles} key cycles, device memory delta since cycle 5: {(free0 - free) / 2**20:.1f} MiB")
