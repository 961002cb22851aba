"""Wall time of libsnark_trusted_setup and of the first / second libsnark_prove on a fresh key (argv: payload counts, default 8 and 20); with ZKG_DEBUG_TIMING=1 the
phases inside (circuit, CSR export, Lagrange, QAP evaluation, GPU fixed-base batches, blobs) are printed by the library."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import zklaim_amd as zkg
from gpu_util import credential_payloads
zkg.init(0)
for k in ([int(a) for a in sys.argv[1:]] or [8, 20]):
    keep = []
    ctx = zkg.make_ctx(credential_payloads(k), keep)
    for rep in range(2):
        t = time.perf_counter(); rc = zkg.libsnark_trusted_setup(ctx); dt = time.perf_counter() - t
        print(f"k={k} setup rc={rc} {dt*1e3:.1f} ms pk {ctx.pk_size/1e6:.1f} MB", flush=True)
    t = time.perf_counter(); rc = zkg.libsnark_prove(ctx); print("first prove", (time.perf_counter()-t)*1e3, flush=True)
    t = time.perf_counter(); rc = zkg.libsnark_prove(ctx); print("second prove", (time.perf_counter()-t)*1e3, flush=True)
