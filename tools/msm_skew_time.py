"""Tuning aid: the generic MSM entry point on witness-like scalars (97 % bits, 3 % full-size), with either sort."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import zklaim_amd as zkg
zkg.init(0)
for n in [int(x) for x in sys.argv[1:]]:
    ks = bench.splitmix_fr(n, 1); d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_b = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zkg.fixed_base_g1_dev(bench.G1_GEN_MONT, d_k.data_ptr(), n, d_b.data_ptr())
    sc = bench.splitmix_fr(n, 2)
    rng = np.random.default_rng(3)
    bits = rng.random(n) < 0.97
    sc[bits] = 0; sc[bits, 0] = rng.integers(0, 2, bits.sum()).astype(np.uint64)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    res = {}
    for hint in (False, True):
        for _ in range(3): out = zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n, mostly_bits=hint)
        t = time.perf_counter()
        for _ in range(10): out = zkg.msm_g1_dev(d_b.data_ptr(), d_sc.data_ptr(), n, mostly_bits=hint)
        res[hint] = (out, (time.perf_counter() - t) / 10 * 1e3)
    assert np.array_equal(res[False][0], res[True][0])
    print(f"n={n} witness-like scalars: {res[False][1]:.3f} ms without the hint, {res[True][1]:.3f} ms with ZKG_SCALARS_MOSTLY_BITS", flush=True)
