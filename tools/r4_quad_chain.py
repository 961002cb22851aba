"""Latency of the 29-bit quad addition in isolation (zkg_g1_add_quad29, chain of 129 dependent additions per quad): run under
rocprofv3 --kernel-trace --stats; n quads = 16 per wavefront."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import zklaim_amd as zkg
from util import random_fr_canonical
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import zkoracle
zkg.init(0)
ks = random_fr_canonical(64, 7)
pts = zkoracle.g1_fixed_base(zkoracle.g1_generator(), ks)          # 64 affine points
jac = np.zeros((64, 12), np.uint64); jac[:, :8] = pts; jac[:, 8:] = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], np.uint64)
for waves in (256, 1024, 2048, 4096):
    n = 16 * waves
    a = np.tile(jac[:32], (n // 32 + 1, 1))[:n]; b = np.tile(jac[32:], (n // 32 + 1, 1))[:n]
    out = zkg.g1_add_quad29(a, b, 64)
    print(waves, "wavefronts", out[0][:2])
for waves in (256, 1024, 2048, 4096):                       # the pair form: 32 pairs per wavefront
    n = 32 * waves
    a = np.tile(jac[:32], (n // 32 + 1, 1))[:n]; b = np.tile(jac[32:], (n // 32 + 1, 1))[:n]
    out = zkg.g1_add_pair29(a, b, 64)
    print(waves, "wavefronts (pairs)", out[0][:2])
