"""Shared fixtures for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np
import pytest

import zklaim_amd
from util import random_fr_canonical


@pytest.fixture(scope="session")
def zkg():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from zklaim_amd import build
    build.build()
    zklaim_amd.init(0)
    yield zklaim_amd
    zklaim_amd.shutdown()


def dev_bases_g1(zkg, n, seed):
    """n synthetic G1 bases k_i*G generated ON DEVICE by the product's fixed-base kernel; returns (torch tensor, numpy copy)."""
    import torch
    ks = random_fr_canonical(n, seed)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_out = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    g1 = np.zeros(8, np.uint64)
    g1[:4] = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
    g1[4:] = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
    zkg.fixed_base_g1_dev(g1, d_k.data_ptr(), n, d_out.data_ptr())
    torch.cuda.synchronize()
    return d_out, d_out.cpu().numpy().view(np.uint64), ks
