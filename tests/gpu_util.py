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


def credential_payloads(k):
    """k satisfiable payloads (the workload of bench.py's prove legs)"""
    return [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i)
            for i in range(k)]


def oracle_pk_from_keypair(oracle, kp, csr, nv, l, keep):
    """the oracle's view of a key made by the product's generator: same arrays, A and B exchanged back when the generator swapped them"""
    A, B, C = csr
    if kp.swapped:
        A, B = B, A
    ocs = oracle.make_r1cs(nv, l, A, B, C, keep)
    m = kp.pk.domain_size or (1 << kp.pk.log_m)
    arrays = {name: kp.array(name, cnt, lim) for name, cnt, lim in (("A_query", nv + 1, 8), ("B_g1", nv + 1, 8), ("B_g2", nv + 1, 16), ("H_query", m - 1, 8),
              ("L_query", nv - l, 8), ("alpha_g1", 1, 8), ("beta_g1", 1, 8), ("delta_g1", 1, 8), ("beta_g2", 1, 16), ("delta_g2", 1, 16))}
    arrays["m"] = m
    keep.append(arrays)
    return ocs, oracle.make_pk(ocs, arrays), m
