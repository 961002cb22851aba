"""Helpers shared by the tests: big-int <-> ABI limb arrays (include/zkg.h conventions)."""
import json
import os

import numpy as np

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT = 1 << 256
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK = (1 << 64) - 1


def golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def h(x):
    return int(x, 16)


def limbs(x):
    return [(x >> (64 * i)) & MASK for i in range(4)]


def from_limbs(a):
    return sum(int(v) << (64 * i) for i, v in enumerate(np.asarray(a).reshape(-1)[:4]))


def arr(vals, p=None):
    """ints -> (n,4) uint64; p given => Montgomery form mod p, else canonical."""
    return np.array([limbs(v * MONT % p if p else v) for v in vals], dtype=np.uint64).reshape(len(vals), 4)


def ints(a, p=None):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    rinv = pow(MONT, -1, p) if p else 1
    return [(from_limbs(r) * rinv % p) if p else from_limbs(r) for r in a]


def g1_aff(pt):
    """golden [x,y] hex (or None) -> 8 limbs Montgomery (all-zero = infinity)"""
    if pt is None:
        return np.zeros(8, np.uint64)
    return arr([h(pt[0]), h(pt[1])], Q).reshape(8)


def g2_aff(pt):
    if pt is None:
        return np.zeros(16, np.uint64)
    return arr([h(pt[0][0]), h(pt[0][1]), h(pt[1][0]), h(pt[1][1])], Q).reshape(16)


def g1_jac_expected(pt):
    """normalised jac as the ABI returns it"""
    if pt is None:
        return arr([0, 1, 0], Q).reshape(12)
    return arr([h(pt[0]), h(pt[1]), 1], Q).reshape(12)


def g2_jac_expected(pt):
    if pt is None:
        return arr([0, 0, 1, 0, 0, 0], Q).reshape(24)
    return arr([h(pt[0][0]), h(pt[0][1]), h(pt[1][0]), h(pt[1][1]), 1, 0], Q).reshape(24)


class SplitMix64:
    """vectorised SplitMix64 (numpy); the RNG BASELINE.md names for synthetic inputs"""

    def __init__(self, seed):
        self.s = np.uint64(seed & MASK)

    def draw(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.s + idx * np.uint64(0x9E3779B97F4A7C15)
            self.s = self.s + np.uint64(n) * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))


R_LIMBS = np.array(limbs(R), dtype=np.uint64)


def random_fr_canonical(n, seed):
    """n uniform values in [0, r): 4 draws -> 254 bits -> rejection (canonical limbs, (n,4) uint64)"""
    rng = SplitMix64(seed)
    out = np.zeros((0, 4), np.uint64)
    while out.shape[0] < n:
        k = int((n - out.shape[0]) * 1.4) + 16
        v = rng.draw(4 * k).reshape(k, 4).copy()
        v[:, 3] &= np.uint64((1 << 62) - 1)
        lt = np.zeros(k, bool); eq = np.ones(k, bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (v[:, i] < R_LIMBS[i]); eq &= v[:, i] == R_LIMBS[i]
        out = np.concatenate([out, v[lt]])
    return np.ascontiguousarray(out[:n])
