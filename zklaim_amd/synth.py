"""Synthetic "zklaim-shaped" R1CS + witness (BASELINE config 4 / SURVEY.md §8d cfg4).

zklaim's credential circuit (zklaim/zklaim_gadget.cpp:348-783 of the reference: one SHA-256 compression, five 64-bit
comparisons and packing per payload) is almost entirely boolean: >95 % of the wires are bits and a row has ~1-3 terms per
matrix.  Until the gadget layer is rebuilt this generator emits systems with the same profile:
bitness, AND and XOR gates over bits, 64-term packing rows and a few full-width products, padded with 1*0=0 rows so that
C + l + 1 lands exactly on a radix-2 domain.  Pure Python integers; no field library needed.
"""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT = (1 << 256) % R
MASK = (1 << 64) - 1


def _limbs_mont(v):
    v = v * MONT % R
    return [(v >> (64 * i)) & MASK for i in range(4)]


class _Csr:
    def __init__(self):
        self.rp, self.col, self.val = [0], [], []

    def row(self, lc):
        for i, c in lc:
            self.col.append(i); self.val.append(c % R)
        self.rp.append(len(self.col))

    def arrays(self):
        cache = {}
        out = np.empty((len(self.val), 4), np.uint64)
        for k, v in enumerate(self.val):
            if v not in cache:
                cache[v] = _limbs_mont(v)
            out[k] = cache[v]
        return np.array(self.rp, np.uint32), np.array(self.col, np.uint32), out


def zklaim_shaped(log_m, num_inputs=41, seed=1, full_width_frac=0.03):
    """-> (n, l, A, B, C, witness (n,4) uint64 Montgomery) with C + l + 1 == 2^log_m exactly."""
    rng = np.random.default_rng(seed)
    C_total = (1 << log_m) - num_inputs - 1
    A, B, Cm = _Csr(), _Csr(), _Csr()
    w = [int.from_bytes(rng.bytes(31), "little") for _ in range(num_inputs)]       # public inputs: packed 248-bit values
    bits = []                                                                       # indices (1-based variable ids) of boolean wires

    def new(v):
        w.append(v % R)
        return len(w)                                                               # variable id (0 is the constant)

    rows = 0
    budget = C_total - 8
    while rows < budget:
        kind = rng.random()
        if len(bits) < 64 or kind < 0.30:                                            # fresh bit + bitness row  b*(1-b)=0
            b = new(int(rng.integers(0, 2))); bits.append(b)
            A.row([(b, 1)]); B.row([(0, 1), (b, -1)]); Cm.row([]); rows += 1
        elif kind < 0.60:                                                            # AND
            x, y = (bits[int(i)] for i in rng.integers(0, len(bits), 2))
            c = new(w[x - 1] * w[y - 1]); bits.append(c)
            A.row([(x, 1)]); B.row([(y, 1)]); Cm.row([(c, 1)]); rows += 1
        elif kind < 0.95 - full_width_frac:                                          # XOR: (2a)*b = a + b - c
            x, y = (bits[int(i)] for i in rng.integers(0, len(bits), 2))
            c = new(w[x - 1] ^ w[y - 1]); bits.append(c)
            A.row([(x, 2)]); B.row([(y, 1)]); Cm.row([(x, 1), (y, 1), (c, -1)]); rows += 1
        elif kind < 0.95:                                                            # full-width product
            x = new(int.from_bytes(rng.bytes(31), "little")); y = new(int.from_bytes(rng.bytes(31), "little"))
            z = new(w[x - 1] * w[y - 1])
            A.row([(x, 1)]); B.row([(y, 1)]); Cm.row([(z, 1)]); rows += 1
        else:                                                                        # 64-bit packing row
            sel = [bits[int(i)] for i in rng.integers(0, len(bits), 64)]
            v = new(sum(w[b - 1] << i for i, b in enumerate(sel)))
            A.row([(b, 1 << i) for i, b in enumerate(sel)]); B.row([(0, 1)]); Cm.row([(v, 1)]); rows += 1
    while rows < C_total:                                                            # pad: 1 * 0 = 0
        A.row([(0, 1)]); B.row([]); Cm.row([]); rows += 1
    n = len(w)
    cache = {0: _limbs_mont(0), 1: _limbs_mont(1)}
    wit = np.empty((n, 4), np.uint64)
    for i, v in enumerate(w):
        if v not in cache:
            if v > 1:
                wit[i] = _limbs_mont(v); continue
        wit[i] = cache[v]
    return n, num_inputs, A.arrays(), B.arrays(), Cm.arrays(), wit
