"""Independent big-integer reference for the alt_bn128 Groth16 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``zklaim_amd/`` imports this file; it is
used by ``tests/`` and by ``tests/golden/gen_golden.py`` to pin the C++ oracle
(``oracle/zkoracle.cpp``) and, through it, the HIP path.

Everything here is plain Python ``int`` arithmetic written from the published
definitions (no Montgomery tricks, no windowing, no FFT butterflies in the
checks that matter): it is slow on purpose so that it shares no structure with
the code it checks.

Reference call sites this mirrors (relative to /root/reference/):
  * zklaim/snark.cpp:126      r1cs_gg_ppzksnark_prover(pk, primary, auxiliary)
  * zklaim/snark.cpp:91       r1cs_gg_ppzksnark_generator(constraint_system)
  * zklaim/libsnark_wrapper.cpp:170-181  operator<<(proof) -> bytes
The arithmetic itself lives in scipr-lab/libsnark (+libff, libfqfft), an
un-vendored, un-pinned submodule (``.gitmodules:1-6``); the algorithms are
restated from their published form.  PARITY UNPINNED: the reference ships no
golden vectors for this path.
"""

# ---------------------------------------------------------------- constants
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT_R = 1 << 256
FR_S = 28                       # two-adicity of R-1
FR_GEN = 5                      # multiplicative generator of Fr (libff alt_bn128 init)
FR_ROOT = pow(FR_GEN, (R - 1) >> FR_S, R)   # primitive 2^28-th root of unity
G1_B = 3
G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def inv(a, p):
    return pow(a, p - 2, p)


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % Q, Q)
    return (a[0] * d % Q, (-a[1] * d) % Q)


G2_B = f2_mul((3, 0), f2_inv((9, 1)))      # twist coefficient 3/(9+u)


# ---------------------------------------------------------------- affine curve ops (None = infinity)
class Field1:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q)
    sub = staticmethod(lambda a, b: (a - b) % Q)
    mul = staticmethod(lambda a, b: a * b % Q)
    neg = staticmethod(lambda a: (-a) % Q)
    inv = staticmethod(lambda a: inv(a, Q))
    b = G1_B


class Field2:
    zero, one = (0, 0), (1, 0)
    add, sub, mul, neg, inv = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_neg, f2_inv))
    b = G2_B


def ec_add(F, P, S):
    if P is None: return S
    if S is None: return P
    x1, y1 = P; x2, y2 = S
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        lam = F.mul(F.mul((3 % Q) if F is Field1 else (3, 0), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, P):
    return None if P is None else (P[0], F.neg(P[1]))


def ec_mul(F, k, P):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(F, acc, P)
        P = ec_add(F, P, P)
        k >>= 1
    return acc


def on_curve(F, P):
    if P is None: return True
    x, y = P
    return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), F.b)


def g1_mul(k, P=G1_GEN): return ec_mul(Field1, k % R, P)
def g2_mul(k, P=G2_GEN): return ec_mul(Field2, k % R, P)
def g1_add(P, S): return ec_add(Field1, P, S)
def g2_add(P, S): return ec_add(Field2, P, S)


def msm_naive(F, bases, scalars):
    acc = None
    for b, s in zip(bases, scalars):
        acc = ec_add(F, acc, ec_mul(F, s % R, b))
    return acc


# ---------------------------------------------------------------- limb / Montgomery helpers (the ABI layout)
def to_limbs(x, n=4):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def to_mont(x, p): return x * MONT_R % p
def from_mont(x, p): return x * inv(MONT_R % p, p) % p


# ---------------------------------------------------------------- evaluation domain (libfqfft basic_radix2_domain)
def omega(logn):
    assert logn <= FR_S
    return pow(FR_ROOT, 1 << (FR_S - logn), R)


def dft_naive(a, w):
    """O(n^2) evaluation: out[k] = sum_j a[j] w^(jk)  (definition of libfqfft FFT)."""
    n = len(a)
    return [sum(a[j] * pow(w, j * k, R) for j in range(n)) % R for k in range(n)]


def fft(a, w):
    """Recursive radix-2 (only for sizes where the naive DFT is too slow)."""
    n = len(a)
    if n == 1:
        return list(a)
    e = fft(a[0::2], w * w % R)
    o = fft(a[1::2], w * w % R)
    out = [0] * n
    t = 1
    for k in range(n // 2):
        v = t * o[k] % R
        out[k] = (e[k] + v) % R
        out[k + n // 2] = (e[k] - v) % R
        t = t * w % R
    return out


def domain_fft(a, inverse=False, coset=False, naive=False):
    """FFT / iFFT / cosetFFT / icosetFFT with coset generator g = FR_GEN, as
    libfqfft's basic_radix2_domain defines them."""
    n = len(a)
    logn = n.bit_length() - 1
    assert 1 << logn == n
    w = omega(logn)
    f = dft_naive if naive else fft
    if not inverse:
        if coset:
            a = [x * pow(FR_GEN, i, R) % R for i, x in enumerate(a)]
        return f(a, w)
    out = f(a, inv(w, R))
    ninv = inv(n, R)
    out = [x * ninv % R for x in out]
    if coset:
        ginv = inv(FR_GEN, R)
        out = [x * pow(ginv, i, R) % R for i, x in enumerate(out)]
    return out


# ---------------------------------------------------------------- evaluation-domain choice and step_radix2_domain
def ceil_log2(n):
    return (n - 1).bit_length() if n > 1 else 0


def evaluation_domain(min_size):
    """libfqfft get_evaluation_domain(min_size) for sizes up to 2^28 -> ('basic', m) or ('step', m).
    [UPSTREAM-RECALL] order of attempts: basic_radix2(min_size), extended_radix2(min_size) (only when
    log m = s + 1, never here), step_radix2(min_size), then the same three on big + rounded_small."""
    assert min_size > 1
    lg = ceil_log2(min_size)
    if min_size == 1 << lg:
        return ('basic', min_size)
    big = 1 << (lg - 1); small = min_size - big
    rounded_small = 1 << ceil_log2(small)
    if small == rounded_small:
        return ('step', min_size)
    if big == rounded_small:
        return ('basic', big + rounded_small)
    return ('step', big + rounded_small)


class Domain:
    """An evaluation domain as a LIST OF POINTS; every operation below is the textbook definition
    (evaluate / interpolate / product), not libfqfft's algorithm.
    basic_radix2_domain(m): x_i = omega_m^i.
    step_radix2_domain(m), m = big + small, big = 2^(ceil_log2(m)-1), small a power of two
    [UPSTREAM-RECALL get_domain_element]: x_i = (omega^2)^i for i < big, omega * omega_small^(i-big)
    after that, with omega = get_root_of_unity(2 big) and omega_small = get_root_of_unity(small)."""

    def __init__(self, kind, m):
        self.kind, self.m = kind, m
        if kind == 'basic':
            w = omega(ceil_log2(m))
            self.points = [pow(w, i, R) for i in range(m)]
        else:
            lg = ceil_log2(m)
            self.big = 1 << (lg - 1); self.small = m - self.big
            assert self.small == 1 << ceil_log2(self.small)
            w = omega(lg); ws = omega(ceil_log2(self.small))
            self.omega = w
            self.points = [pow(w, 2 * i, R) for i in range(self.big)] + [w * pow(ws, i, R) % R for i in range(self.small)]
        assert len(set(self.points)) == m

    @staticmethod
    def for_size(min_size):
        return Domain(*evaluation_domain(min_size))

    def Z(self, t):
        z = 1
        for x in self.points:
            z = z * (t - x) % R
        return z

    def Z_poly(self):
        """coefficients of prod (X - x_i), low to high (monic, degree m)."""
        c = [1]
        for x in self.points:
            nc = [0] * (len(c) + 1)
            for i, v in enumerate(c):
                nc[i + 1] = (nc[i + 1] + v) % R
                nc[i] = (nc[i] - v * x) % R
            c = nc
        return c

    def lagrange_at(self, t):
        out = []
        for i, xi in enumerate(self.points):
            num = den = 1
            for j, xj in enumerate(self.points):
                if j != i:
                    num = num * (t - xj) % R; den = den * (xi - xj) % R
            out.append(num * inv(den, R) % R)
        return out

    def evaluate(self, coeffs):           # FFT: coefficients -> values on the domain
        assert len(coeffs) == self.m
        out = []
        for x in self.points:
            acc = 0
            for c in reversed(coeffs):
                acc = (acc * x + c) % R
            out.append(acc)
        return out

    def interpolate(self, values):        # iFFT: values -> coefficients (degree < m)
        assert len(values) == self.m
        zp = self.Z_poly()
        out = [0] * self.m
        for xi, v in zip(self.points, values):
            # q = Z / (X - xi) by synthetic division; L_i = q / q(xi)
            q = [0] * self.m
            carry = 0
            for k in range(self.m, 0, -1):
                carry = (zp[k] + carry * xi) % R
                q[k - 1] = carry
            qx = 0
            for c in reversed(q):
                qx = (qx * xi + c) % R
            f = v * inv(qx, R) % R
            for k in range(self.m):
                out[k] = (out[k] + f * q[k]) % R
        return out

    def fft(self, a, inverse=False, coset=False):
        """FFT / iFFT / cosetFFT(g) / icosetFFT(g), g = Fr::multiplicative_generator, by definition."""
        if not inverse:
            if coset:
                a = [x * pow(FR_GEN, i, R) % R for i, x in enumerate(a)]
            return self.evaluate(a)
        out = self.interpolate(a)
        if coset:
            ginv = inv(FR_GEN, R)
            out = [x * pow(ginv, i, R) % R for i, x in enumerate(out)]
        return out


# ---------------------------------------------------------------- R1CS / QAP / Groth16
class R1CS:
    """rows: list of (a, b, c); each a dict {var_index: coeff}, index 0 = constant one,
    1..num_inputs = primary input, rest auxiliary."""

    def __init__(self, num_variables, num_inputs, rows):
        self.n, self.l, self.rows = num_variables, num_inputs, rows

    def domain(self):
        """libfqfft get_evaluation_domain(C + l + 1) as r1cs_to_qap_instance_map asks for it."""
        return Domain.for_size(len(self.rows) + self.l + 1)

    def is_satisfied(self, w):
        z = [1] + list(w)
        ev = lambda lc: sum(c * z[i] for i, c in lc.items()) % R
        return all(ev(a) * ev(b) % R == ev(c) for a, b, c in self.rows)

    def swapped_if_beneficial(self):
        ta, tb = set(), set()
        for a, b, _ in self.rows:
            ta.update(a); tb.update(b)
        if len(tb) > len(ta):
            return R1CS(self.n, self.l, [(b, a, c) for a, b, c in self.rows])
        return self


def lagrange_at(m, t):
    """u[i] = L_i(t) over the size-m radix-2 domain, from the closed form."""
    logm = m.bit_length() - 1
    w = omega(logm)
    z = (pow(t, m, R) - 1) % R
    out = []
    for i in range(m):
        wi = pow(w, i, R)
        out.append(z * wi % R * inv(m * (t - wi) % R, R) % R)
    return out


def qap_evaluate(cs, t):
    dom = cs.domain()
    m = dom.m
    u = lagrange_at(m, t) if dom.kind == 'basic' else dom.lagrange_at(t)
    At = [0] * (cs.n + 1); Bt = [0] * (cs.n + 1); Ct = [0] * (cs.n + 1)
    C = len(cs.rows)
    for i in range(cs.l + 1):
        At[i] = u[C + i]
    for i, (a, b, c) in enumerate(cs.rows):
        for k, v in a.items(): At[k] = (At[k] + u[i] * v) % R
        for k, v in b.items(): Bt[k] = (Bt[k] + u[i] * v) % R
        for k, v in c.items(): Ct[k] = (Ct[k] + u[i] * v) % R
    Zt = (pow(t, m, R) - 1) % R if dom.kind == 'basic' else dom.Z(t)
    return m, At, Bt, Ct, Zt


def groth16_setup(cs, t, alpha, beta, gamma, delta, k1=1, k2=1):
    """CRS with KNOWN trapdoor, shaped like libsnark's r1cs_gg_ppzksnark_generator
    (called at snark.cpp:91).  Returns (cs_used, crs dict of affine points)."""
    cs = cs.swapped_if_beneficial()
    m, At, Bt, Ct, Zt = qap_evaluate(cs, t)
    dinv, ginv = inv(delta, R), inv(gamma, R)
    g1 = g1_mul(k1); g2 = g2_mul(k2)
    crs = dict(
        m=m,
        alpha_g1=g1_mul(alpha, g1), beta_g1=g1_mul(beta, g1), delta_g1=g1_mul(delta, g1),
        beta_g2=g2_mul(beta, g2), delta_g2=g2_mul(delta, g2), gamma_g2=g2_mul(gamma, g2),
        A=[g1_mul(x, g1) for x in At],
        B1=[g1_mul(x, g1) for x in Bt],
        B2=[g2_mul(x, g2) for x in Bt],
        H=[g1_mul(pow(t, i, R) * Zt % R * dinv, g1) for i in range(m - 1)],
        L=[g1_mul((beta * At[i] + alpha * Bt[i] + Ct[i]) % R * dinv, g1) for i in range(cs.l + 1, cs.n + 1)],
        IC=[g1_mul((beta * At[i] + alpha * Bt[i] + Ct[i]) % R * ginv, g1) for i in range(cs.l + 1)],
        scalars=dict(At=At, Bt=Bt, Ct=Ct, Zt=Zt, k1=k1, k2=k2),
    )
    return cs, crs


def qap_witness_h(cs, w):
    """coefficients_for_H by interpolation and polynomial long division (definition, not FFTs)."""
    dom = cs.domain()
    m = dom.m
    C = len(cs.rows)
    z = [1] + list(w)
    ev = lambda lc: sum(c * z[i] for i, c in lc.items()) % R
    aA = [0] * m; aB = [0] * m; aC = [0] * m
    for i in range(cs.l + 1):
        aA[C + i] = z[i]
    for i, (a, b, c) in enumerate(cs.rows):
        aA[i] = (aA[i] + ev(a)) % R; aB[i] = ev(b); aC[i] = ev(c)
    if dom.kind == 'basic':
        pa = domain_fft(aA, inverse=True); pb = domain_fft(aB, inverse=True); pc = domain_fft(aC, inverse=True)
        zp = [R - 1] + [0] * (m - 1) + [1]
    else:
        pa = dom.interpolate(aA); pb = dom.interpolate(aB); pc = dom.interpolate(aC)
        zp = dom.Z_poly()
    prod = [0] * (2 * m - 1)
    for i, x in enumerate(pa):
        if x:
            for j, y in enumerate(pb):
                prod[i + j] = (prod[i + j] + x * y) % R
    for i, x in enumerate(pc):
        prod[i] = (prod[i] - x) % R
    # divide by the (monic) vanishing polynomial Z(X)
    h = [0] * (m - 1)
    rem = list(prod)
    for k in range(2 * m - 2, m - 1, -1):
        c = rem[k]
        h[k - m] = c
        if c:
            for j in range(m + 1):
                rem[k - m + j] = (rem[k - m + j] - c * zp[j]) % R
    assert all(v == 0 for v in rem), "witness does not satisfy the QAP"
    return h + [0, 0]        # m+1 coefficients like libsnark's coefficients_for_H


def groth16_prove(cs, crs, w, r, s):
    """Definition-level prover: returns affine (A in G1, B in G2, C in G1)."""
    h = qap_witness_h(cs, w)
    z = [1] + list(w)
    m = crs['m']
    At = msm_naive(Field1, crs['A'], z)
    Bt1 = msm_naive(Field1, crs['B1'], z)
    Bt2 = msm_naive(Field2, crs['B2'], z)
    Ht = msm_naive(Field1, crs['H'], h[:m - 1])
    Lt = msm_naive(Field1, crs['L'], z[cs.l + 1:])
    gA = g1_add(g1_add(crs['alpha_g1'], At), g1_mul(r, crs['delta_g1']))
    gB1 = g1_add(g1_add(crs['beta_g1'], Bt1), g1_mul(s, crs['delta_g1']))
    gB2 = g2_add(g2_add(crs['beta_g2'], Bt2), g2_mul(s, crs['delta_g2']))
    gC = g1_add(g1_add(Ht, Lt), g1_add(g1_mul(s, gA), g1_mul(r, gB1)))
    gC = g1_add(gC, ec_neg(Field1, g1_mul(r * s % R, crs['delta_g1'])))
    return gA, gB2, gC


def groth16_check_dlog(cs, crs, w, r, s, t, alpha, beta, gamma, delta, proof):
    """Known-trapdoor check: the proof points must equal [a]G1, [b]G2, [c]G1 with
    a = alpha + A(t) + r delta, b = beta + B(t) + s delta,
    c = (sum_{aux} w_i (beta A_i + alpha B_i + C_i)(t) + h(t) Z(t))/delta + s a + r b - r s delta."""
    sc = crs['scalars']
    z = [1] + list(w)
    A = sum(x * y for x, y in zip(sc['At'], z)) % R
    B = sum(x * y for x, y in zip(sc['Bt'], z)) % R
    h = qap_witness_h(cs, w)
    ht = sum(c * pow(t, i, R) for i, c in enumerate(h)) % R
    a = (alpha + A + r * delta) % R
    b = (beta + B + s * delta) % R
    aux = sum(z[i] * (beta * sc['At'][i] + alpha * sc['Bt'][i] + sc['Ct'][i]) for i in range(cs.l + 1, cs.n + 1)) % R
    c = ((aux + ht * sc['Zt']) * inv(delta, R) + s * a + r * b - r * s * delta) % R
    k1, k2 = sc['k1'], sc['k2']
    return proof == (g1_mul(a * k1), g2_mul(b * k2), g1_mul(c * k1))


# ---------------------------------------------------------------- serialisation (libsnark defaults)
def ser_fq(x):
    """Fp operator<< with BINARY_OUTPUT + MONTGOMERY_OUTPUT: 4 LE u64 limbs of x*R mod q."""
    return to_mont(x, Q).to_bytes(32, 'little')


def ser_g1(P):
    if P is None:                       # to_affine_coordinates() of zero is (0, 1, 0)
        return b'1' + ser_fq(0) + b'1'
    return b'0' + ser_fq(P[0]) + (b'1' if P[1] & 1 else b'0')


def ser_g2(P):
    if P is None:
        return b'1' + ser_fq(0) + ser_fq(0) + b'1'
    return b'0' + ser_fq(P[0][0]) + ser_fq(P[0][1]) + (b'1' if P[1][0] & 1 else b'0')


def ser_proof(proof):
    """r1cs_gg_ppzksnark_proof operator<<: g_A, g_B, g_C, each followed by OUTPUT_NEWLINE
    (empty under BINARY_OUTPUT) -> 34 + 66 + 34 = 134 bytes."""
    return ser_g1(proof[0]) + ser_g2(proof[1]) + ser_g1(proof[2])


# ---------------------------------------------------------------- deterministic RNG used by benches/tests
class SplitMix64:
    def __init__(self, seed): self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        while True:
            v = from_limbs([self.next() for _ in range(4)]) & ((1 << 254) - 1)
            if v < R:
                return v
