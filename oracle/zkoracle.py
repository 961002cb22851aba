"""ctypes binding of the CPU oracle (oracle/libzkoracle.so).

TEST INFRASTRUCTURE ONLY — see the header of oracle/zkoracle.cpp.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from zklaim_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzkoracle.so")


def build(force=False):
    src = os.path.join(_HERE, "zkoracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libzkoracle.so"])
    return _SO


class R1CS(C.Structure):
    _fields_ = [("num_variables", C.c_uint32), ("num_inputs", C.c_uint32), ("num_constraints", C.c_uint32), ("reserved", C.c_uint32)] + \
        [(f"{m}_{f}", C.c_void_p) for m in "abc" for f in ("rowptr", "col", "val")]


class PK(C.Structure):
    _fields_ = [("cs", R1CS), ("log_m", C.c_uint32), ("domain_size", C.c_uint32)] + \
        [(k, C.c_void_p) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "A_query", "B_g1", "B_g2", "H_query", "L_query")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.zko_evaluation_domain_size.restype = C.c_size_t
        _lib.zko_evaluation_domain_size.argtypes = [C.c_size_t]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---- wrappers (all arrays are np.uint64, shapes (..., limbs)) -------------------------------
def fp_op(field, op, a, b=None):
    out = np.zeros(4, np.uint64)
    a = u64(a); b = None if b is None else u64(b)
    assert lib().zko_fp_op(field, op, _p(a), _p(b), _p(out)) == 0
    return out


def fq2_op(op, a, b=None):
    out = np.zeros(8, np.uint64)
    a = u64(a); b = None if b is None else u64(b)
    assert lib().zko_fq2_op(op, _p(a), _p(b), _p(out)) == 0
    return out


def g1_generator():
    o = np.zeros(8, np.uint64); lib().zko_g1_generator(_p(o)); return o


def g2_generator():
    o = np.zeros(16, np.uint64); lib().zko_g2_generator(_p(o)); return o


def _binop(name, a, b, n):
    o = np.zeros(n, np.uint64); a = u64(a); b = u64(b)
    assert getattr(lib(), name)(_p(a), _p(b), _p(o)) == 0
    return o


def g1_add(a, b): return _binop("zko_g1_add", a, b, 12)
def g2_add(a, b): return _binop("zko_g2_add", a, b, 24)
def g1_scalar_mul(base, k): return _binop("zko_g1_scalar_mul", base, k, 12)
def g2_scalar_mul(base, k): return _binop("zko_g2_scalar_mul", base, k, 24)
def g1_on_curve(a): return bool(lib().zko_g1_on_curve(_p(u64(a))))
def g2_on_curve(a): return bool(lib().zko_g2_on_curve(_p(u64(a))))


def g1_sum(pts):
    pts = u64(pts); o = np.zeros(12, np.uint64)
    lib().zko_g1_sum(_p(pts), C.c_size_t(pts.size // 12), _p(o)); return o


def g2_sum(pts):
    pts = u64(pts); o = np.zeros(24, np.uint64)
    lib().zko_g2_sum(_p(pts), C.c_size_t(pts.size // 24), _p(o)); return o


def g1_fixed_base(base, scalars):
    scalars = u64(scalars); n = scalars.size // 4
    o = np.zeros((n, 8), np.uint64)
    lib().zko_g1_fixed_base(_p(u64(base)), _p(scalars), C.c_size_t(n), _p(o)); return o


def g2_fixed_base(base, scalars):
    scalars = u64(scalars); n = scalars.size // 4
    o = np.zeros((n, 16), np.uint64)
    lib().zko_g2_fixed_base(_p(u64(base)), _p(scalars), C.c_size_t(n), _p(o)); return o


def fft(a, inverse=False, coset=False):
    """FFT / iFFT / cosetFFT / icosetFFT on get_evaluation_domain(n): n a power of two (basic_radix2_domain) or
    2^a + 2^b (step_radix2_domain)"""
    a = u64(a).copy(); n = a.size // 4
    logn = n.bit_length() - 1
    if 1 << logn == n:
        assert lib().zko_fft(_p(a), logn, int(inverse), int(coset)) == 0
    else:
        assert lib().zko_domain_fft(_p(a), C.c_size_t(n), int(inverse), int(coset)) == 0, "not a step_radix2 size"
    return a.reshape(n, 4)


def domain_lagrange(m, t_canonical):
    """evaluate_all_lagrange_polynomials(t) and compute_vanishing_polynomial(t) -> ((m,4) Montgomery, (4,) Montgomery)"""
    out = np.zeros((m, 4), np.uint64); z = np.zeros(4, np.uint64)
    assert lib().zko_domain_lagrange(C.c_size_t(m), _p(u64(t_canonical)), _p(out), _p(z)) == 0
    return out, z


NAIVE, BDLO12, MIXED = 0, 1, 2


def msm_g1(bases, scalars, method=BDLO12, chunks=1):
    bases = u64(bases); scalars = u64(scalars); o = np.zeros(12, np.uint64)
    assert lib().zko_msm_g1(_p(bases), _p(scalars), C.c_size_t(scalars.size // 4), _p(o), method, chunks) == 0
    return o


def msm_g2(bases, scalars, method=BDLO12, chunks=1):
    bases = u64(bases); scalars = u64(scalars); o = np.zeros(24, np.uint64)
    assert lib().zko_msm_g2(_p(bases), _p(scalars), C.c_size_t(scalars.size // 4), _p(o), method, chunks) == 0
    return o


def evaluation_domain_size(min_size):
    return int(lib().zko_evaluation_domain_size(min_size))


def evaluation_domain_is_step(min_size):
    return bool(lib().zko_evaluation_domain_is_step(C.c_size_t(min_size)))


def make_r1cs(n, l, A, B, Cm, keep):
    """A, B, Cm: (rowptr uint32[C+1], col uint32[nnz], val uint64[nnz,4]).  `keep` collects array refs."""
    cs = R1CS()
    cs.num_variables, cs.num_inputs, cs.num_constraints = n, l, len(A[0]) - 1
    for name, (rp, col, val) in zip("abc", (A, B, Cm)):
        rp = np.ascontiguousarray(rp, np.uint32); col = np.ascontiguousarray(col, np.uint32); val = u64(val)
        keep += [rp, col, val]
        setattr(cs, f"{name}_rowptr", rp.ctypes.data); setattr(cs, f"{name}_col", col.ctypes.data); setattr(cs, f"{name}_val", val.ctypes.data)
    return cs


def r1cs_is_satisfied(cs, w):
    return bool(lib().zko_r1cs_is_satisfied(C.byref(cs), _p(u64(w))))


def qap_witness_h(cs, w, m):
    out = np.zeros((m + 1, 4), np.uint64)
    assert lib().zko_qap_witness_h(C.byref(cs), _p(u64(w)), _p(out)) == 0
    return out


def groth16_setup(cs, td_canonical):
    """td: 5x4 canonical limbs (t, alpha, beta, gamma, delta).  Returns dict of arrays + a PK struct."""
    n, l = cs.num_variables, cs.num_inputs
    m = evaluation_domain_size(cs.num_constraints + l + 1)
    assert m, "no radix-2 or step domain of that size"
    z = lambda *s: np.zeros(s, np.uint64)
    d = dict(alpha_g1=z(8), beta_g1=z(8), delta_g1=z(8), beta_g2=z(16), delta_g2=z(16), A_query=z(n + 1, 8), B_g1=z(n + 1, 8),
             B_g2=z(n + 1, 16), H_query=z(m - 1, 8), L_query=z(n - l, 8), At=z(n + 1, 4), Bt=z(n + 1, 4), Ct=z(n + 1, 4), Zt=z(4))
    td = u64(td_canonical)
    rc = lib().zko_groth16_setup(C.byref(cs), _p(td), *[_p(d[k]) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "A_query",
                                                                           "B_g1", "B_g2", "H_query", "L_query", "At", "Bt", "Ct", "Zt")])
    assert rc == 0
    d["m"] = m
    return d


def make_pk(cs, crs):
    pk = PK()
    pk.cs = cs
    pk.log_m = (crs["m"] - 1).bit_length()
    pk.domain_size = crs["m"]
    for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "A_query", "B_g1", "B_g2", "H_query", "L_query"):
        setattr(pk, k, crs[k].ctypes.data)
    return pk


def groth16_prove(pk, w, r, s, check_satisfied=True, chunks=1):
    out = np.zeros(256, np.uint8); ln = C.c_size_t(0)
    rc = lib().zko_groth16_prove(C.byref(pk), _p(u64(w)), _p(u64(r)), _p(u64(s)), int(check_satisfied), _p(out), C.byref(ln), chunks)
    return rc, bytes(out[:ln.value])


def pk_write_blob(pk):
    lib().zko_pk_write_blob.restype = C.c_size_t
    need = lib().zko_pk_write_blob(C.byref(pk), None, C.c_size_t(0))
    out = np.zeros(need, np.uint8)
    got = lib().zko_pk_write_blob(C.byref(pk), _p(out), C.c_size_t(need))
    assert got == need
    return out.tobytes()


def num_threads():
    return lib().zko_num_threads()
