"""ctypes binding of include/zkg.h (the drop-in boundary below snark.cpp:126 of the reference).

Array conventions are those of zkg.h: numpy uint64 arrays of little-endian limbs, Montgomery form
unless stated, G1 affine 8 limbs, G2 affine 16 limbs, normalised "jac" outputs 12 / 24 limbs.
*_dev functions take raw device pointers (e.g. ``torch.Tensor.data_ptr()``) and a HIP stream handle.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzkg.so")

DECLARED_SYMBOLS = [
    "zkg_init", "zkg_shutdown", "zkg_last_error", "zkg_device_info", "zkg_ntt", "zkg_ntt_dev", "zkg_evaluation_domain_size", "zkg_ntt_domain", "zkg_ntt_domain_dev", "zkg_msm_g1", "zkg_msm_g2",
    "zkg_msm_g1_dev", "zkg_msm_g2_dev", "zkg_msm_g1_windows_dev", "zkg_g1_sum", "zkg_g2_sum", "zkg_g1_fixed_base_dev", "zkg_g2_fixed_base_dev",
    "zkg_crs_upload", "zkg_crs_upload_blob", "zkg_pk_blob_inspect", "zkg_crs_free", "zkg_crs_num_variables", "zkg_groth16_prove", "zkg_groth16_prove_sparse", "zkg_circuit_sparse_witness", "zkg_qap_witness_h", "zkg_prove_stage_ms", "zkg_timing_reset",
    "zkg_timing_dominant_ms", "zkg_zklaim_circuit_new", "zkg_zklaim_witness_new", "zkg_circuit_num_variables", "zkg_circuit_free", "zkg_circuit_r1cs", "zkg_circuit_witness",
    "zkg_circuit_is_satisfied", "zkg_circuit_first_unsatisfied", "zkg_zklaim_input_map", "zkg_groth16_setup", "zkg_keypair_free",
    "zkg_keypair_pk", "zkg_keypair_swapped", "zkg_keypair_pk_blob", "zkg_keypair_vk_blob", "zkg_groth16_verify", "zkg_pairing_probe", "zkg_pairing_selfcheck",
    "zkg_compat_reset", "zkg_field_op", "zkg_init_multi", "zkg_msm_g1_shards_upload", "zkg_msm_g1_shards_free", "zkg_msm_g1_shards_count",
    "zkg_msm_g1_multi", "zkg_g1_add_quad29", "zkg_crs_shard_h", "zkg_msm_g1_bases_upload", "zkg_msm_g1_resident", "zkg_msm_g1_bases_free",
    "zkg_prover_peak_in_flight", "zkg_msm_g1_host_scalars", "zkg_multi_rccl_calls", "zkg_g1_add_pair29",
]
# the reference's own seam, exported with its original names (zklaim.h:257-259)
COMPAT_SYMBOLS = ["libsnark_trusted_setup", "libsnark_prove", "libsnark_verify"]


OK, ERROR, UNSATISFIED = 0, 1, 2          # include/zkg.h


class ZkgError(RuntimeError):
    pass


class R1CS(C.Structure):
    _fields_ = [("num_variables", C.c_uint32), ("num_inputs", C.c_uint32), ("num_constraints", C.c_uint32), ("reserved", C.c_uint32)] + \
        [(f"{m}_{f}", C.c_void_p) for m in "abc" for f in ("rowptr", "col", "val")]


class PK(C.Structure):
    _fields_ = [("cs", R1CS), ("log_m", C.c_uint32), ("domain_size", C.c_uint32)] + \
        [(k, C.c_void_p) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "A_query", "B_g1", "B_g2", "H_query", "L_query")]


_lib = None


def lib():
    """Loads libzkg.so.  Fails loudly when it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ZkgError(f"{_SO} is missing: build it with `python -m zklaim_amd.build` (hipcc, gfx950)")
        # One ROCm stack per process: PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  If libzkg.so pulled in the
        # system copies first and torch loaded afterwards, two HSA runtimes would coexist and device discovery fails.  When torch
        # is installed, let it load its runtime first; libzkg.so's libamdhip64.so.7 dependency then resolves to the same objects.
        if "torch" not in sys.modules and not os.environ.get("ZKG_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = C.CDLL(_SO)
        _lib.zkg_last_error.restype = C.c_char_p
        _lib.zkg_timing_dominant_ms.restype = C.c_float
        _lib.zkg_crs_upload.restype = C.c_void_p
        _lib.zkg_crs_upload.argtypes = [C.c_void_p]
        _lib.zkg_crs_upload_blob.restype = C.c_void_p
        _lib.zkg_crs_upload_blob.argtypes = [C.c_void_p, C.c_size_t]
        _lib.zkg_crs_free.argtypes = [C.c_void_p]
    return _lib


def _check(rc, what):
    if rc != 0:
        raise ZkgError(f"{what} failed (rc={rc}): {lib().zkg_last_error().decode()}")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _vp(x):
    return C.c_void_p(int(x) if x else 0)


_initialised = False


def init(device=0):
    global _initialised
    _check(lib().zkg_init(int(device)), "zkg_init")
    _initialised = True


def shutdown():
    global _initialised
    if _initialised:
        lib().zkg_shutdown()
    _initialised = False


def device_info():
    name = C.create_string_buffer(128); cus = C.c_int(0)
    _check(lib().zkg_device_info(name, C.c_size_t(128), C.byref(cus)), "zkg_device_info")
    return name.value.decode(), cus.value


def field_op(field, op, a, b=None):
    """element-wise device arithmetic (zkg_field_op): field 0 Fq, 1 Fr, 2 Fq2; op 0 mul 1 add 2 sub 3 inv 4 to_mont 5 from_mont 6 neg 7 sqr;
    Fq only: 10-14 the 29-bit representation of the accumulation kernel (mul, add, sub, zero test, composite)"""
    a = _u64(a); limbs = 8 if field == 2 else 4
    out = np.zeros_like(a)
    bb = None if b is None else _u64(b)
    _check(lib().zkg_field_op(int(field), int(op), _p(a), _p(bb), C.c_size_t(a.size // limbs), _p(out)), "zkg_field_op")
    return out


def g1_add_quad29(a_jac, b_jac, chain=0):
    """out[i] = a[i] + b[i] (then `chain` rounds of x <- 2x + b[i]) on the GPU through the 29-bit quad addition of the reduction kernels"""
    a = _u64(a_jac); b = _u64(b_jac); out = np.zeros_like(a)
    _check(lib().zkg_g1_add_quad29(_p(a), _p(b), C.c_size_t(a.size // 12), int(chain), _p(out)), "zkg_g1_add_quad29")
    return out.reshape(-1, 12)


def g1_add_pair29(a_jac, b_jac, chain=0):
    """the same through the pair form of the addition (xyzz29_add_pair, the bucket reduction's since round 4)"""
    a = _u64(a_jac); b = _u64(b_jac); out = np.zeros_like(a)
    _check(lib().zkg_g1_add_pair29(_p(a), _p(b), C.c_size_t(a.size // 12), int(chain), _p(out)), "zkg_g1_add_pair29")
    return out.reshape(-1, 12)


# ---- NTT (libfqfft basic_radix2_domain FFT/iFFT/cosetFFT/icosetFFT) -------------------------------
def ntt(a, inverse=False, coset=False):
    """FFT / iFFT / cosetFFT / icosetFFT on the domain get_evaluation_domain(len(a)) names: a power of two
    (basic_radix2_domain, zkg_ntt) or 2^a + 2^b (step_radix2_domain, zkg_ntt_domain)"""
    a = _u64(a).copy(); n = a.size // 4
    logn = n.bit_length() - 1
    if n == 0:
        raise ZkgError("ntt: empty input")
    if (1 << logn) == n:
        _check(lib().zkg_ntt(_p(a), C.c_uint(logn), int(inverse), int(coset)), "zkg_ntt")
    else:
        _check(lib().zkg_ntt_domain(_p(a), C.c_size_t(n), int(inverse), int(coset)), "zkg_ntt_domain")
    return a.reshape(n, 4)


def pk_blob_inspect(blob):
    """host-only walk of a pk blob (zkg_pk_blob_inspect): dict of its sizes, or ZkgError"""
    out = np.zeros(8, np.uint64)
    buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob) if len(blob) else None
    L = lib()
    L.zkg_pk_blob_inspect.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    _check(L.zkg_pk_blob_inspect(C.cast(buf, C.c_void_p) if buf is not None else None, C.c_size_t(len(blob)), _p(out)), "zkg_pk_blob_inspect")
    names = ("A_query", "B_values", "H_query", "L_query", "num_inputs", "num_constraints", "terms", "domain_size")
    return {k: int(v) for k, v in zip(names, out)}


def evaluation_domain_size(min_size):
    """libfqfft get_evaluation_domain(min_size) -> (m, is_step)"""
    m = C.c_size_t(0); st = C.c_int(0)
    _check(lib().zkg_evaluation_domain_size(C.c_size_t(min_size), C.byref(m), C.byref(st)), "zkg_evaluation_domain_size")
    return m.value, bool(st.value)


def ntt_dev(d_ptr, logn, inverse=False, coset=False, stream=0):
    _check(lib().zkg_ntt_dev(_vp(d_ptr), C.c_uint(logn), int(inverse), int(coset), _vp(stream)), "zkg_ntt_dev")


# ---- MSM (libff multi_exp / multi_exp_with_mixed_addition) -----------------------------------------
def msm_g1(bases, scalars):
    bases = _u64(bases); scalars = _u64(scalars); out = np.zeros(12, np.uint64)
    _check(lib().zkg_msm_g1(_p(bases), _p(scalars), C.c_size_t(scalars.size // 4), _p(out)), "zkg_msm_g1")
    return out


def msm_g2(bases, scalars):
    bases = _u64(bases); scalars = _u64(scalars); out = np.zeros(24, np.uint64)
    _check(lib().zkg_msm_g2(_p(bases), _p(scalars), C.c_size_t(scalars.size // 4), _p(out)), "zkg_msm_g2")
    return out


def init_multi(devices):
    global _initialised
    d = (C.c_int * len(devices))(*devices)
    _check(lib().zkg_init_multi(d, len(devices)), "zkg_init_multi")
    _initialised = True


class MsmShards:
    """G1 bases sharded by points over several devices of this process (zkg_msm_g1_shards_upload); msm() = zkg_msm_g1_multi"""

    def __init__(self, bases, devices):
        bases = _u64(bases)
        L = lib()
        L.zkg_msm_g1_shards_upload.restype = C.c_void_p
        L.zkg_msm_g1_shards_upload.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.zkg_msm_g1_shards_free.argtypes = [C.c_void_p]
        L.zkg_msm_g1_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        d = (C.c_int * len(devices))(*devices)
        self.n = bases.size // 8; self.ndev = len(devices)
        self._h = L.zkg_msm_g1_shards_upload(_p(bases), self.n, d, len(devices))
        if not self._h:
            raise ZkgError("zkg_msm_g1_shards_upload failed: " + L.zkg_last_error().decode())

    def msm(self, scalars, with_partials=False):
        scalars = _u64(scalars); out = np.zeros(12, np.uint64); parts = np.zeros((self.ndev, 12), np.uint64)
        assert scalars.size // 4 == self.n
        _check(lib().zkg_msm_g1_multi(C.c_void_p(self._h), _p(scalars), _p(out), _p(parts)), "zkg_msm_g1_multi")
        return (out, parts) if with_partials else out

    def free(self):
        if self._h:
            lib().zkg_msm_g1_shards_free(C.c_void_p(self._h)); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


SCALARS_MONT, SCALARS_MOSTLY_BITS = 1, 2


def msm_g1_dev(d_bases, d_scalars, n, scalars_mont=False, stream=0, mostly_bits=False):
    out = np.zeros(12, np.uint64)
    flags = (SCALARS_MONT if scalars_mont else 0) | (SCALARS_MOSTLY_BITS if mostly_bits else 0)
    _check(lib().zkg_msm_g1_dev(_vp(d_bases), _vp(d_scalars), C.c_size_t(n), flags, _p(out), _vp(stream)), "zkg_msm_g1_dev")
    return out


def msm_g1_host_scalars(d_bases, scalars_host_ptr, n, scalars_mont=False, stream=0):
    """zkg_msm_g1_host_scalars: bases resident (device pointer), scalars at a HOST address (pinned memory lets the upload overlap the work)"""
    out = np.zeros(12, np.uint64)
    lib().zkg_msm_g1_host_scalars.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    _check(lib().zkg_msm_g1_host_scalars(_vp(d_bases), _vp(scalars_host_ptr), C.c_size_t(n), SCALARS_MONT if scalars_mont else 0, _p(out), _vp(stream)), "zkg_msm_g1_host_scalars")
    return out


class ResidentBases:
    """zkg_msm_g1_bases_upload / zkg_msm_g1_resident: fixed G1 bases (device pointer, n affine points) kept with their per-window tables"""

    def __init__(self, d_bases, n):
        lib().zkg_msm_g1_bases_upload.restype = C.c_void_p
        lib().zkg_msm_g1_bases_upload.argtypes = [C.c_void_p, C.c_size_t]
        self._h = lib().zkg_msm_g1_bases_upload(_vp(d_bases), C.c_size_t(n))
        if not self._h:
            raise ZkgError("zkg_msm_g1_bases_upload failed: " + last_error())
        self.n = n

    def msm(self, d_scalars, scalars_mont=False, stream=0):
        """`stream`: the HIP stream whose queued work produced d_scalars (0 = the null stream); the job is ordered behind it"""
        out = np.zeros(12, np.uint64)
        lib().zkg_msm_g1_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        _check(lib().zkg_msm_g1_resident(C.c_void_p(self._h), _vp(d_scalars), C.c_size_t(self.n), SCALARS_MONT if scalars_mont else 0, _p(out), _vp(stream)), "zkg_msm_g1_resident")
        return out

    def free(self):
        if self._h:
            lib().zkg_msm_g1_bases_free.argtypes = [C.c_void_p]
            lib().zkg_msm_g1_bases_free(C.c_void_p(self._h)); self._h = None


def msm_g1_windows_dev(d_bases, d_scalars, n, first_window, window_stride, scalars_mont=False, stream=0):
    """partial MSM over the Pippenger windows first_window, first_window + window_stride, ... (window-sharded multi-GPU variant)"""
    out = np.zeros(12, np.uint64)
    _check(lib().zkg_msm_g1_windows_dev(_vp(d_bases), _vp(d_scalars), C.c_size_t(n), int(scalars_mont), C.c_uint(first_window), C.c_uint(window_stride),
                                        _p(out), _vp(stream)), "zkg_msm_g1_windows_dev")
    return out


def msm_g2_dev(d_bases, d_scalars, n, scalars_mont=False, stream=0):
    out = np.zeros(24, np.uint64)
    _check(lib().zkg_msm_g2_dev(_vp(d_bases), _vp(d_scalars), C.c_size_t(n), int(scalars_mont), _p(out), _vp(stream)), "zkg_msm_g2_dev")
    return out


def g1_sum(points_jac):
    pts = _u64(points_jac); out = np.zeros(12, np.uint64)
    _check(lib().zkg_g1_sum(_p(pts), C.c_size_t(pts.size // 12), _p(out)), "zkg_g1_sum")
    return out


def g2_sum(points_jac):
    pts = _u64(points_jac); out = np.zeros(24, np.uint64)
    _check(lib().zkg_g2_sum(_p(pts), C.c_size_t(pts.size // 24), _p(out)), "zkg_g2_sum")
    return out


def fixed_base_g1_dev(base, d_scalars, n, d_out, stream=0):
    base = _u64(base)
    _check(lib().zkg_g1_fixed_base_dev(_p(base), _vp(d_scalars), C.c_size_t(n), _vp(d_out), _vp(stream)), "zkg_g1_fixed_base_dev")


def fixed_base_g2_dev(base, d_scalars, n, d_out, stream=0):
    base = _u64(base)
    _check(lib().zkg_g2_fixed_base_dev(_p(base), _vp(d_scalars), C.c_size_t(n), _vp(d_out), _vp(stream)), "zkg_g2_fixed_base_dev")


def timing_reset():
    lib().zkg_timing_reset()


def timing_dominant_ms():
    n = C.c_int(0)
    ms = lib().zkg_timing_dominant_ms(C.byref(n))
    return float(ms), n.value


# ---- Groth16 (r1cs_gg_ppzksnark_prover, snark.cpp:126) ---------------------------------------------
def make_r1cs(n, l, A, B, Cm, keep):
    """A, B, Cm = (rowptr uint32[C+1], col uint32[nnz], val uint64[nnz,4] Montgomery Fr); `keep` pins the arrays."""
    cs = R1CS()
    cs.num_variables, cs.num_inputs, cs.num_constraints = n, l, len(A[0]) - 1
    for name, (rp, col, val) in zip("abc", (A, B, Cm)):
        rp = np.ascontiguousarray(rp, np.uint32); col = np.ascontiguousarray(col, np.uint32); val = _u64(val)
        keep += [rp, col, val]
        setattr(cs, f"{name}_rowptr", rp.ctypes.data); setattr(cs, f"{name}_col", col.ctypes.data); setattr(cs, f"{name}_val", val.ctypes.data)
    return cs


def make_pk(cs, arrays, log_m, keep, domain_size=0):
    """domain_size: m when it is a step_radix2 size 2^(log_m-1) + 2^b; 0 for m = 2^log_m"""
    pk = PK(); pk.cs = cs; pk.log_m = log_m; pk.domain_size = domain_size
    for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "A_query", "B_g1", "B_g2", "H_query", "L_query"):
        a = _u64(arrays[k]); keep.append(a)
        setattr(pk, k, a.ctypes.data)
    return pk


class Crs:
    """Device-resident proving key (zkg_crs_upload): parsed once, reused for every proof."""

    def __init__(self, pk=None, blob=None, m=None):
        """from a zkg_pk struct of flat arrays, or from a libsnark pk byte blob (ctx->pk)"""
        if blob is not None:
            buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
            self._h = lib().zkg_crs_upload_blob(C.cast(buf, C.c_void_p), C.c_size_t(len(blob)))
            self.m = m
        else:
            self._h = lib().zkg_crs_upload(C.byref(pk))
            self.m = pk.domain_size or (1 << pk.log_m)
        if not self._h:
            raise ZkgError("zkg_crs_upload failed: " + lib().zkg_last_error().decode())

    def prove(self, witness, r, s, check_satisfied=True):
        """-> (rc, proof bytes); rc == UNSATISFIED (2): the gate of snark.cpp:121-124 refused the witness (libsnark_prove returns 1 there)."""
        out = np.zeros(256, np.uint8); ln = C.c_size_t(0)
        rc = lib().zkg_groth16_prove(C.c_void_p(self._h), _p(_u64(witness)), _p(_u64(r)), _p(_u64(s)), int(check_satisfied), _p(out), C.byref(ln))
        if rc not in (OK, UNSATISFIED):
            _check(rc, "zkg_groth16_prove")
        return rc, bytes(out[:ln.value])

    def prove_sparse(self, tags, full_index, full_values, r, s, check_satisfied=True):
        """zkg_groth16_prove_sparse: tags uint8[n] (0 zero, 1 one, 2 listed), listed variables as (index, 4 Montgomery limbs)"""
        tags = np.ascontiguousarray(tags, np.uint8); full_index = np.ascontiguousarray(full_index, np.uint32); full_values = _u64(full_values)
        out = np.zeros(256, np.uint8); ln = C.c_size_t(0)
        rc = lib().zkg_groth16_prove_sparse(C.c_void_p(self._h), _p(tags), _p(full_index), _p(full_values), C.c_size_t(full_index.size), _p(_u64(r)), _p(_u64(s)),
                                            int(check_satisfied), _p(out), C.byref(ln))
        if rc not in (OK, UNSATISFIED):
            _check(rc, "zkg_groth16_prove_sparse")
        return rc, bytes(out[:ln.value])

    def shard_h(self, devices):
        """zkg_crs_shard_h: the H query's tables sharded by points over `devices` (one GPU listed several times rehearses the path)"""
        d = (C.c_int * len(devices))(*devices)
        lib().zkg_crs_shard_h.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _check(lib().zkg_crs_shard_h(C.c_void_p(self._h), d, len(devices)), "zkg_crs_shard_h")

    def qap_witness_h(self, witness):
        out = np.zeros((self.m + 1, 4), np.uint64)
        _check(lib().zkg_qap_witness_h(C.c_void_p(self._h), _p(_u64(witness)), _p(out)), "zkg_qap_witness_h")
        return out

    def stage_ms(self):
        ms = (C.c_float * 8)()
        _check(lib().zkg_prove_stage_ms(C.c_void_p(self._h), ms), "zkg_prove_stage_ms")
        return list(ms)

    def free(self):
        if self._h:
            lib().zkg_crs_free(C.c_void_p(self._h)); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- zklaim front-end structures (include/zklaim_abi.h) and the credential circuit --------------------------------
class ZklaimPayload(C.Structure):
    _fields_ = [("data_ref", C.c_uint64 * 5), ("data_op", C.c_int * 5), ("salt", C.c_uint64), ("hash", C.c_ubyte * 32), ("priv", C.c_uint8),
                ("pre", C.c_ubyte * 48)]


class ZklaimWrapPayload(C.Structure):
    pass


ZklaimWrapPayload._fields_ = [("next", C.POINTER(ZklaimWrapPayload)), ("pl", ZklaimPayload)]


class ZklaimCtx(C.Structure):
    _fields_ = [("num_of_payloads", C.c_size_t), ("pl_ctx_head", C.POINTER(ZklaimWrapPayload)), ("pk_size", C.c_size_t), ("pk", C.c_void_p),
                ("vk_size", C.c_size_t), ("vk", C.c_void_p), ("proof_size", C.c_size_t), ("proof", C.c_void_p), ("pub_key", C.c_ubyte * 32),
                ("signature", C.c_ubyte * 64)]


OPS = dict(less=1, less_or_eq=3, eq=2, greater_or_eq=10, greater=8, not_eq=9, noop=99)


def make_ctx(payloads, keep):
    """payloads: list of dicts(attrs=[5 u64], refs=[5 u64], ops=[5 names], salt=u64, hash=bytes|None).  hash None -> SHA-256(pre),
    as zklaim_hash_pl computes it (zklaim.c:114-121)."""
    import hashlib
    import struct
    ctx = ZklaimCtx()
    ctx.num_of_payloads = len(payloads)
    nodes = []
    for p in payloads:
        node = ZklaimWrapPayload()
        pre = struct.pack("<5QQ", *p["attrs"], p.get("salt", 0))
        node.pl.pre[:] = list(pre)
        for j in range(5):
            node.pl.data_ref[j] = p["refs"][j]
            node.pl.data_op[j] = OPS[p["ops"][j]]
        node.pl.salt = p.get("salt", 0)
        h = p.get("hash") or hashlib.sha256(pre).digest()
        node.pl.hash[:] = list(h)
        nodes.append(node)
    for a, b in zip(nodes, nodes[1:]):
        a.next = C.pointer(b)
    if nodes:
        ctx.pl_ctx_head = C.pointer(nodes[0])
    keep.append(nodes)
    return ctx


class ZklaimCircuit:
    """R1CS (+ witness) of zklaim_gadget for a zklaim_ctx, built on the host by libzkg.so"""

    def __init__(self, ctx, with_witness=True, witness_only=False, reference_quirk=False):
        L = lib()
        L.zkg_zklaim_circuit_new.restype = C.c_void_p
        L.zkg_zklaim_circuit_new.argtypes = [C.c_void_p, C.c_int]
        L.zkg_zklaim_witness_new.restype = C.c_void_p
        L.zkg_zklaim_witness_new.argtypes = [C.c_void_p]
        L.zkg_circuit_num_variables.restype = C.c_uint32
        L.zkg_circuit_num_variables.argtypes = [C.c_void_p]
        L.zkg_circuit_witness.restype = C.c_void_p
        L.zkg_circuit_first_unsatisfied.restype = C.c_long
        for f in (L.zkg_circuit_free, L.zkg_circuit_witness, L.zkg_circuit_is_satisfied, L.zkg_circuit_first_unsatisfied):
            f.argtypes = [C.c_void_p]
        L.zkg_circuit_r1cs.argtypes = [C.c_void_p, C.c_void_p]
        if witness_only:
            self._h = L.zkg_zklaim_witness_new(C.cast(C.pointer(ctx), C.c_void_p))
        else:
            self._h = L.zkg_zklaim_circuit_new(C.cast(C.pointer(ctx), C.c_void_p), (1 if with_witness else 0) | (2 if reference_quirk else 0))
        if not self._h:
            raise ZkgError("zkg_zklaim_circuit_new failed: " + L.zkg_last_error().decode())
        self.r1cs = R1CS()
        _check(L.zkg_circuit_r1cs(self._h, C.byref(self.r1cs)), "zkg_circuit_r1cs")
        if witness_only:
            self.r1cs.num_variables = L.zkg_circuit_num_variables(self._h)
        self.with_witness = with_witness

    def witness(self):
        p = lib().zkg_circuit_witness(self._h)
        if not p:
            return None
        n = self.r1cs.num_variables
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, 4)).copy()

    def sparse_witness(self):
        """-> (tags uint8[n], full_index uint32[count], full_values uint64[count,4]): the witness as zkg_groth16_prove_sparse takes it"""
        tags = C.POINTER(C.c_uint8)(); idx = C.POINTER(C.c_uint32)(); vals = C.POINTER(C.c_uint64)(); cnt = C.c_size_t(0)
        _check(lib().zkg_circuit_sparse_witness(C.c_void_p(self._h), C.byref(tags), C.byref(idx), C.byref(vals), C.byref(cnt)), "zkg_circuit_sparse_witness")
        n, c = self.r1cs.num_variables, cnt.value
        t = np.ctypeslib.as_array(tags, shape=(n,)).copy()
        i = np.ctypeslib.as_array(idx, shape=(c,)).copy() if c else np.zeros(0, np.uint32)
        v = np.ctypeslib.as_array(vals, shape=(c, 4)).copy() if c else np.zeros((0, 4), np.uint64)
        return t, i, v

    def is_satisfied(self):
        return bool(lib().zkg_circuit_is_satisfied(self._h))

    def first_unsatisfied(self):
        return int(lib().zkg_circuit_first_unsatisfied(self._h))

    def csr(self):
        """numpy copies of the three CSR matrices: (rowptr, col, val) x 3"""
        out = []
        C_ = self.r1cs.num_constraints
        for m in "abc":
            rp = np.ctypeslib.as_array(C.cast(getattr(self.r1cs, m + "_rowptr"), C.POINTER(C.c_uint32)), shape=(C_ + 1,)).copy()
            nnz = int(rp[-1])
            col = np.ctypeslib.as_array(C.cast(getattr(self.r1cs, m + "_col"), C.POINTER(C.c_uint32)), shape=(max(nnz, 1),))[:nnz].copy()
            val = np.ctypeslib.as_array(C.cast(getattr(self.r1cs, m + "_val"), C.POINTER(C.c_uint64)), shape=(max(nnz, 1), 4))[:nnz].copy()
            out.append((rp, col, val))
        return out

    def free(self):
        if self._h:
            lib().zkg_circuit_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def zklaim_input_map(ctx):
    L = lib()
    L.zkg_zklaim_input_map.restype = C.c_size_t
    L.zkg_zklaim_input_map.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    n = L.zkg_zklaim_input_map(C.cast(C.pointer(ctx), C.c_void_p), None, 0)
    out = np.zeros((n, 4), np.uint64)
    L.zkg_zklaim_input_map(C.cast(C.pointer(ctx), C.c_void_p), _p(out), n)
    return out


# ---- key generation / verification (r1cs_gg_ppzksnark_generator, r1cs_gg_ppzksnark_verifier_strong_IC) ------------------
class Keypair:
    def __init__(self, r1cs, trapdoor=None):
        L = lib()
        L.zkg_groth16_setup.restype = C.c_void_p
        L.zkg_groth16_setup.argtypes = [C.c_void_p, C.c_void_p]
        L.zkg_keypair_pk.restype = C.POINTER(PK)
        for f in (L.zkg_keypair_pk, L.zkg_keypair_free, L.zkg_keypair_swapped):
            f.argtypes = [C.c_void_p]
        for f in (L.zkg_keypair_pk_blob, L.zkg_keypair_vk_blob):
            f.restype = C.c_size_t; f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        td = None if trapdoor is None else _u64(trapdoor)
        self._h = L.zkg_groth16_setup(C.byref(r1cs), _p(td))
        if not self._h:
            raise ZkgError("zkg_groth16_setup failed: " + L.zkg_last_error().decode())
        self.pk = L.zkg_keypair_pk(self._h).contents
        self.swapped = bool(L.zkg_keypair_swapped(self._h))

    def _blob(self, fn):
        n = fn(self._h, None, 0)
        out = np.zeros(n, np.uint8)
        assert fn(self._h, _p(out), n) == n
        return out.tobytes()

    def pk_blob(self): return self._blob(lib().zkg_keypair_pk_blob)
    def vk_blob(self): return self._blob(lib().zkg_keypair_vk_blob)

    def array(self, name, count, limbs):
        return np.ctypeslib.as_array(C.cast(getattr(self.pk, name), C.POINTER(C.c_uint64)), shape=(count, limbs)).copy()

    def free(self):
        if self._h:
            lib().zkg_keypair_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def groth16_verify(vk_blob, primary_input, proof):
    """0 valid, 1 invalid, 2 malformed (host pairing; no GPU needed)"""
    L = lib()
    L.zkg_groth16_verify.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    vk = np.frombuffer(vk_blob, np.uint8).copy(); pr = np.frombuffer(proof, np.uint8).copy()
    x = _u64(primary_input)
    return L.zkg_groth16_verify(_p(vk), vk.size, _p(x) if x.size else None, x.size // 4, _p(pr), pr.size)


def pairing_probe(a, b):
    out = np.zeros(384, np.uint8)
    lib().zkg_pairing_probe(_p(_u64(a)), _p(_u64(b)), _p(out))
    return out.tobytes()


def pairing_selfcheck(exponent):
    """0 when the Frobenius maps and the final exponentiation's last chunk agree with square-and-multiply (exponent: Python int)"""
    n = (exponent.bit_length() + 31) // 32
    e = np.array([(exponent >> (32 * i)) & 0xFFFFFFFF for i in range(n)], np.uint32)
    return lib().zkg_pairing_selfcheck(e.ctypes.data_as(C.c_void_p), n)


def libsnark_trusted_setup(ctx): return lib().libsnark_trusted_setup(C.byref(ctx))
def libsnark_prove(ctx): return lib().libsnark_prove(C.byref(ctx))
def libsnark_verify(ctx): return lib().libsnark_verify(C.byref(ctx))


def ctx_blob(ctx, which):
    size = getattr(ctx, which + "_size"); ptr = getattr(ctx, which)
    return C.string_at(ptr, size) if ptr and size else b""
