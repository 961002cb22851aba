"""zklaim_amd — MI355X-native Groth16 prover hot path behind zklaim's C front-end.

The product is ``libzkg.so`` (hand-written HIP for gfx950 + the C ABI of ``include/zkg.h``).
This package is the thin ctypes mirror used by the tests and ``bench.py``; the arithmetic lives
in ``csrc/``.  There is no CPU fallback: importing works anywhere, but every compute entry point
raises unless the HIP library is built and a GPU is visible.
"""
from .api import (ZkgError, device_info, evaluation_domain_size, fixed_base_g1_dev, fixed_base_g2_dev, g1_sum, g2_sum, init, lib, msm_g1, msm_g1_dev, msm_g1_windows_dev,  # noqa: F401
                  msm_g2, msm_g2_dev, ntt, ntt_dev, shutdown, timing_dominant_ms, timing_reset, Crs, PK, R1CS, make_r1cs, make_pk,
                  DECLARED_SYMBOLS, ZklaimCircuit, ZklaimCtx, make_ctx, zklaim_input_map, OPS, Keypair, groth16_verify, pairing_probe, pairing_selfcheck,
                  libsnark_trusted_setup, libsnark_prove, libsnark_verify, ctx_blob, COMPAT_SYMBOLS, field_op, g1_add_quad29, g1_add_pair29, OK, ERROR, UNSATISFIED, init_multi, MsmShards, pk_blob_inspect, ResidentBases, msm_g1_host_scalars)
