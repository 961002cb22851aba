#!/usr/bin/env python3
"""bench.py — headline measurement of the zklaim Groth16 hot path on MI355X.

Workload at N GPUs (BASELINE.json configs[1], weak scaling): every rank holds a resident shard of 2^20 alt_bn128 G1
bases + 2^20 uniform scalars and computes its partial Pippenger MSM (libff multi_exp, reached from snark.cpp:126 of the
reference); the N normalised partial points are all-gathered (RCCL via torch.distributed; RCCL has no elliptic-curve reduce
op) and summed.  One "step" = one full MSM over all N*2^20 points.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `value` = algorithmic GB/s = 96 B/point * points / wall time.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# Under torch.distributed.run the process also holds RCCL's streams.  HIP maps streams onto 4 hardware queues by default; the piece-wise step
# drives four streams of its own (caller's, copy, record conversion, high-priority sort), and with RCCL's beside them two of those land on one
# queue: the upload then serialises with the compute it should hide (measured on one rank: 2.02 -> 2.67 ms per step; 8 or 16 queues restore
# 2.02, tools/r4_dist_ab.sh).  Must be in the environment before the HIP runtime starts, i.e. before torch is imported.
if "WORLD_SIZE" in os.environ:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)
SEED = 0x5A4B4C41494D0000
LOGN = 20
BYTES_PER_POINT = 96          # 64 B affine base + 32 B scalar
HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MASK = (1 << 64) - 1
R_LIMBS = np.array([0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029], dtype=np.uint64)
G1_GEN_MONT = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f,
                        0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)


def kernel_source_sha16():
    """sha256[:16] over the HIP sources the dominant kernel is built from: profiles/pmc_traffic.json records the value it was collected
    on (tools/pmc_summary.py), and the traffic figure is only reported when it still matches the sources this run was built from"""
    import hashlib
    hsh = hashlib.sha256()
    for f in ("msm.hip", "curve.hip.hpp", "fp.hip.hpp", "mont_asm.inc", "fq29.hip.hpp", "f29_asm.inc"):
        with open(os.path.join(ROOT, "zklaim_amd", "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def stats_ms(samples):
    a = np.sort(np.asarray(samples, dtype=np.float64)) * 1e3
    return {"median": round(float(np.median(a)), 4), "min": round(float(a[0]), 4), "p95": round(float(np.percentile(a, 95)), 4),
            "p99": round(float(np.percentile(a, 99)), 4), "max": round(float(a[-1]), 4), "n": int(a.size)}


def splitmix_fr(n, seed):
    """n uniform canonical scalars in [0, r): SplitMix64 stream, 4 draws -> 254 bits -> rejection."""
    out = np.zeros((0, 4), np.uint64)
    state = np.uint64(seed & MASK)
    with np.errstate(over="ignore"):
        while out.shape[0] < n:
            k = int((n - out.shape[0]) * 1.4) + 16
            idx = np.arange(1, 4 * k + 1, dtype=np.uint64)
            z = state + idx * np.uint64(0x9E3779B97F4A7C15)
            state = state + np.uint64(4 * k) * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            v = (z ^ (z >> np.uint64(31))).reshape(k, 4).copy()
            v[:, 3] &= np.uint64((1 << 62) - 1)
            lt = np.zeros(k, bool); eq = np.ones(k, bool)
            for i in (3, 2, 1, 0):
                lt |= eq & (v[:, i] < R_LIMBS[i]); eq &= v[:, i] == R_LIMBS[i]
            out = np.concatenate([out, v[lt]])
    return np.ascontiguousarray(out[:n])


G2_GEN_MONT = np.array([0x8e83b5d102bc2026, 0xdceb1935497b0172, 0xfbb8264797811adf, 0x19573841af96503b,
                        0xafb4737da84c6140, 0x6043dd5a5802d8c4, 0x09e950fc52a02f86, 0x14fef0833aea7b6b,
                        0x619dfa9d886be9f6, 0xfe7fd297f59e9b78, 0xff9e1a62231b7dfe, 0x28fd7eebae9e4206,
                        0x64095b56c71856ee, 0xdc57f922327d3cbb, 0x55f935be33351076, 0x0da4a0e693fd6482], dtype=np.uint64)


def agree_ok(torch, dist, ok, use_dist):
    """True only when EVERY rank reports ok: a rank that failed before a collective must not leave the others waiting in it"""
    if not use_dist:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def all_ranks_same(torch, dist, point, use_dist):
    """every rank holds the same combined point (12 limbs): compared across ranks, not assumed"""
    if not use_dist:
        return True
    mine = torch.from_numpy(np.ascontiguousarray(point, dtype=np.uint64).view(np.int64).reshape(-1)).cuda()
    every = torch.empty(dist.get_world_size() * mine.numel(), dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(every, mine)
    return bool((every.view(dist.get_world_size(), -1) == mine).all().item())


def msm_leg(zkg, torch, dist, logn, seed_off, steps, use_dist, world, check_cpu=False):
    """one more resident-input G1 MSM leg at 2^logn points per rank (BASELINE configs[4]: 2^23 per GPU = 2^26 over 8): per-step wall time,
    max over ranks, barrier + synchronize on both sides like the headline loop.  Self-checking: the rank's partial is compared with the sum
    of its 2^20-point pieces (each through the two-pass sort the headline uses — a different code path from the one-pass sort of a 2^23-point
    launch), with the oracle on all points when check_cpu, and the combined point is compared across ranks."""
    from zklaim_amd import dist as zdist
    n = 1 << logn
    d_bases = d_sc = None; ok = True
    try:                                                                  # everything before the first collective: a failure here is agreed on below
        ks = splitmix_fr(n, SEED + 0x600 + seed_off); sc = splitmix_fr(n, SEED + 0x700 + seed_off)
        d_k = torch.from_numpy(ks.view(np.int64)).cuda()
        d_bases = torch.empty((n, 8), dtype=torch.int64, device="cuda")
        zkg.fixed_base_g1_dev(G1_GEN_MONT, d_k.data_ptr(), n, d_bases.data_ptr())
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        del d_k
        stream = torch.cuda.current_stream().cuda_stream
        part0 = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n, stream=stream)
        piece = min(n, 1 << 20)
        pieces = np.stack([zkg.msm_g1_dev(d_bases[lo:].data_ptr(), d_sc[lo:].data_ptr(), piece, stream=stream) for lo in range(0, n, piece)])
        pieces_ok = bool(np.array_equal(zkg.g1_sum(pieces), part0))
    except Exception as exc:
        print(f"msm leg 2^{logn}: setup failed on this rank:", exc, file=sys.stderr); ok = False
    if not agree_ok(torch, dist, ok, use_dist):
        return None

    def step():
        part = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n, stream=stream)
        return zdist.combine_partials_g1(part, device="cuda") if use_dist else part

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    step(); result = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per = dt / steps
    out = {"points_per_gpu": n, "total_points": n * world, "n_gpus": world, "steps": steps, "ms_per_step": round(per * 1e3, 3),
           "GBps_algorithmic": round(BYTES_PER_POINT * n * world / per / 1e9, 3), "partial_equals_sum_of_2p20_pieces": pieces_ok,
           "all_ranks_same_result": all_ranks_same(torch, dist, result, use_dist)}
    if check_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import zkoracle
        threads = zkoracle.num_threads()
        t1 = time.perf_counter()
        ref = zkoracle.msm_g1(d_bases.cpu().numpy().view(np.uint64), sc, zkoracle.BDLO12, threads)
        out["cpu_check"] = {"gpu_matches_cpu": bool(np.array_equal(ref, part0)), "seconds": round(time.perf_counter() - t1, 2), "cores": threads, "kind": "port",
                            "sample": f"all 2^{logn} points of this leg, oracle BDLO12 bucket method, chunked over {threads} threads"}
    return out


def extras(zkg, torch, args, with_cpu):
    """Secondary legs of the metric, single GPU: NTT 2^20 (BASELINE configs[2]) and a full Groth16 prove on a zklaim-shaped
    system padded to m = 2^logm (configs[3]); not part of `value`."""
    out = {}
    # ---- NTT 2^20 forward + inverse, resident
    logn = 20; n = 1 << logn
    a = splitmix_fr(n, SEED + 3)
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    for inv in (False, True):
        zkg.ntt_dev(d_a.data_ptr(), logn, inverse=inv)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        zkg.ntt_dev(d_a.data_ptr(), logn, inverse=False); zkg.ntt_dev(d_a.data_ptr(), logn, inverse=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * reps)
    roundtrip_ok = bool(np.array_equal(d_a.cpu().numpy().view(np.uint64), a))
    # device time of one transform, HIP events on the launch stream (back-to-back launches, no host gaps)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        zkg.ntt_dev(d_a.data_ptr(), logn, inverse=False, stream=torch.cuda.current_stream().cuda_stream)
    ev[1].record(); torch.cuda.synchronize()
    dev_ms = ev[0].elapsed_time(ev[1]) / reps
    out["ntt_2p20"] = {"ms_per_transform": round(dt * 1e3, 4), "device_ms_per_transform": round(dev_ms, 4), "GBps_algorithmic": round(64 * n / dt / 1e9, 2),
                       "GBps_algorithmic_device": round(64 * n / (dev_ms * 1e-3) / 1e9, 2), "bytes_per_element": 64,
                       "frac_of_hbm_peak": round(64 * n / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "forward_inverse_roundtrip_exact": roundtrip_ok}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import zkoracle
        t1 = time.perf_counter()
        ref = zkoracle.fft(a, inverse=False)
        cpu_dt = time.perf_counter() - t1
        d_a.copy_(torch.from_numpy(a.view(np.int64))); zkg.ntt_dev(d_a.data_ptr(), logn, inverse=False); torch.cuda.synchronize()
        out["ntt_2p20"]["cpu_baseline"] = {"seconds": round(cpu_dt, 3), "cores": 1, "kind": "port", "sample": "one forward 2^20 transform, oracle serial radix-2 FFT",
                                          "gpu_matches_cpu": bool(np.array_equal(d_a.cpu().numpy().view(np.uint64), ref))}

    # a SECOND figure for the headline workload, not `value`: the same 2^20 points as fixed bases kept resident with their per-window tables
    # (zkg_msm_g1_bases_upload: what the prover builds for a key's H query) — compared with the plain path's point
    out["msm_resident_tables"] = resident_tables_leg(zkg, torch)
    # BASELINE configs[4]'s per-GPU share (2^23 of the 2^26 points), and on request the whole 2^26 job on this one GPU (the strong-scaling reference)
    out["msm_config5_share_2p23"] = msm_leg(zkg, torch, None, 23, 0, 5, False, 1, check_cpu=with_cpu)
    if args.config5_reference:
        out["msm_config5_single_gpu_2p26"] = msm_leg(zkg, torch, None, 26, 0, 3, False, 1)
    out["groth16_prove"] = prove_leg(zkg, torch, args, with_cpu, args.prove_logm)
    if args.prove_logm != 20 and not args.no_northstar:
        out["groth16_prove_2p20"] = prove_leg(zkg, torch, args, with_cpu, 20)                # the north star's 2^20-constraint case
    return out


def resident_tables_leg(zkg, torch, logn=20, steps=20):
    n = 1 << logn
    ks = splitmix_fr(n, SEED + 1); sc = splitmix_fr(n, SEED + 2)                   # the headline workload's bases and scalars
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_bases = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zkg.fixed_base_g1_dev(G1_GEN_MONT, d_k.data_ptr(), n, d_bases.data_ptr())
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = zkg.ResidentBases(d_bases.data_ptr(), n)
    t_build = time.perf_counter() - t0
    plain = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n)
    for _ in range(3):
        got = h.msm(d_sc.data_ptr())
    each = []
    for _ in range(steps):
        t0 = time.perf_counter(); got = h.msm(d_sc.data_ptr()); each.append(time.perf_counter() - t0)
    h.free()
    dt = sum(each) / steps
    return {"workload": "2^20-point alt_bn128 G1 MSM, fixed bases resident with precomputed window tables (16 levels: 16 x the bases' memory, twice that with the 29-bit records)",
            "ms_per_step": round(dt * 1e3, 4), "ms_per_step_stats": stats_ms(each), "GBps_algorithmic": round(BYTES_PER_POINT * n / dt / 1e9, 3),
            "table_build_ms": round(t_build * 1e3, 1), "same_point_as_plain_path": bool(np.array_equal(got, plain)),
            "note": "a second figure: the headline `value` is the plain path (bases as given, nothing precomputed)"}


def prove_leg(zkg, torch, args, with_cpu, logm):
    # ---- Groth16 prove on zklaim's own credential circuit (zklaim_gadget rebuilt on the host, zklaim_amd/csrc/zklaim_circuit.hip):
    #      k payloads -> m = 2^logm (k = 8 -> 2^18, BASELINE configs[3]; k = 20 -> 2^20, the north-star size).  Keys come from the
    #      product's GPU generator with a fixed trapdoor; (r, s) fixed; the CPU oracle proves the same instance for byte parity.
    k_payloads = {15: 1, 16: 2, 17: 4, 18: 8, 19: 16, 20: 37}.get(logm, 8)      # 37 payloads: C + l + 1 = 1 023 318 -> m = 2^20
    t_syn = time.perf_counter()
    keep = []
    pls = [dict(attrs=[1990 + i, 7 * i, 42, i, 5], refs=[2100, 7 * i, 41, 0, 5], ops=["less", "eq", "greater", "noop", "greater_or_eq"], salt=0x5A4B + i)
           for i in range(k_payloads)]
    ctx = zkg.make_ctx(pls, keep)
    ck = zkg.ZklaimCircuit(ctx)
    assert ck.is_satisfied()
    nv, l, ncons = ck.r1cs.num_variables, ck.r1cs.num_inputs, ck.r1cs.num_constraints
    w = ck.witness()
    kp = zkg.Keypair(ck.r1cs, splitmix_fr(5, SEED + 4))
    m = kp.pk.domain_size or (1 << kp.pk.log_m)
    logm = kp.pk.log_m
    t_setup = time.perf_counter() - t_syn
    crs = zkg.Crs(kp.pk)
    rs = splitmix_fr(2, SEED + 5)
    for _ in range(4):                      # the first calls grow the runtime's per-stream pools (9 streams in flight); steady state after ~3
        rc, proof = crs.prove(w, rs[0], rs[1])
    assert rc == 0, "credential must satisfy the circuit"
    verified = zkg.groth16_verify(kp.vk_blob(), w[:l], proof) == 0
    reps = 30                                                     # the reference's benchmark protocol: RUNS = 30 (src/main_benchmark.c:175-176)
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc, proof2 = crs.prove(w, rs[0], rs[1])
        each.append(time.perf_counter() - t0)
    dt = sum(each) / reps
    tags, fidx, fvals = ck.sparse_witness()                      # the seam's form of the same witness: tags + the ~3 % non-bit values
    rc_s, proof_s = crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
    each_sparse = []
    for _ in range(reps):
        t0 = time.perf_counter()
        crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
        each_sparse.append(time.perf_counter() - t0)
    dt_sparse = sum(each_sparse) / reps
    # latency tail: many proofs back to back on the resident key (the seam's sparse form), median / p95 / p99 / max
    tail_n = 1000 if m <= (1 << 18) else 300
    tail = []
    for _ in range(tail_n):
        t0 = time.perf_counter()
        crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
        tail.append(time.perf_counter() - t0)
    # throughput with several callers of the ONE resident key (its three prover slots; the C ABI releases nothing but the GIL-free call itself):
    # three host threads prove back to back, every proof compared with the single caller's bytes
    import threading
    n_callers, per_caller = 3, (60 if m <= (1 << 18) else 20)
    wrong = []

    def caller():
        for _ in range(per_caller):
            rc_c, proof_c = crs.prove_sparse(tags, fidx, fvals, rs[0], rs[1])
            if rc_c != 0 or proof_c != proof_s:
                wrong.append(rc_c)
    threads = [threading.Thread(target=caller) for _ in range(n_callers)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt_callers = time.perf_counter() - t0
    A, B, C = ck.csr()
    nnz = int(len(A[1]) + len(B[1]) + len(C[1]))
    alg_bytes = 7 * 64 * m + 96 * (nv + 1) + (128 + 64 + 32) * (nv + 1) + 96 * (m - 1) + 96 * (nv - l)
    g = {"circuit": f"zklaim_gadget, {k_payloads} payloads (SHA-256 + 5 comparisons each)", "log_m": logm, "domain_size": int(m), "num_variables": int(nv), "num_inputs": int(l),
         "num_constraints": int(ncons), "nnz": nnz, "ms_per_proof": round(dt * 1e3, 3), "proofs_per_sec": round(1.0 / dt, 3),
         "ms_per_proof_stats": stats_ms(each), "ms_per_proof_sparse_witness": round(dt_sparse * 1e3, 3), "ms_per_proof_sparse_witness_stats": stats_ms(each_sparse),
         "latency_tail_sparse_witness": stats_ms(tail),
         "three_callers_one_key": {"proofs_per_sec": round(n_callers * per_caller / dt_callers, 1), "callers": n_callers, "proofs": n_callers * per_caller,
                                   "all_proofs_equal_single_caller": not wrong, "what": "sparse-witness proofs from three host threads on one resident key (its prover slots overlap on the GPU)"},
         "timing_note": "wall clock around the C-ABI call: witness H2D (dense: 32 B per variable; sparse: tags + listed values), all device work, proof D2H, host assembly", "sparse_witness_same_bytes": bool(rc_s == 0 and proof_s == proof),
         "algorithmic_bytes_per_proof": int(alg_bytes), "GBps_algorithmic": round(alg_bytes / dt / 1e9, 2), "stage_ms": [round(x, 3) for x in crs.stage_ms()],
         "stage_names": ["r1cs_matvec", "7_ntt+pointwise", "witness_A_Bg1_L_bucket_method", "-", "witness_Bg2_bucket_method", "msm_H", "-", "wall_total_incl_host_assembly"],
         "stage_note": "the stages run on separate HIP streams and overlap; the flat sums over the witness bits equal to one run on a fourth stream",
         "trusted_setup_seconds_gpu": round(t_setup, 2), "deterministic": proof2 == proof, "proof_verifies": verified}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import zkoracle
        okeep = []
        if kp.swapped:
            A, B = B, A
        ocs = zkoracle.make_r1cs(nv, l, A, B, C, okeep)
        arrays = {name: kp.array(name, cnt, lim) for name, cnt, lim in (("A_query", nv + 1, 8), ("B_g1", nv + 1, 8), ("B_g2", nv + 1, 16), ("H_query", m - 1, 8),
                  ("L_query", nv - l, 8), ("alpha_g1", 1, 8), ("beta_g1", 1, 8), ("delta_g1", 1, 8), ("beta_g2", 1, 16), ("delta_g2", 1, 16))}
        arrays["m"] = m
        opk = zkoracle.make_pk(ocs, arrays)
        t1 = time.perf_counter()
        rc_o, proof_o = zkoracle.groth16_prove(opk, w, rs[0], rs[1], True, 1)
        cpu_dt = time.perf_counter() - t1
        g["cpu_baseline"] = {"proofs_per_sec": round(1.0 / cpu_dt, 4), "seconds": round(cpu_dt, 2), "cores": 1, "kind": "port",
                             "sample": "one full prove of the same credential, oracle restatement of r1cs_gg_ppzksnark_prover, single thread"}
        g["proof_bytes_match_cpu"] = bool(rc_o == 0 and proof_o == proof)
        g["speedup_vs_cpu_1core"] = round(cpu_dt / dt, 1)
        g["speedup_note"] = "against the single-thread PORT (oracle restatement), not libsnark's own x86-64 assembly build; a ratio says nothing about kernel quality"
    crs.free(); kp.free(); ck.free()
    # the SAME credential on the REFERENCE's relation: zklaim_gadget.cpp:583-699 assigns the pack_PL / pack_REF / pack_OPS packings but never
    # generates their constraints, so its R1CS is 78 rows per payload smaller than the hardened shape proved above
    # (ZKG_CIRCUIT_REFERENCE_QUIRK rebuilds it).  Same payloads, own key, 30 proofs: the "zklaim gadget" half of the metric named on the
    # reference's shape as well.
    try:
        ckq = zkg.ZklaimCircuit(ctx, reference_quirk=True)
        wq = ckq.witness()
        kpq = zkg.Keypair(ckq.r1cs, splitmix_fr(5, SEED + 4))
        crsq = zkg.Crs(kpq.pk)
        for _ in range(4):
            rcq, proofq = crsq.prove(wq, rs[0], rs[1])
        eachq = []
        for _ in range(reps):
            t0 = time.perf_counter(); crsq.prove(wq, rs[0], rs[1]); eachq.append(time.perf_counter() - t0)
        dtq = sum(eachq) / reps
        g["reference_shape_circuit"] = {"what": "ZKG_CIRCUIT_REFERENCE_QUIRK: the packings left unconstrained as zklaim_gadget.cpp:583-699 leaves them",
                                        "num_constraints": int(ckq.r1cs.num_constraints), "num_variables": int(ckq.r1cs.num_variables), "domain_size": int(kpq.pk.domain_size or (1 << kpq.pk.log_m)),
                                        "ms_per_proof": round(dtq * 1e3, 3), "proofs_per_sec": round(1.0 / dtq, 3), "ms_per_proof_stats": stats_ms(eachq),
                                        "proof_verifies": bool(rcq == 0 and zkg.groth16_verify(kpq.vk_blob(), wq[:ckq.r1cs.num_inputs], proofq) == 0)}
        crsq.free(); kpq.free(); ckq.free()
    except Exception as exc:                                     # never lose the line over this extra
        g["reference_shape_circuit"] = {"error": repr(exc)}
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=LOGN, help="log2 points per GPU (default: BASELINE config 2)")
    ap.add_argument("--sharding", choices=["points", "windows"], default="points",
                    help="multi-GPU partition: points (default; each rank owns 2^logn points) or windows (every rank holds all N*2^logn points "
                         "and owns every N-th Pippenger window; SURVEY.md section 8e's variant)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the NTT and Groth16-prove legs (reported under 'extras')")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline steps (no resident-scalars legs): what tools/pmc_collect.sh profiles, so that every launch it sees belongs to a headline step")
    ap.add_argument("--config5-reference", action="store_true", help="also time the whole 2^26-point MSM of BASELINE configs[4] on ONE GPU (6 GB of inputs; the strong-scaling reference for the 8 x 2^23 run)")
    ap.add_argument("--no-northstar", action="store_true", help="skip the second prove leg (37 payloads, m = 2^20: the north star's 2^20-constraint case)")
    ap.add_argument("--prove-logm", type=int, default=18, help="log2 of the evaluation domain of the prove leg: 18 = 8 payloads (BASELINE configs[3]), 20 = 37 payloads (the north star's 2^20-constraint case)")
    args = ap.parse_args()
    gc.disable()        # the interpreter's cyclic collector pauses for tens of ms every few hundred calls (tools/prove_outliers.py): not the product's latency

    # stdout carries exactly ONE JSON line: libraries that print banners (RCCL prints its version on first use) go to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import zklaim_amd as zkg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    use_dist = "WORLD_SIZE" in os.environ and "RANK" in os.environ           # launched by torch.distributed.run (any N, also N=1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    zkg.init(local_rank)
    arch, cus = zkg.device_info()

    n = 1 << args.logn
    by_windows = args.sharding == "windows" and world > 1
    if by_windows:
        n *= world                                  # every rank holds the whole job's points (same seeds everywhere) and owns W/world windows
    # ---- synthetic resident inputs: bases k_i*G1 built on device by the product's fixed-base kernel
    ks = splitmix_fr(n, SEED + 1 + (0 if by_windows else 0x100 * rank))
    sc = splitmix_fr(n, SEED + 2 + (0 if by_windows else 0x100 * rank))
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_bases = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zkg.fixed_base_g1_dev(G1_GEN_MONT, d_k.data_ptr(), n, d_bases.data_ptr())
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    from zklaim_amd import dist as zdist

    # The timed step is SURVEY.md section 8(d)'s / BASELINE.md's: wall clock around the C-ABI call INCLUDING the upload of the scalars (32 B x
    # points from pinned host memory), bases resident, result back on the host.  zkg_msm_g1_host_scalars cuts the job into pieces whose
    # uploads run under the work of the pieces before them.  The resident-scalars figure (zkg_msm_g1_dev) is reported beside it.
    h_sc = torch.from_numpy(sc.view(np.int64)).pin_memory()

    def step_resident():
        if by_windows:
            w0, ws = zdist.window_shard(world, rank)
            part = zkg.msm_g1_windows_dev(d_bases.data_ptr(), d_sc.data_ptr(), n, w0, ws, stream=stream)
        else:
            part = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n, stream=stream)   # normalised partial (host)
        return zdist.combine_partials_g1(part, device="cuda") if use_dist else part

    def step():
        if by_windows:
            d_sc.copy_(h_sc, non_blocking=True)              # the window-sharded variant has no piece-wise form: upload, then the resident call
            return step_resident()
        part = zkg.msm_g1_host_scalars(d_bases.data_ptr(), h_sc.data_ptr(), n, stream=stream)
        return zdist.combine_partials_g1(part, device="cuda") if use_dist else part

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, like the upload of the bases: the library's work buffers are allocated by its first two calls (24 ms and 6 ms, then 1.9 - 2.0 ms per
    # call: tools/first_steps.py), so they are made here whatever --warmup says; the W warm-up steps and the K timed steps follow as asked
    for _ in range(2):
        step()
    for _ in range(args.warmup):
        result = step()
    fence()
    zkg.timing_reset()
    per_step = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_s = time.perf_counter()
        result = step()                                   # returns after the result point is on the host (the call synchronises its stream)
        per_step.append(time.perf_counter() - t_s)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms, launches = zkg.timing_dominant_ms()
    kern_ms = kern_ms * launches / max(1, args.steps)                        # the accumulation's launches of one step (one per piece) added up
    same_everywhere = all_ranks_same(torch, dist, result, use_dist)          # outside the timed region
    # the same step with the scalars already in HBM (rounds 1-3's headline), and with ONE upload followed by the resident call (what the
    # piece-wise entry point replaces) — reported beside `value`, same result required
    resident = None
    if world == 1 and not args.headline_only:
        for _ in range(4):                                   # (the job's buffers grow from the pieces' sizes to the whole vector's on the first calls)
            step_resident()
        zkg.timing_reset()
        each = []
        for _ in range(args.steps):
            t_s = time.perf_counter(); r2 = step_resident(); each.append(time.perf_counter() - t_s)
        res_kern_ms, res_launches = zkg.timing_dominant_ms()
        d_sc2 = torch.empty_like(d_sc)
        def step_plain_upload():
            d_sc2.copy_(h_sc, non_blocking=True)
            return zkg.msm_g1_dev(d_bases.data_ptr(), d_sc2.data_ptr(), n, stream=stream)
        step_plain_upload()
        each_u = []
        for _ in range(args.steps):
            t_s = time.perf_counter(); r3 = step_plain_upload(); each_u.append(time.perf_counter() - t_s)
        resident = {"ms_per_step": stats_ms(each), "GBps_algorithmic_mean": round(BYTES_PER_POINT * n / float(np.mean(each)) / 1e9, 3),
                    "GBps_algorithmic_median": round(BYTES_PER_POINT * n / float(np.median(each)) / 1e9, 3),
                    "same_result": bool(np.array_equal(r2, result)), "what": "zkg_msm_g1_dev: bases AND scalars resident in HBM when the timed call starts (the headline of rounds 1-3)",
                    "accumulation_kernel": {"kernel_ms": round(res_kern_ms, 4), "launches": res_launches, "achieved_GBps": round(BYTES_PER_POINT * n / (res_kern_ms * 1e-3) / 1e9, 3) if res_kern_ms > 0 else None,
                                            "frac_of_hbm_peak": round(BYTES_PER_POINT * n / (res_kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6) if res_kern_ms > 0 else None,
                                            "note": "k_bucket_accum29 as ONE launch with the chip to itself; in the headline step its launches (one per piece) share the chip with the next piece's digit sort (roofline.kernel_ms)"},
                    "one_upload_then_resident_call": {"ms_per_step": stats_ms(each_u), "GBps_algorithmic_median": round(BYTES_PER_POINT * n / float(np.median(each_u)) / 1e9, 3),
                                                      "same_result": bool(np.array_equal(r3, result))}}

    total_points = n if by_windows else n * world
    n = n // world if by_windows else n            # per-rank share of the points, for the per-launch algorithmic bytes below
    ms_per_step = dt / args.steps * 1e3
    value = BYTES_PER_POINT * total_points / (dt / args.steps) / 1e9
    achieved = BYTES_PER_POINT * n / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None

    line = {
        "metric": "Groth16 proofs/sec (zklaim gadget, alt_bn128) + G1 MSM GB/s vs HBM roofline", "metric_component": "G1 MSM GB/s (value, unit); Groth16 proofs/sec of the zklaim gadget in extras.groth16_prove and proofs_per_sec", "value": round(value, 3), "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_step_stats": stats_ms(per_step),
        "timed_region": "SURVEY.md section 8(d): wall clock around the C-ABI call (zkg_msm_g1_host_scalars) incl. the upload of the scalars from pinned host memory and the result's return; bases resident",
        "scalars_resident": resident, "all_ranks_same_result": same_everywhere, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs, Montgomery (254-bit Fq/Fr: 9 x 29-bit in the accumulation, reduction and NTT kernels, 8 x 32-bit elsewhere)", "data": "synthetic",
        "config": {"workload": f"2^{args.logn}-point alt_bn128 G1 Pippenger MSM per GPU, random scalars/bases (BASELINE configs[1])",
                   "points_per_gpu": n, "total_points": total_points, "arch": arch, "compute_units": cus,
                   "sharding": ("windows sharded per rank (every rank holds all points); " if by_windows else "points sharded per rank; ") + "all-gather of normalised partial points + EC add" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "k_bucket_accum29", "achieved": None if achieved is None else round(achieved, 3), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                     "kernel_ms": round(kern_ms, 4), "launches": launches, "launches_per_step": round(launches / max(1, args.steps), 2),
                     "kernel_ms_note": "the accumulation's launches of ONE step added up (the piece-wise step launches it once per piece), HIP events on the launch stream around every launch of every fourth step of the timed region (their records cost the stream 2 % of a step when every launch carries them; ZKG_KERNEL_TIMER_STRIDE=1)",
                     "note": "integer-VALU-bound kernel (10 Montgomery products of 9 x 29-bit limbs per 96 input bytes and window); see DESIGN.md"},
    }
    if kern_ms > 0:
        # second, honest roofline for this kernel: vector-ALU issue cycles.  One XYZZ mixed addition on the 29-bit representation issues
        # 2 224 vector instructions per wavefront (1 549 v_mad_u64_u32), 8 661 cycles at the measured 4.2 cycles per VOP3-encoded and 2.3 per
        # VOP2-encoded wave-instruction (profiles/r3_mul_variants.txt; counted from the kernel's ISA, tools/r3_pmc_accum.sh gives the same
        # instruction count from SQ_INSTS_VALU).  N*W additions per launch, 64 lanes per wave-instruction, 4 SIMDs per CU.
        windows = (255 + 15) // 16 if args.logn >= 20 else None
        if windows:
            cycles = n * windows / 64.0 * 8661.0
            peak_cycles_per_s = cus * 4 * 2.4e9
            line["valu_roofline"] = {"kernel": "k_bucket_accum29", "mixed_additions_per_launch": n * windows, "issue_cycles_per_wave_addition": 8661,
                                     "achieved_Gadd_per_s": round(n * windows / (kern_ms * 1e-3) / 1e9, 3),
                                     "frac_of_issue_peak": round(cycles / (kern_ms * 1e-3) / peak_cycles_per_s, 4),
                                     "note": "fraction of the SIMDs' issue cycles at the nominal 2.4 GHz; the chip holds ~2.2 GHz under this kernel (profiles/r3_effective_clock_grbm.txt)"}

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; they come from the committed
    # rocprofv3 --pmc passes over this same command (tools/pmc_collect.sh -> profiles/pmc_traffic.json), valid for N = 2^20.
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if args.logn == LOGN and os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            line["roofline"]["algorithmic_bytes_per_step"] = BYTES_PER_POINT * n
            if pmc.get("source_sha16") == kernel_source_sha16():
                line["roofline"]["traffic"] = pmc.get("dominant_hbm_bytes_per_step")
                line["roofline"]["traffic_unit"] = "bytes per step, all of the accumulation's launches of a step (FETCH_SIZE x gather calibration + WRITE_SIZE, separate --pmc passes)"
                line["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (collected on these kernel sources: sha16 " + pmc["source_sha16"] + ")"
            else:
                line["roofline"]["traffic_note"] = "profiles/pmc_traffic.json was collected on other kernel sources; not reported (re-run tools/pmc_collect.sh)"
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle's restatement of libff multi_exp<BDLO12> (kind "port": reference libsnark is an absent submodule),
        # timed on this host on the SAME 2^logn-point workload; also used to check the GPU result.
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import zkoracle
        bases_h = d_bases.cpu().numpy().view(np.uint64)
        sample_n = min(n, 1 << 20)
        t1 = time.perf_counter()
        ref = zkoracle.msm_g1(bases_h[:sample_n], sc[:sample_n], zkoracle.BDLO12, 1)
        cpu_dt = time.perf_counter() - t1
        threads = zkoracle.num_threads()
        t2 = time.perf_counter()
        ref_mt = zkoracle.msm_g1(bases_h[:sample_n], sc[:sample_n], zkoracle.BDLO12, threads)
        cpu_dt_mt = time.perf_counter() - t2
        parity = bool(np.array_equal(ref, result)) if sample_n == n else None
        line["cpu_baseline"] = {"value": round(BYTES_PER_POINT * sample_n / cpu_dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port",
                                "sample": f"one full 2^{sample_n.bit_length() - 1}-point MSM (the bench workload), oracle BDLO12 bucket method, single thread "
                                          f"({cpu_dt:.2f} s) — mirrors the reference default MULTICORE=OFF",
                                "value_all_cores": round(BYTES_PER_POINT * sample_n / cpu_dt_mt / 1e9, 5), "cores_all": threads,
                                "seconds_all_cores": round(cpu_dt_mt, 3), "gpu_matches_cpu": parity and bool(np.array_equal(ref_mt, ref))}
    if rank == 0 and world == 1 and not args.no_extras:
        line["extras"] = extras(zkg, torch, args, not args.no_cpu_baseline)
        line["proofs_per_sec"] = line["extras"]["groth16_prove"]["proofs_per_sec"]          # the other half of BASELINE.json's metric
        line["proofs_per_sec_three_callers"] = line["extras"]["groth16_prove"]["three_callers_one_key"]["proofs_per_sec"]   # same key, three host threads
    if (world > 1 or (use_dist and os.environ.get("ZKG_BENCH_TEST_REPLICAS"))) and not args.no_extras:      # the env switch lets one GPU rehearse this branch
        # the prover does not shard (replicas only): every rank proves the same 8-payload credential on its own GPU
        # with its own resident key; the job's proofs/sec is the sum over ranks
        try:
            g = prove_leg(zkg, torch, args, False, args.prove_logm)
            pps = torch.tensor([g["proofs_per_sec"], 1.0], dtype=torch.float64, device="cuda")
        except Exception as exc:                                     # never lose the headline line over the extra
            print("prove replicas failed:", exc, file=sys.stderr)
            pps = torch.tensor([0.0, 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(pps, op=dist.ReduceOp.SUM)
        if rank == 0 and int(pps[1].item()) == world:
            line["proofs_per_sec"] = round(float(pps[0].item()), 3)
            line["proofs_per_sec_note"] = f"{world} independent prover replicas (one per GPU), zklaim gadget, 8 payloads, m = 2^18"
    if (world > 1 or (use_dist and os.environ.get("ZKG_BENCH_TEST_REPLICAS"))) and not args.no_extras:      # (the env switch lets one GPU rehearse this branch)
        # BASELINE configs[4]: 2^23 points per GPU (2^26 at 8 GPUs), same exchange; every rank takes part, rank 0 reports
        # (msm_leg agrees on every rank's setup before its first collective and returns None on all ranks if one of them failed)
        c5 = msm_leg(zkg, torch, dist, 23, 0x100 * rank, 3, use_dist, world)
        # the north star's scaling claim is a STRONG-scaling one (the 2^26-point job on 8 GPUs against the same job on one): rank 0 runs the
        # whole job's points (world x 2^23) alone while the others wait, and the ratio is reported beside the leg
        ref = None
        if rank == 0 and c5:
            try:
                ref = msm_leg(zkg, torch, None, 23 + max(0, (world - 1).bit_length()), 0x900, 3, False, 1) if world > 1 else {"ms_per_step": c5["ms_per_step"], "total_points": c5["total_points"]}
            except Exception as exc:
                print("config-5 single-GPU reference failed:", exc, file=sys.stderr)
        if use_dist:
            dist.barrier()
        if rank == 0 and c5:
            if ref and ref.get("total_points") == c5["total_points"]:
                c5["strong_scaling"] = {"single_gpu_ms_for_the_same_total_points": ref["ms_per_step"], "n_gpus": world,
                                        "speedup_vs_one_gpu": round(ref["ms_per_step"] / c5["ms_per_step"], 3),
                                        "note": "the whole job's points on ONE GPU (rank 0, the others idle) against the same points sharded over all ranks, exchange and EC sum included"}
            line["msm_config5"] = c5
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    zkg.shutdown()
    # a run whose ranks disagree on the combined point (or whose upload-inclusive step differs from the resident one) is not a measurement
    bad = not same_everywhere or (resident is not None and not (resident["same_result"] and resident["one_upload_then_resident_call"]["same_result"]))
    c5_line = line.get("msm_config5") if rank == 0 else None
    if c5_line and not (c5_line.get("all_ranks_same_result") and c5_line.get("partial_equals_sum_of_2p20_pieces")):
        bad = True
    if bad:
        print("bench: results differ (all_ranks_same_result / same_result false): the line above is NOT a valid measurement", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
