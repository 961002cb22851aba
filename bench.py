#!/usr/bin/env python3
"""bench.py — headline measurement of the zklaim Groth16 hot path on MI355X.

Workload at N GPUs (BASELINE.json configs[1], weak scaling): every rank holds a resident shard of 2^20 alt_bn128 G1
bases + 2^20 uniform scalars and computes its partial Pippenger MSM (libff multi_exp, reached from snark.cpp:126 of the
reference); the N normalised partial points are all-gathered (RCCL via torch.distributed; RCCL has no elliptic-curve reduce
op) and summed.  One "step" = one full MSM over all N*2^20 points.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `value` = algorithmic GB/s = 96 B/point * points / wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SEED = 0x5A4B4C41494D0000
LOGN = 20
BYTES_PER_POINT = 96          # 64 B affine base + 32 B scalar
HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MASK = (1 << 64) - 1
R_LIMBS = np.array([0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029], dtype=np.uint64)
G1_GEN_MONT = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f,
                        0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=LOGN, help="log2 points per GPU (default: BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    zkg.init(local_rank)
    arch, cus = zkg.device_info()

    n = 1 << args.logn
    # ---- synthetic resident inputs: bases k_i*G1 built on device by the product's fixed-base kernel
    ks = splitmix_fr(n, SEED + 1 + 0x100 * rank)
    sc = splitmix_fr(n, SEED + 2 + 0x100 * rank)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_bases = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    zkg.fixed_base_g1_dev(G1_GEN_MONT, d_k.data_ptr(), n, d_bases.data_ptr())
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    from zklaim_amd import dist as zdist

    def step():
        part = zkg.msm_g1_dev(d_bases.data_ptr(), d_sc.data_ptr(), n, stream=stream)       # normalised partial (host)
        return zdist.combine_partials_g1(part, device="cuda") if world > 1 else part

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        result = step()
    fence()
    zkg.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms, launches = zkg.timing_dominant_ms()

    total_points = n * world
    ms_per_step = dt / args.steps * 1e3
    value = BYTES_PER_POINT * total_points / (dt / args.steps) / 1e9
    achieved = BYTES_PER_POINT * n / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None

    line = {
        "metric": "G1 MSM GB/s vs HBM roofline (alt_bn128 Pippenger; Groth16 prover hot path)", "value": round(value, 3), "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32x8-montgomery (254-bit Fq/Fr)", "data": "synthetic",
        "config": {"workload": f"2^{args.logn}-point alt_bn128 G1 Pippenger MSM per GPU, random scalars/bases (BASELINE configs[1])",
                   "points_per_gpu": n, "total_points": total_points, "arch": arch, "compute_units": cus,
                   "sharding": "points sharded per rank; all-gather of normalised partial points + EC add" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "k_bucket_accum<Fq>", "achieved": None if achieved is None else round(achieved, 3), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                     "kernel_ms": round(kern_ms, 4), "launches": launches,
                     "note": "integer-VALU-bound kernel (about 10 Montgomery multiplications per 96 input bytes); see DESIGN.md"},
    }

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
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    zkg.shutdown()


if __name__ == "__main__":
    main()
