"""Turns the rocprofv3 --pmc passes of tools/pmc_collect.sh into profiles/pmc_traffic.json (per-kernel HBM bytes per launch).

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counter values are KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide (16 B/lane) coalesced stream and is otherwise uncalibrated, so the factor for the MSM's 64-B-per-lane gather is
measured here on a known byte count (tools/pmc_calib.hip); WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections, csv, glob, json, sys

src, dst = sys.argv[1], sys.argv[2]


def load(pat):
    return list(csv.DictReader(open(glob.glob(pat)[0])))


CALLS = 6          # tools/pmc_collect.sh runs bench.py --headline-only --steps 3 --warmup 1: every launch belongs to one of these 4 steps or to bench.py's 2 setup calls


def per_kernel(rows):
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]) * 1024.0)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


calib_f = per_kernel(load(f"{src}/calib_FETCH_SIZE/*/*counter_collection.csv"))
calib_w = per_kernel(load(f"{src}/calib_WRITE_SIZE/*/*counter_collection.csv"))
n_gather = 1 << 24
known_gather = n_gather * 64 + n_gather * 4
known_stream = (1 << 24) * 64
gather_factor = known_gather / calib_f["k_gather64"][0]
stream_factor = known_stream / calib_f["k_stream16"][0]
store_factor = known_stream / calib_w["k_store16"][0]
bf = per_kernel(load(f"{src}/bench_FETCH_SIZE/*/*counter_collection.csv"))
bw = per_kernel(load(f"{src}/bench_WRITE_SIZE/*/*counter_collection.csv"))
out = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --no-cpu-baseline --no-extras --headline-only --steps 3 --warmup 1",
       "calibration": {"gather64_true_over_reported": round(gather_factor, 4), "stream16_true_over_reported": round(stream_factor, 4),
                       "store16_true_over_reported": round(store_factor, 4),
                       "note": "FETCH_SIZE of k_bucket_accum29 is corrected with the 64-B gather factor (its reads are 64-B record gathers, four 16-B loads per lane, plus a 4-B/lane index stream)"},
       "kernels": {}}
for k in sorted(set(bf) | set(bw)):
    if not k.startswith("zk::"):
        continue
    f = bf.get(k, (0, 0))[0]; w = bw.get(k, (0, 0))[0]
    out["kernels"][k] = {"fetch_bytes_raw": int(f), "write_bytes_raw": int(w), "launches": bf.get(k, (0, 0))[1]}
acc = [k for k in out["kernels"] if k.startswith("zk::k_bucket_accum29")] or [k for k in out["kernels"] if k.startswith("zk::k_bucket_accum<zk::Fp<zk::FqParams>")]
if acc:
    k = acc[0]; e = out["kernels"][k]
    e["hbm_bytes_corrected"] = int(e["fetch_bytes_raw"] * gather_factor + e["write_bytes_raw"] * store_factor)
    out["dominant_kernel"] = k
    out["dominant_hbm_bytes_per_launch"] = e["hbm_bytes_corrected"]
    out["dominant_launches_per_step"] = e["launches"] / CALLS
    out["dominant_hbm_bytes_per_step"] = int(e["hbm_bytes_corrected"] * e["launches"] / CALLS)
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_sha16
out["source_sha16"] = kernel_source_sha16()          # bench.py reports the traffic figure only while the kernel sources still hash to this
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
