"""Timeline of the LAST proof in a rocprofv3 --kernel-trace of tools/zklaim_prove_profile.py: every kernel with its start offset, duration,
queue, and the idle time of its queue before it.  Usage: python tools/proof_timeline.py <..._kernel_trace.csv> [kernels-per-proof marker]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a proof starts with the witness upload's first kernel: k_expand_tags (sparse) / k_classify
starts = [i for i, r in enumerate(rows) if "k_expand_tags" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
busy_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")[:44]
    gap_q = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    idle = max(0, s - busy_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{q:>3s}  queue-gap {gap_q:7.1f}  chip-idle-before {idle:6.1f}  {name}  grid {r.get('Grid_Size_X', '')}")
    last_end[q] = e
    busy_end = max(busy_end, e)
print("span us", (busy_end - t0) / 1e3)
