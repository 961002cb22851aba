"""From a rocprofv3 --kernel-trace of tools/prove_throughput.py <k> 2 <n>: for a window in the two-caller phase, how the queues' kernels
overlap — per queue the busy fraction, and for every pair of queues the time both were running a kernel.
Usage: python tools/two_callers_trace.py <..._kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
lo, hi = t_end - 12_000_000, t_end - 2_000_000            # 10 ms near the end (the two-caller loop runs last)
win = [r for r in rows if lo <= int(r["Start_Timestamp"]) <= hi]
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
span = hi - lo
for q, iv in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in iv)
    print(f"queue {q}: {len(iv)} kernels, busy {100 * busy / span:.1f} % of the window")
def overlap(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
qs = sorted(byq)
for i in range(len(qs)):
    for j in range(i + 1, len(qs)):
        print(f"queues {qs[i]} & {qs[j]}: both busy {100 * overlap(byq[qs[i]], byq[qs[j]]) / span:.1f} %")
# longest gaps of the busiest queue
q0 = max(byq, key=lambda q: len(byq[q])); iv = byq[q0]
gaps = sorted(((iv[k + 1][0] - iv[k][1]) / 1e3, iv[k][2], iv[k + 1][2]) for k in range(len(iv) - 1))[-8:]
print("longest gaps on queue", q0, [(round(g, 1), a, b) for g, a, b in gaps])
