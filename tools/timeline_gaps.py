#!/usr/bin/env python3
"""Busy / idle analysis of a rocprofv3 --kernel-trace CSV: per step (separated by the longest idle gaps), the union of
kernel intervals vs the wall time, the biggest gaps and the kernels around them.  usage: timeline_gaps.py <kernel_trace.csv> [n_steps]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?")))
rows.sort()
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n_last:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, name, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  wall {(t1 - t0) / 1e6:.3f} ms  busy (union) {busy / 1e6:.3f} ms  idle {(t1 - t0 - busy) / 1e6:.3f} ms  "
      f"sum of kernel times {sum(e - s for s, e, _, _ in rows) / 1e6:.3f} ms")
gaps.sort(reverse=True)
print("largest gaps (us, at ms, next kernel):")
for g, at, name in gaps[:15]:
    print(f"  {g / 1e3:8.1f} us at {at / 1e6:8.3f} ms before {name}")
small = [g for g, _, _ in gaps if g < 20000]
print(f"gaps < 20 us: {len(small)}, total {sum(small) / 1e6:.3f} ms, median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us")
queues = {}
for s, e, name, q in rows:
    queues.setdefault(q, [0, 0])
    queues[q][0] += 1
    queues[q][1] += e - s
print("per queue:", {q: (n, round(t / 1e6, 3)) for q, (n, t) in queues.items()})
