#!/usr/bin/env python3
"""One step of bench.py under rocprofv3 --kernel-trace --memory-copy-trace as a timeline: every kernel and copy with its start,
duration and the idle gap in front of it; steps are cut at the rasteriser's transform kernel.
usage: timeline_steps.py <kernel_trace.csv> <memory_copy_trace.csv> [step index to print, default 20]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:64]))
for r in csv.DictReader(open(sys.argv[2])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[12:]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("transform_kernel")]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
spans = [(rows[b][0] - rows[a][0], sum(e - s for s, e, _ in rows[a:b]), b - a) for a, b in zip(starts[:-1], starts[1:])]
med = sorted(w for w, _, _ in spans[8:])[len(spans[8:]) // 2]
print(f"{len(spans)} steps; transform-to-transform (under the profiler) median {med / 1e3:.1f} us; busy per step median "
      f"{sorted(b for _, b, _ in spans[8:])[len(spans[8:]) // 2] / 1e3:.1f} us; {spans[which][2]} kernels + copies per step")
seg = rows[starts[which] - 3:starts[which + 1] + 1]
t0 = prev = seg[0][0]
print(f"--- step {which}: start us, duration us, idle gap in front us")
for s, e, n in seg:
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f}  {max(0, s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
