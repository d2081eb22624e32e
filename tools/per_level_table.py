#!/usr/bin/env python3
"""Sums bench.py's MVLM_BENCH_PER_LAYER stderr table per spatial level: launches, microseconds and the time the same
FLOPs take at the fp32-matrix peak.  usage: per_level_table.py bench.err [views]"""
import re
import sys
from collections import defaultdict

PEAK = 157.3
rows = defaultdict(lambda: [0, 0.0, 0.0])
seen = set()
for line in open(sys.argv[1]):
    m = re.match(r"\s+slot\s+(\d+)\s+(\S+)\s+@\s*(\d+)\s+(\S+)\s+([\d.]+) us/launch\s+([\d.]+) TFLOP/s", line)
    if not m:
        continue
    slot, name, size, variant, us, tf = int(m[1]), m[2], int(m[3]), m[4], float(m[5]), float(m[6])
    if (slot, variant) in seen:          # the fast-mode table comes first in a default run: keep the last block only
        rows.clear()
        seen.clear()
    seen.add((slot, variant))
    r = rows[size]
    r[0] += 1
    r[1] += us
    r[2] += us * tf / PEAK
tot = [0, 0.0, 0.0]
print(f"{'level':>6} {'launches':>8} {'us':>10} {'us at peak':>11} {'frac':>6}")
for size in sorted(rows, reverse=True):
    n, us, ideal = rows[size]
    print(f"{size:>6} {n:>8} {us:>10.1f} {ideal:>11.1f} {ideal / us:>6.3f}")
    for i, v in enumerate((n, us, ideal)):
        tot[i] += v
print(f"{'all':>6} {tot[0]:>8} {tot[1]:>10.1f} {tot[2]:>11.1f} {tot[2] / tot[1]:>6.3f}")
