#!/usr/bin/env python3
"""Ranks the variants of a tools/tune_conv.py JSON table per (batch, shape): the four fastest and every variant named on
the command line.  usage: tune_rank.py table.json [min_size] [variant-substring ...]"""
import json
import sys

t = json.load(open(sys.argv[1]))
min_size = int(sys.argv[2]) if len(sys.argv) > 2 else 0
watch = sys.argv[3:]
for r in t:
    if r["ksize"] != 3 or r["size"] < min_size:
        continue
    items = sorted(r["all_us"].items(), key=lambda kv: kv[1])
    top = " | ".join(f"{k.replace('conv3x3_', '')} {v}" for k, v in items[:4])
    extra = "  ||  " + "  ".join(f"{k.replace('conv3x3_', '')} {v}" for k, v in items if any(w in k for w in watch)) if watch else ""
    print(f"B{r['batch']:3d} {r['cin']:3d}->{r['cout']:3d} @{r['size']:3d} x{r['layers']}  {top}{extra}")
