#!/usr/bin/env python3
"""Kernel variant per layer kind, measured INSIDE the landmark network (runs on the GPU box).

tools/tune_conv.py times one layer at a time on idle data: the launch re-reads its tensors out of the MALL and has the
chip to itself.  In a forward pass a layer's tensors were written by other layers (from 12 views up they come from HBM),
and a grid with a fractional last round pays for it differently - the ranking of the tiles changes (round 4: conv4.conv1
of a 12-view pass ran at 104 TFLOP/s on the tile that wins the single-layer benchmark at 124).  This tool therefore
times every candidate where it will run: for each (ksize, cin_pad, cout_pad, size, kind) of the network's 3x3 layers and
each kernel variant that can serve it (mvlm_conv_variant_serves), the variant is forced for that key only
(mvlm_conv_set_override), the pass runs launch by launch with per-launch HIP events, and the key's launches (+ the pool
kernels, which a pool-capable tile makes unnecessary) are summed.  One sweep of coordinate descent over the keys, largest
first; winners that beat the current choice by more than 2 % stay in place for the keys after them.

usage: python tools/tune_in_network.py [--batches 8,12,...] [--write-header] [--out gpurun_out/conv_net_tune.json]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
import time
from collections import defaultdict
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
HEADER = REPO / "mvlm_amd" / "csrc" / "conv_tuned_net.h"
N_VARIANTS = 33
CAP = 1024


def profile_pass(pred, images, out, passes):
    """-> list of per-pass record lists [(slot, variant, ms, shape6)]"""
    ctx = pred.ctx
    slot, var = (C.c_int32 * CAP)(), (C.c_int32 * CAP)()
    fl, ms = (C.c_double * CAP)(), (C.c_float * CAP)()
    shapes = (C.c_int32 * (6 * CAP))()
    runs = []
    for _ in range(passes):
        pred.predict_device(images, out=out)
        n = ctx.lib.mvlm_cnn_get_profile(ctx.handle, slot, var, fl, ms, CAP)
        m = ctx.lib.mvlm_cnn_get_profile_shapes(ctx.handle, shapes, CAP)
        assert n == m and n > 0
        runs.append([(slot[i], var[i], ms[i] * 1e3, tuple(shapes[6 * i + k] for k in range(6))) for i in range(n)])
    return runs


def key_time(runs, key):
    """mean us per pass of the launches with this (ksize, cin_pad, cout_pad, size, kind) + all pool-kernel launches"""
    tot = []
    for rec in runs:
        tot.append(sum(t for s, v, t, sh in rec if (s >= 0 and sh[:5] == key and not (v & 0x1000)) or s == -1))
    return float(np.min(tot))  # the quietest pass: clocks and neighbours only ever add time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,20,24,32,48,64,96,128")
    ap.add_argument("--nets", default="dtu3d:geometry+depth,bu3dfe:RGB+depth")
    ap.add_argument("--write-header", action="store_true")
    ap.add_argument("--merge", action="store_true", help="keep the entries of the existing header that this run does not re-measure")
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "conv_net_tune.json"))
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--min-gain", type=float, default=0.02)
    ap.add_argument("--max-size", type=int, default=256, help="only keys of feature maps up to this size (re-tuning the small levels)")
    args = ap.parse_args()

    import torch

    from mvlm_amd.prediction import BU3DFEPredictor, DTU3DPredictor

    rows = []
    seen_keys = set()  # (key, batch) tuned by an earlier net: the second net only adds what the first does not have
    for net in args.nets.split(","):
        name, mode = net.split(":")
        pred = (DTU3DPredictor if name == "dtu3d" else BU3DFEPredictor)(image_mode=mode, weights="synthetic:0", verbose=False)
        ctx, lib = pred.ctx, pred.ctx.lib
        names = {v: lib.mvlm_conv_variant_name(v).decode() for v in range(N_VARIANTS)}
        nl = pred.get_lm_count()
        for batch in [int(b) for b in args.batches.split(",")]:
            rs = np.random.RandomState(batch)
            images = torch.from_numpy(rs.rand(batch, 256, 256, 4).astype(np.float32)).cuda()
            out = torch.empty((nl, batch, 3), dtype=torch.float32, device="cuda")
            pred.set_execution(graphs=False)
            ctx.check(lib.mvlm_conv_set_override(ctx.handle, 0, 0, 0, 0, 0, -1))
            pred.predict_device(images, out=out)  # first use: launch attributes, workspace
            ctx.check(lib.mvlm_cnn_set_profiling(ctx.handle, 1))
            t0 = time.time()
            base = profile_pass(pred, images, out, args.passes)
            total0 = min(sum(t for _, _, t, _ in rec) for rec in base)
            keys = defaultdict(float)
            current = {}
            for s, v, t, sh in base[0]:
                if s >= 0 and sh[0] == 3 and not (v & 0x1000) and sh[2] % 32 == 0 and sh[3] <= args.max_size:
                    keys[sh[:5]] += t
                    current[sh[:5]] = v
            gained = 0.0
            for key in sorted(keys, key=lambda k: -keys[k]):
                if (key, batch) in seen_keys:
                    continue
                seen_keys.add((key, batch))
                cands = [v for v in range(N_VARIANTS) if lib.mvlm_conv_variant_serves(v, *key)]
                if key[3] <= 8 and batch <= 32:
                    cands += [v + 256 * lg for v in range(N_VARIANTS) for lg in (1, 2) if lib.mvlm_conv_variant_serves(v + 256 * lg, *key)]
                cur = current[key]
                if len(cands) < 2:
                    continue
                res = {}
                for v in cands:
                    ctx.check(lib.mvlm_conv_set_override(ctx.handle, *key, v))
                    try:
                        res[v] = key_time(profile_pass(pred, images, out, args.passes), key)
                    except Exception as e:  # noqa: BLE001 - a variant that cannot run this layer after all
                        print("   ", lib.mvlm_conv_variant_name(v).decode(), "failed:", e, file=sys.stderr)
                if cur not in res:
                    res[cur] = key_time(base, key)
                best = min(res, key=res.get)
                win = best != cur and res[best] < (1.0 - args.min_gain) * res[cur]
                keep = best if win else cur
                ctx.check(lib.mvlm_conv_set_override(ctx.handle, *key, keep))
                if win:
                    gained += res[cur] - res[best]
                rows.append(dict(net=net, batch=batch, key=list(key), current=cur, current_name=lib.mvlm_conv_variant_name(cur).decode(),
                                 current_us=round(res[cur], 1), best=best, best_name=lib.mvlm_conv_variant_name(best).decode(),
                                 best_us=round(res[best], 1), kept=keep, all_us={str(k): round(v, 1) for k, v in res.items()}))
                mark = f"   <-- {lib.mvlm_conv_variant_name(best).decode()} {res[best]:.1f} us" if win else ""
                print(f"{net} B{batch:3d} k{key[0]} {key[1]:3d}->{key[2]:3d} @{key[3]:3d} kind {key[4]}  {lib.mvlm_conv_variant_name(cur).decode():24s} "
                      f"{res[cur]:9.1f} us{mark}", flush=True)
            final = profile_pass(pred, images, out, args.passes)
            total1 = min(sum(t for _, _, t, _ in rec) for rec in final)
            ctx.check(lib.mvlm_cnn_set_profiling(ctx.handle, 0))
            print(f"== {net} batch {batch}: conv kernels of a pass {total0 / 1e3:.3f} ms -> {total1 / 1e3:.3f} ms "
                  f"(sum of per-key gains {gained:.0f} us; {time.time() - t0:.0f} s)", flush=True)
            rows.append(dict(net=net, batch=batch, summary=True, before_us=round(total0, 1), after_us=round(total1, 1)))
            del images, out
        ctx.check(lib.mvlm_conv_set_override(ctx.handle, 0, 0, 0, 0, 0, -1))
        del pred
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rows, indent=0))
    if args.write_header:
        write_header(rows, args.merge)


def write_header(rows, merge=False):
    lines = ["// GENERATED by tools/tune_in_network.py --write-header on an MI355X: kernel variant per (layer shape, kind, device batch),",
             "// every candidate timed INSIDE a forward pass of the landmark network on real activations (HIP events per launch).",
             "// {ksize, cin_pad, cout_pad, size, kind, batch, variant}; kind 0 plain / residual-block layer, 1 scatter into the skip tensor,",
             "// 2 pooled output wanted; sorted; the dispatcher uses the entry of the smallest tuned batch >= the launch's batch.",
             "#ifndef MVLM_CONV_TUNED_NET_H", "#define MVLM_CONV_TUNED_NET_H",
             "struct ConvTunedNet { short ksize, cin_pad, cout_pad, size, kind, batch, variant; };",
             "static const ConvTunedNet MVLM_CONV_TUNED_NET[] = {"]
    import re

    table = {}
    if merge and HEADER.exists():
        for m in re.finditer(r"^    \{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (-?\d+)\},\s*// (.*)$", HEADER.read_text(), re.M):
            table[tuple(int(m[i]) for i in range(1, 7))] = (int(m[7]), m[8])
    for r in rows:
        if r.get("summary"):
            continue
        # (every tuned key is listed, changed or not: the dispatcher takes the entry of the smallest tuned batch >= B, so a key
        #  missing at one batch would inherit another batch's winner)
        kept_name = r["best_name"] if r["kept"] == r["best"] else r["current_name"]
        note = f"{kept_name} {r['best_us'] if r['kept'] == r['best'] else r['current_us']} us"
        if r["kept"] != r["current"]:
            note += f" (was {r['current_name']} {r['current_us']} us)"
        table[tuple(r["key"]) + (r["batch"],)] = (r["kept"], note + f" [{r['net']}]")
    for k in sorted(table):
        v, note = table[k]
        lines.append(f"    {{{k[0]}, {k[1]}, {k[2]}, {k[3]}, {k[4]}, {k[5]}, {v}}},  // {note}")
    n = sum(1 for ln in lines if ln.startswith("    {"))
    if n == 0:
        lines.append("    {0, 0, 0, 0, 0, 0, -1},")
    lines += ["};", f"static const int MVLM_CONV_TUNED_NET_N = {n};", "#endif", ""]
    HEADER.write_text("\n".join(lines))
    (REPO / "gpurun_out" / "conv_tuned_net.h").write_text("\n".join(lines))
    print(f"wrote {HEADER} ({len(table)} entries)")


if __name__ == "__main__":
    main()
