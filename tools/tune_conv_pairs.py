#!/usr/bin/env python3
"""Which independent convolutions of the hourglass should share a launch?  (runs on the GPU box)

At every hourglass level the skip block (up1*, at `size`) and the first block of the next lower level (low1*, at
size / 2) do not depend on each other (reference paulsenpredictor.py:301-361); conv j of both has the same channels.
For every such pair shape and device batch this times the two tuned single launches back to back
(mvlm_conv_pair_bench, variant -1) against ONE two-problem launch of every kernel variant that can serve both
(conv_pair_kernel), with K parts for the small levels, and writes the winners that beat the single launches by more
than 3 % to mvlm_amd/csrc/conv_pair_tuned.h (--write-header) - the table mvlm_conv_pair_variant() reads.

usage: python tools/tune_conv_pairs.py [--batches 1,2,...] [--write-header] [--out gpurun_out/conv_pair_tune.json]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
HEADER = REPO / "mvlm_amd" / "csrc" / "conv_pair_tuned.h"
N_VARIANTS = 33
# (cin, cout, flags) of a 256-channel residual block's conv1 / conv2 / conv3 (flags: 1 pre-BN, 2 residual + raw copy)
CONVS = [(256, 128, 3), (128, 64, 3), (64, 64, 3)]
SIZES = [128, 64, 32, 16, 8]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,20,24,32,48,64,96,128")
    ap.add_argument("--write-header", action="store_true")
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "conv_pair_tune.json"))
    ap.add_argument("--iters", type=int, default=8)
    args = ap.parse_args()

    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    lib = ctx.lib
    names = {v: lib.mvlm_conv_variant_name(v).decode() for v in range(N_VARIANTS)}
    base_ids = [v for v, n in names.items() if n.startswith("conv3x3_") and "c80" not in n and "c84" not in n and "c96" not in n]
    table = []
    for batch in [int(b) for b in args.batches.split(",")]:
        for size in SIZES:
            for cin, cout, flags in CONVS:
                ms = C.c_float()

                def run(variant):
                    best = None
                    for _ in range(2):  # best of two: the first call of a variant also sets its launch attributes
                        rc = lib.mvlm_conv_pair_bench(ctx.handle, batch, cin, cout, size, flags, variant, args.iters, C.byref(ms))
                        if rc != 0:
                            return None
                        best = ms.value if best is None else min(best, ms.value)
                    return best * 1e3

                single = run(-1)
                if single is None:
                    print("single launches failed", batch, size, cin, cout, lib.mvlm_last_error(ctx.handle), file=sys.stderr)
                    continue
                res = {}
                for v in base_ids:
                    sk = names[v].startswith("conv3x3_sk")
                    # K parts only where they pay for single launches too: split-K tiles, levels <= 16x16 / <= 8x8, small batches
                    lgs = [(0, 0)]
                    if sk and batch <= 32 and size <= 16:
                        lgs += [(0, 1), (0, 2)] + ([(1, 1), (1, 2), (2, 2)] if size <= 8 else [])
                    for l0, l1 in lgs:
                        code = v | (l0 << 8) | (l1 << 10)
                        t = run(code)
                        if t is not None:
                            res[code] = round(t, 2)
                if not res:
                    continue
                best = min(res, key=res.get)
                pname = lib.mvlm_conv_variant_name(best | 0x1000).decode()
                row = dict(batch=batch, cin=cin, cout=cout, size=size, flags=flags, single_us=round(single, 2), best=best,
                           best_name=pname, best_us=res[best], all_us={str(k): v for k, v in res.items()})
                table.append(row)
                mark = f"   <-- {100 * (1 - res[best] / single):.0f} % saved" if res[best] < 0.97 * single else ""
                print(f"B{batch:3d} {cin:3d}->{cout:3d} @{size:3d}|{size // 2:<3d} two launches {single:8.1f} us   pair {pname:34s} {res[best]:8.1f} us{mark}", flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(table, indent=0))
    for batch in sorted({r["batch"] for r in table}):
        rows = [r for r in table if r["batch"] == batch]
        t_single = sum(r["single_us"] for r in rows)
        t_pair = sum(min(r["single_us"], r["best_us"]) if r["best_us"] < 0.97 * r["single_us"] else r["single_us"] for r in rows)
        print(f"batch {batch}: 15 pairs per hourglass as single launches {t_single:.0f} us, with the winning pairs {t_pair:.0f} us "
              f"(x 2 hourglasses: {2 * (t_single - t_pair):.0f} us saved per forward pass)")
    if args.write_header:
        write_header(table)


def write_header(table):
    lines = ["// GENERATED by tools/tune_conv_pairs.py --write-header on an MI355X: pairs of independent 3x3 convolutions (conv j of a",
             "// hourglass level's skip block at `size` and conv j of the next level's first block at size / 2, same channels) that run",
             "// faster as ONE two-problem launch than as two tuned single launches.  {cin_pad, cout_pad, size, batch, variant}; variant =",
             "// base id | log2(kparts of the size problem) << 8 | log2(kparts of the size / 2 problem) << 10, or -1 = two launches;",
             "// sorted; the dispatcher uses the entry of the smallest tuned batch >= the launch's batch for the same shape.",
             "#ifndef MVLM_CONV_PAIR_TUNED_H", "#define MVLM_CONV_PAIR_TUNED_H",
             "struct ConvPairTuned { short cin_pad, cout_pad, size, batch, variant; };",
             "static const ConvPairTuned MVLM_CONV_PAIR_TUNED[] = {"]
    rows = {}
    for r in table:
        win = r["best_us"] < 0.97 * r["single_us"]
        rows[(r["cin"], r["cout"], r["size"], r["batch"])] = (r["best"] if win else -1, r)
    for key in sorted(rows):
        v, r = rows[key]
        what = f"{r['best_name']} {r['best_us']} us" if v >= 0 else f"two launches (best pair {r['best_name']} {r['best_us']} us)"
        lines.append(f"    {{{key[0]}, {key[1]}, {key[2]}, {key[3]}, {v}}},  // {what}; single launches {r['single_us']} us")
    if not rows:
        lines.append("    {0, 0, 0, 0, -1},")
    lines += ["};", f"static const int MVLM_CONV_PAIR_TUNED_N = {len(rows)};", "#endif", ""]
    HEADER.write_text("\n".join(lines))
    out_copy = REPO / "gpurun_out" / "conv_pair_tuned.h"
    out_copy.write_text("\n".join(lines))
    print(f"wrote {HEADER} ({len(rows)} entries) and {out_copy}")


if __name__ == "__main__":
    main()
