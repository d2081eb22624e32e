#!/usr/bin/env python3
"""Time every convolution kernel variant on every layer shape of the landmark network, per device batch.

Runs on the GPU box (through gpurun): for each distinct (ksize, cin, cout, size, kind) of
mvlm_amd.arch.conv_slots and each batch in --batches, mvlm_conv_bench launches each variant that can serve
the shape and reports ms per launch.  Output: a JSON table (gpurun_out/conv_tune.json) of all timings plus
the winner per (shape, batch), and a summary of where the dispatcher's own choice loses more than 3 %.
mvlm_amd/csrc/conv_mfma.hip's pick_variant() rules are derived from this table.

usage: python tools/tune_conv.py [--batches 8,12,16,24,32,48,64,96] [--nl 84] [--channels 4] [--out file]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

N_VARIANTS = 33
# tiles that fold several images into one MFMA column are only used at the level they were written for
NATIVE_WIDTH = {"conv3x3_c32_t8x8x2": 8, "conv3x3_c32_t4x4x8": 4, "conv3x3_sk_t4x4x2": 4, "conv3x3_sk16_t4x4x2": 4, "conv3x3_sk8_t4x4x2": 4}
HEADER = REPO / "mvlm_amd" / "csrc" / "conv_tuned.h"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,20,24,32,40,48,64,80,96,112,128")
    ap.add_argument("--nl", default="73,84")
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--write-header", action="store_true", help=f"write the winners to {HEADER.relative_to(REPO)}")
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "conv_tune.json"))
    ap.add_argument("--iters", type=int, default=6)
    args = ap.parse_args()

    from mvlm_amd import _lib, arch

    ctx = _lib.get_context(0)
    lib = ctx.lib
    sizes = arch.conv_spatial_sizes()
    shapes = {}
    for nl in [int(v) for v in args.nl.split(",")]:
        for s in arch.conv_slots(nl, args.channels):
            if not s.present or s.ksize == 2 or s.name == "conv11":  # conv11 runs as the four parity kernels (fixed variant)
                continue
            flags = (1 if s.pre_bn else 0) | (4 if s.has_bias else 0) | (8 if s.post_bn else 0)
            if ".conv1" in s.name or ".conv2" in s.name:
                flags |= 2
            key = (s.ksize, s.cin, s.cout, sizes[s.name], flags)
            users = shapes.setdefault(key, [])
            if s.name not in users:
                users.append(s.name)
    names = {v: lib.mvlm_conv_variant_name(v).decode() for v in range(N_VARIANTS)}
    # split-K tiles with the input channels divided over 2 / 4 workgroups per output tile (id + 256 log2(parts)): they
    # only pay at the 8x8 / 4x4 levels of small batches (measured: slower from 16x16 up)
    kpart_ids = [v + 256 * lg for v in range(N_VARIANTS) if names[v].startswith("conv3x3_sk") for lg in (1, 2)]
    names.update({v: lib.mvlm_conv_variant_name(v).decode() for v in kpart_ids})
    table = []
    for batch in [int(b) for b in args.batches.split(",")]:
        for (k, cin, cout, size, flags), users in sorted(shapes.items()):
            ms, used = C.c_float(), C.c_int()
            rc = lib.mvlm_conv_bench(ctx.handle, batch, cin, cout, k, size, flags, -2, args.iters, C.byref(ms), C.byref(used))
            if rc != 0:
                print("auto failed", k, cin, cout, size, lib.mvlm_last_error(ctx.handle), file=sys.stderr)
                continue
            auto_v, auto_ms = used.value, ms.value
            res = {}
            for v in list(range(N_VARIANTS)) + (kpart_ids if k == 3 and size <= 8 and batch <= 32 else []):
                if names[v] == "?" or not names[v].startswith(f"conv{k}x{k}") or NATIVE_WIDTH.get(names[v].split("_k")[0], size) != size:
                    continue
                rc = lib.mvlm_conv_bench(ctx.handle, batch, cin, cout, k, size, flags, v, args.iters, C.byref(ms), C.byref(used))
                if rc == 0:
                    res[names[v]] = round(ms.value * 1e3, 2)
            best = min(res, key=res.get)
            flop = 2.0 * cin * cout * k * k * size * size * batch
            if res[best] >= 0.97 * auto_ms * 1e3 and names[auto_v] in res:
                best = names[auto_v]  # within the noise of the rule-based choice: keep that
            row = dict(batch=batch, ksize=k, cin=cin, cout=cout, size=size, flags=flags, layers=len(users), example=users[0],
                       auto=names[auto_v], auto_us=round(auto_ms * 1e3, 2), best=best, best_us=res[best],
                       best_tflops=round(flop / (res[best] * 1e-6) / 1e12, 1), all_us=res)
            table.append(row)
            mark = "" if res[best] >= 0.97 * row["auto_us"] else f"   <-- {best} {res[best]:.1f} us"
            print(f"B{batch:3d} k{k} {cin:3d}->{cout:3d} @{size:3d} f{flags:2d} x{len(users):2d}  auto {names[auto_v]:20s} {row['auto_us']:8.1f} us"
                  f"  {flop / (auto_ms * 1e-3) / 1e12:6.1f} TF{mark}", flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(table, indent=0))
    for batch in sorted({r["batch"] for r in table}):
        rows = [r for r in table if r["batch"] == batch]
        t_auto = sum(r["auto_us"] * r["layers"] for r in rows)
        t_best = sum(r["best_us"] * r["layers"] for r in rows)
        print(f"batch {batch}: sum over layers auto {t_auto / 1e3:.2f} ms, best {t_best / 1e3:.2f} ms")
    if args.write_header:
        write_header(table, {n: v for v, n in names.items()})


def write_header(table, ids):
    """Entries where a variant beats the rule-based choice by more than 3 %, as a C table sorted by shape and batch:
    pick_variant() takes the entry of the smallest tuned batch >= the launch's batch."""
    from mvlm_amd import weights

    def pads(r):
        cin_pad = (r["cin"] + 7) // 8 * 8 if r["ksize"] == 1 else (r["cin"] + 3) // 4 * 4
        plain = (r["flags"] & 4) and not (r["flags"] & (1 | 2 | 8))
        cout_pad = (r["cout"] + 15) // 16 * 16 if plain and (r["cout"] + 15) // 16 * 16 in weights.COUT_TAIL_PADS else (r["cout"] + 31) // 32 * 32
        if plain and r["ksize"] == 3 and r["cout"] == weights.COUT_EXACT_84:
            cout_pad = 84  # conv6 / conv10 (conv11 is not tuned: fixed kernels)
        return cin_pad, cout_pad

    rows = {}
    for r in table:
        cin_pad, cout_pad = pads(r)
        rows[(r["ksize"], cin_pad, cout_pad, r["size"], r["batch"])] = (ids[r["best"]], ids[r["auto"]], r)
    lines = ["// GENERATED by tools/tune_conv.py --write-header on an MI355X: kernel variant per (layer shape, device batch)",
             "// where a measured variant beats pick_variant()'s rules by more than 3 %.  {ksize, cin_pad, cout_pad, size, batch, variant};",
             "// sorted; the dispatcher uses the entry of the smallest tuned batch >= the launch's batch for the same shape.",
             "#ifndef MVLM_CONV_TUNED_H", "#define MVLM_CONV_TUNED_H",
             "struct ConvTuned { short ksize, cin_pad, cout_pad, size, batch, variant; };",
             "static const ConvTuned MVLM_CONV_TUNED[] = {"]
    n = 0
    for key in sorted(rows):
        best, auto, r = rows[key]
        lines.append(f"    {{{key[0]}, {key[1]}, {key[2]}, {key[3]}, {key[4]}, {best}}},  // {r['best']} {r['best_us']} us (rules: {r['auto']} {r['auto_us']} us)")
        n += 1
    lines += ["};", f"static const int MVLM_CONV_TUNED_N = {n};", "#endif", ""]
    HEADER.write_text("\n".join(lines))
    out_copy = REPO / "gpurun_out" / "conv_tuned.h"
    out_copy.write_text("\n".join(lines))
    print(f"wrote {HEADER} ({n} entries) and {out_copy}")


if __name__ == "__main__":
    main()
