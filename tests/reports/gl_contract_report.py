#!/usr/bin/env python3
"""Per scene of tests/golden/gl_raster.npz (the reference renderer's GL work drawn by a real OpenGL): how the CPU oracle and -
on a GPU box - the HIP rasteriser compare with it, every disagreement counted into its class (tests/gl_contract.py).
Test infrastructure (imports oracle/).  usage: tests/reports/gl_contract_report.py   -> profiles/rNN_gl_contract.txt"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gl_contract import compare, load  # noqa: E402
from oracle import raster  # noqa: E402

meta, scenes = load()
bits = meta["gl"]["subpixel_bits"]
print(f"OpenGL: {meta['gl']}")
try:
    import torch

    gpu = torch.cuda.is_available()
except Exception:  # noqa: BLE001
    gpu = False


def hip(sc, b):
    from mvlm_amd.utils import HipRenderer3D, Mesh

    r = HipRenderer3D(n_views=len(sc["poses"]), verbose=False, subpixel_bits=b)
    out = r.render_device(Mesh(sc["verts"], sc["tris"], sc["uvs"], sc["tex"]), sc["poses"]).cpu().numpy()
    r.check()
    return out


print(f"{'scene':11s} {'views':>5s} {'pixels':>8s} {'covered':>8s} | at the GL's {bits} sub-pixel bits: clip texel depth+-1 (=24-bit) unexplained"
      f" | at 8 bits (default): pixels whose RGB differs | HIP == oracle (4 / 8 bits)")
tot = np.zeros(5, np.int64)
for name, sc in scenes.items():
    o4 = raster.multiview_render(sc["verts"], sc["tris"], sc["uvs"], sc["tex"], sc["poses"], subpixel_bits=bits)
    o8 = raster.multiview_render(sc["verts"], sc["tris"], sc["uvs"], sc["tex"], sc["poses"], subpixel_bits=8)
    r = compare(sc, o4)
    got8 = np.round(o8 * 255.0).astype(np.uint8)
    d8 = int((got8[..., :3] != sc["image_u8"][..., :3]).any(-1).sum())
    same = "-"
    if gpu:
        same = f"{np.array_equal(hip(sc, bits), o4)} / {np.array_equal(hip(sc, 8), o8)}"
    if name != "third":
        tot += [r["clip"], r["texel"], r["depth1"], r["d24"], r["unexplained"]]
    print(f"{name:11s} {sc['poses'].shape[0]:5d} {r['pixels']:8d} {r['covered']:8d} | {r['clip']:5d} {r['texel']:5d} {r['depth1']:6d} ({r['d24']:5d}) {r['unexplained']:5d}"
          f" | {d8:6d} ({100.0 * d8 / r['pixels']:.3f} %){' = lattice scene: bits cannot matter' if sc['lattice'] else ''} | {same}")
print(f"totals without the plane z = 0: clip {tot[0]}, texel {tot[1]}, depth +-1 {tot[2]} (of which equal under the 24-bit reading {tot[3]}), unexplained {tot[4]}")
