#!/usr/bin/env python3
"""How far do the landmarks move when the ONE numeric choice OpenGL leaves to the implementation changes?  The planted-peak
detector (tests/planted.py: every landmark channel peaks at a known surface point, the consensus runs its inlier branch)
end to end with the rasteriser at 8 sub-pixel bits (GPUs; the default) and at 4 (the software GL of tests/golden/gl_raster.npz):
pixels that differ, argmax pixels that differ, landmark displacement.  What it says: the reference's own output depends on the
OpenGL beneath its VTK by this much, which bounds what "equal to the VTK path" can mean without naming that OpenGL.
Test infrastructure (imports tests/ helpers).  usage: tests/reports/gl_sensitivity_report.py   -> profiles/rNN_gl_sensitivity.txt"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mvlm_amd import config  # noqa: E402
from mvlm_amd.pipeline import pipeline_from_config  # noqa: E402
from test_planted_cpu import planted_scene  # noqa: E402


def run(n_views, dense_eps=0.0):
    mesh, pts, sd, poses = planted_scene(n_views=n_views, dense_eps=dense_eps)
    pipe = pipeline_from_config(config.default_config("DTU3D", "RGB", n_views=n_views), weights=sd, verbose=False)
    out = {}
    for bits in (8, 4):
        pipe.renderer_3d.subpixel_bits = bits
        np.random.seed(1)
        lm, err = pipe.predict_mesh_device(mesh, poses)
        img = pipe.renderer_3d.render_device(mesh, poses)
        mx = pipe.predictor_2d.predict_device(img).cpu().numpy()
        out[bits] = (lm, err, img.cpu().numpy(), mx)
    (l8, e8, i8, m8), (l4, e4, i4, m4) = out[8], out[4]
    px = float((i8 != i4).any(-1).mean())
    am = ~np.all(m8[:, :, :2] == m4[:, :, :2], axis=2)
    step = np.abs(m8[:, :, :2] - m4[:, :, :2]).max(axis=2)[am]
    d = np.linalg.norm(l8 - l4, axis=1)
    t8, t4 = np.linalg.norm(l8 - pts, axis=1), np.linalg.norm(l4 - pts, axis=1)
    print(f"{n_views:3d} views{' dense' if dense_eps else '      '}: pixels differing {100 * px:.2f} %; argmax pixels differing {int(am.sum())} of {am.size} "
          f"({100 * am.mean():.1f} %, by at most {step.max() if am.any() else 0:.0f} px); RANSAC error {e8:.4f} / {e4:.4f}; "
          f"landmarks 8 vs 4 bits: median {np.median(d):.4f}, max {d.max():.4f} model units (1 px = 1.17); "
          f"distance to the planted truth: median {np.median(t8):.2f} / {np.median(t4):.2f}")
    return np.median(d), d.max()


if __name__ == "__main__":
    for n in (8, 16, 48, 96):
        run(n)
    run(16, dense_eps=0.003)
