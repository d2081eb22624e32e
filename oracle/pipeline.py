"""Oracle: the whole ``predict_one_file`` path on the CPU, stage by stage.

Follows src/mvlm/pipeline/general_pipeline.py:67-131 with the oracle pieces:
raster (unpinned, VTK absent) -> cnn -> maxima -> rays -> consensus -> surface snap.
Used by the end-to-end parity test, ``smoke()`` and the ``cpu_baseline`` leg of
bench.py.  TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
from __future__ import annotations

import time

import numpy as np

from . import cnn, estimator, raster, surface


def predict_mesh(mesh_verts, mesh_tris, mesh_uvs, mesh_tex, transform_stack, state_dict, chan_sel,
                 mode="quantile", q=0.5, thr=0.5, batch_size=2, timings: dict | None = None, shading: str = "texture",
                 predictor=None):
    """-> (landmarks [NL,3] f64, mean error, intermediates dict).  Uses the global numpy RNG
    for the RANSAC draw like the reference (seed it for reproducibility).
    ``predictor``: any other Predictor2D's ``predict_landmarks_from_images`` (general_pipeline.py:90:
    image_stack -> (landmarks [NL,N,3], valid [N])) in place of the landmark network - the MediaPipe-shaped
    pipeline of BASELINE configs[4], whose detector is a third-party package; views with ``valid`` False are dropped
    before rays and consensus (general_pipeline.py:93-95)."""
    t = timings if timings is not None else {}
    t0 = time.time()
    images = raster.multiview_render(mesh_verts, mesh_tris, mesh_uvs, mesh_tex, transform_stack, shading=shading)
    t["render"] = time.time() - t0
    t0 = time.time()
    if predictor is not None:
        lms, valid = predictor(images)
    else:
        lms, valid = cnn.predict_landmarks_from_images(state_dict, images, chan_sel, batch_size=batch_size)
    t["prediction"] = time.time() - t0
    lms = lms[:, valid, :]
    poses = np.asarray(transform_stack)[valid]
    t0 = time.time()
    s, e = estimator.estimate_landmark_lines(256, lms, poses)
    t["lines"] = time.time() - t0
    t0 = time.time()
    draws = []
    pts, err = estimator.estimate_landmarks_from_lines(lms, s, e, mode, q, thr, draws=draws)
    t["consensus"] = time.time() - t0
    t0 = time.time()
    out = surface.project_landmarks_to_surface(mesh_verts, mesh_tris, pts)
    t["project"] = time.time() - t0
    return out, err, dict(images=images, maxima=lms, starts=s, ends=e, raw=pts, draws=draws)
