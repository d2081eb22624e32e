"""The hand-made detector of tests/planted.py on the CPU oracle: through all 138 live convolutions the final
heatmaps peak at the planted surface points, so the consensus takes its RANSAC *inlier* branch for every landmark
(SURVEY.md 8d "planted-peak"; with random weights it always takes the fail branch, error 1e8).
The same scene goes through the HIP path in tests/test_gpu_parity.py::test_planted_peaks_through_the_network."""
import contextlib
import io

import numpy as np

import planted
from mvlm_amd import arch
from oracle import pipeline as opipe
from oracle import poses as oposes


def planted_scene(nl=73, mode="RGB", n_views=12, grid=100, seed=0):
    mesh = planted.gradient_textured_mesh(grid, 1024, seed)
    knots = planted.landmark_knots(nl, seed)
    pts = planted.surface_points(mesh, knots)
    sd = planted.planted_state_dict(nl, mode, knots)
    np.random.seed(3)
    poses = oposes.generate_3d_transformations(n_views)
    return mesh, pts, sd, poses


def test_planted_peaks_reach_the_inlier_branch_on_the_oracle():
    mesh, pts, sd, poses = planted_scene()
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        out, err, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses, sd,
                                             arch.CHANNEL_SELECT["RGB"])
    assert err < 10.0                        # mean one-shot-RANSAC residual: no landmark fell back (that adds 1e8 / NL)
    d = np.linalg.norm(out - pts, axis=1)
    assert d.max() < 6.0 and np.median(d) < 4.0   # ~2 px of systematic offset (2x max-pool, the (row-1, col-0.5) rule)
