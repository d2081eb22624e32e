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


def planted_scene(nl=73, mode="RGB", n_views=12, grid=100, seed=0, dense_eps=0.0):
    mesh = planted.gradient_textured_mesh(grid, 1024, seed)
    knots = planted.landmark_knots(nl, seed)
    pts = planted.surface_points(mesh, knots)
    sd = planted.planted_state_dict(nl, mode, knots, dense_eps=dense_eps)
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


def test_dense_planted_network_keeps_the_inlier_branch_on_the_oracle():
    """The same detector with dense random weights in all 138 convolutions (planted_state_dict(dense_eps=0.003)): the noise they
    add reaches +-20 % of a peak's height, the peaks still win, every landmark stays on the inlier branch."""
    mesh, pts, sd, poses = planted_scene(dense_eps=0.003)
    assert all(np.count_nonzero(v) == v.size for k, v in sd.items() if k.endswith(".weight") and v.ndim == 4)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        out, err, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses, sd,
                                             arch.CHANNEL_SELECT["RGB"])
    assert err < 10.0
    d = np.linalg.norm(out - pts, axis=1)
    assert d.max() < 8.0 and np.median(d) < 4.0
    sc = inter["maxima"][:, :, 2]
    assert sc.min() > 0.0 and sc.max() - sc.min() > 0.15   # the dense weights move the peak heights visibly (plain: 0.121..0.247)
