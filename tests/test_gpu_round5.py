"""Round-5 GPU tests: the surface snap on meshes with stray points and degenerate triangles, BASELINE configs[4] as a whole
pipeline against the oracle, the 8-way shard of configs[3] against the single process, one result per scan across the
execution modes, the opt-in precisions at full size.  Run with -m gpu."""
import contextlib
import io
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]


def _face(grid, tex=64, seed=0):
    from mvlm_amd.utils.synthetic import face_like_mesh

    return face_like_mesh(grid, tex, seed)


# --------------------------------------------------------------------------------------
# surface snap (estimator3d.py:252-285): points of the file that no triangle uses, degenerate triangles
@pytest.mark.parametrize("grid", [12, 60])
def test_snap_ignores_stray_points_and_survives_degenerate_triangles(grid):
    """The .ply / .vtk / .stl / .wrl readers and mvlm_mesh_upload keep a file's point list as it is.  A point no triangle
    uses is not part of the surface (the reference: vtkCleanPolyData + a CELL locator, estimator3d.py:258-270), so it must
    neither attract a landmark nor - as an upper bound of the search that is closer than the surface - prune the real
    winner.  Zero-area triangles (three collinear corners; a repeated corner = a segment) stay cells of the surface."""
    from mvlm_amd.utils import HipEstimator3D
    from mvlm_amd.utils.mesh_io import Mesh
    from oracle import surface

    base = _face(grid, 16, 5)
    rs = np.random.RandomState(grid)
    v = base.verts.astype(np.float64)
    n_q = 64
    # queries 3-40 units off the surface, each with a stray point 0.01-0.5 units beside it (far closer than any triangle)
    tri = base.tris[rs.randint(0, base.n_tris, n_q)]
    on_surface = (v[tri[:, 0]] + v[tri[:, 1]] + v[tri[:, 2]]) / 3
    queries = on_surface + rs.uniform(3, 40, (n_q, 1)) * np.array([0.0, 0.0, 1.0]) + rs.normal(0, 2.0, (n_q, 3))
    strays = queries + rs.uniform(0.01, 0.5, (n_q, 1)) * rs.standard_normal((n_q, 3))
    # a collinear triangle and a repeated-corner triangle (a == b: the segment a-c) sticking out of the mesh at x = 200..260
    extra_v = np.array([[200.0, 0, 0], [230.0, 0, 0], [260.0, 0, 0], [200.0, 50, 0], [260.0, 50, 0]])
    n0 = base.n_verts + n_q
    extra_t = np.array([[n0, n0 + 1, n0 + 2], [n0 + 3, n0 + 3, n0 + 4]], np.int32)
    verts = np.concatenate([base.verts, strays.astype(np.float32), extra_v.astype(np.float32)])
    tris = np.concatenate([base.tris, extra_t])
    m = Mesh(verts=verts, tris=tris)
    pts = np.concatenate([queries,
                          [[215.0, 3.0, 1.0], [245.0, -2.0, 0.5],      # nearest: the collinear triangle's edge
                           [230.0, 53.0, 2.0], [205.0, 47.0, -1.0],    # nearest: the segment of the a == b triangle
                           [290.0, 25.0, 0.0]]])                        # beyond both
    got = HipEstimator3D(verbose=False).project_landmarks_to_surface(m, pts)
    want = surface.project_landmarks_to_surface(verts, tris, pts)
    assert np.isfinite(want).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    # none of the landmarks went to its stray point, and the answers are those of the mesh without the strays
    assert np.linalg.norm(got[:n_q] - strays, axis=1).min() > 1.0
    clean = surface.project_landmarks_to_surface(np.concatenate([base.verts, extra_v.astype(np.float32)]),
                                                 np.concatenate([base.tris, extra_t - n_q]), pts)
    np.testing.assert_allclose(got, clean, rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[n_q + 2], [230.0, 50.0, 0.0], atol=1e-9)  # foot of the perpendicular on the a-c segment
    np.testing.assert_allclose(got[n_q], [215.0, 0.0, 0.0], atol=1e-9)


def test_snap_with_only_unusable_bound_candidates():
    """Every triangle around the nearest used vertex is a point (a == b == c): the bound pass finds a candidate whose walk
    is still a finite distance, the search must return the same triangle the walk over all triangles picks."""
    from mvlm_amd.utils import HipEstimator3D
    from mvlm_amd.utils.mesh_io import Mesh
    from oracle import surface

    verts = np.array([[0, 0, 0], [10, 0, 0], [0, 10, 0], [5, 5, 30], [50, 50, 50]], np.float32)
    tris = np.array([[3, 3, 3], [0, 1, 2]], np.int32)     # a point cell above the one real triangle; vertex 4 is stray
    pts = np.array([[5.0, 5.0, 28.0], [5.0, 5.0, 10.0], [49.0, 49.0, 49.0], [2.0, 2.0, -3.0]])
    m = Mesh(verts=verts, tris=tris)
    got = HipEstimator3D(verbose=False).project_landmarks_to_surface(m, pts)
    want = surface.project_landmarks_to_surface(verts, tris, pts)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[0], [5, 5, 30], atol=1e-12)
    np.testing.assert_allclose(got[1], [5, 5, 0], atol=1e-12)
    assert np.linalg.norm(got[2] - verts[4]) > 20  # not the stray vertex
