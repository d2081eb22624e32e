"""Parity of the HIP path against the CPU oracle and the reference-generated golden
vectors.  Everything here calls through the C ABI (libmvlm_hip.so).  Run with -m gpu."""
import contextlib
import ctypes as C
import io

import numpy as np
import pytest
import torch

from conftest import seeded_images

pytestmark = pytest.mark.gpu

MODES = {"RGB": 3, "depth": 1, "RGB+depth": 4, "geometry+depth": 2}


@pytest.fixture(scope="module")
def ctx():
    from mvlm_amd import _lib

    return _lib.get_context(0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# --------------------------------------------------------------------------------------
# convolution kernel vs torch (float64 reference; tolerance is fp32 accumulation noise)
CONV_CASES = [
    # cin, cout, size, k, batch, opts
    (256, 256, 32, 3, 2, dict(bias=True, post=True)),            # variant c128 8x32, two cout tiles
    (256, 128, 64, 3, 1, dict(pre=True, res=True)),              # c128
    (128, 64, 32, 3, 2, dict(pre=True)),                         # c64
    (64, 32, 32, 3, 1, dict(pre=True, res=True)),                # c32 16x32
    (256, 73, 32, 3, 1, dict(bias=True)),                        # c80: 73 -> 64 + one 16-row strip
    (73, 73, 64, 3, 2, dict(bias=True, up=True)),                # c80, cin 73 -> 76, upsampled input (conv11 shape)
    (84, 84, 64, 3, 1, dict(bias=True, up=True)),                # cin padding 84 -> 88, upsampled input
    (73, 256, 32, 3, 1, dict(bias=True, res=True)),              # cin 73 -> 80
    (256, 84, 32, 3, 2, dict(bias=True)),                        # c84: 84 = 64 + a 16-row + a 4-row strip (conv6 / conv10)
    (40, 84, 64, 3, 1, dict(bias=True)),                         # c84, ten 4-channel chunks, two row tiles per image
    (3, 64, 64, 3, 1, dict(bias=True, post=True)),               # conv1-like
    (64, 128, 32, 1, 2, dict(pre=True)),                         # 1x1 resample
    (256, 128, 16, 3, 3, dict(pre=True, res=True)),              # 16x16 tiles
    (128, 64, 16, 3, 2, dict(pre=True)),
    (256, 128, 8, 3, 5, dict(pre=True, res=True)),               # 8x8, 4 images per tile, ragged batch
    (64, 64, 8, 3, 4, dict(pre=True)),
    (256, 128, 4, 3, 19, dict(pre=True, res=True)),              # 4x4, 16 images per tile, ragged batch
    (64, 64, 4, 3, 16, dict(pre=True)),
]


@pytest.mark.parametrize("cin,cout,size,k,batch,opts", CONV_CASES)
def test_conv2d_matches_torch(ctx, cin, cout, size, k, batch, opts):
    rs = np.random.RandomState(cin * 7 + cout + size)
    up = opts.get("up", False)
    s_in = size // 2 if up else size
    x = rs.standard_normal((batch, cin, s_in, s_in)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    bias = rs.standard_normal(cout).astype(np.float32) if opts.get("bias") else None
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), rs.standard_normal(cin).astype(np.float32) * 0.3) if opts.get("pre") else None
    post = (rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.standard_normal(cout).astype(np.float32) * 0.3) if opts.get("post") else None
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32) if opts.get("res") else None

    xd, yd = dev(x), torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    rd = dev(res) if res is not None else None
    p = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, k, p(bias),
                                  p(pre[0]) if pre else None, p(pre[1]) if pre else None,
                                  p(post[0]) if post else None, p(post[1]) if post else None,
                                  C.c_void_p(rd.data_ptr()) if rd is not None else None, int(up), C.c_void_p(yd.data_ptr())))
    got = yd.cpu().numpy()

    t = torch.from_numpy(x).double()
    if pre:
        t = torch.relu(t * torch.from_numpy(pre[0]).double()[None, :, None, None] + torch.from_numpy(pre[1]).double()[None, :, None, None])
    if up:
        t = torch.nn.functional.interpolate(t, scale_factor=2, mode="nearest")
    y = torch.nn.functional.conv2d(t, torch.from_numpy(w).double(), None if bias is None else torch.from_numpy(bias).double(), 1, k // 2)
    if post:
        y = torch.relu(y * torch.from_numpy(post[0]).double()[None, :, None, None] + torch.from_numpy(post[1]).double()[None, :, None, None])
    if res is not None:
        y = y + torch.from_numpy(res).double()
    want = y.numpy()
    # fp32 fma chain over K = cin*k*k terms of O(1/sqrt(K)) magnitude: error ~ 1e-7 * sqrt(K)
    tol = 5e-6 * max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() < tol


# --------------------------------------------------------------------------------------
# renderer: bit-exact against the CPU restatement (both implement DESIGN.md's contract)
def _mesh(grid, tex, seed=0, textured=True):
    from mvlm_amd.utils.synthetic import face_like_mesh

    m = face_like_mesh(grid, tex, seed)
    if not textured:
        m.uvs, m.texture = None, None
    return m


# grids 3 and 4: cells of 85 / 57 pixels - triangles on both sides of the 64-pixel extent below which the kernels take
# their exact 24-bit integer path (raster.hip), in one render (foreshortened views shrink some of them under it)
@pytest.mark.parametrize("grid,n_views,textured", [(40, 8, True), (40, 8, False), (224, 24, True), (3, 24, True), (4, 24, False)])
def test_render_bit_exact(grid, n_views, textured):
    from mvlm_amd.utils import HipRenderer3D
    from oracle import raster

    m = _mesh(grid, 128, 1, textured)
    r = HipRenderer3D(n_views=n_views, verbose=False)
    np.random.seed(0)
    poses = r.generate_3d_transformations()
    got = r.render_device(m, poses).cpu().numpy()
    want = raster.multiview_render(m.verts, m.tris, m.uvs, m.texture, poses)
    assert got.shape == (n_views, 256, 256, 4)
    np.testing.assert_array_equal(got, want)
    cover = (want[..., 3] != np.float32(1 / 255)).mean()
    assert 0.1 < cover < 0.9  # the mesh is really in view


def test_render_samples_the_last_texel():
    """The texel is fetched with ONE 4-byte load at byte 3 * index (raster.hip); for the last texel of the image that load
    ends one byte behind the texture, inside the spare bytes the upload reserves (api.hip).  A 256 x 256 texture is exactly
    three 64-KB allocation granules, so without the reserve the load would leave the buffer: every pixel of a window-filling
    triangle samples that texel."""
    from mvlm_amd.utils import HipRenderer3D, Mesh
    from oracle import raster

    verts = np.array([[-400, -400, 0], [400, -400, 0], [0, 600, 0]], np.float32)
    tris = np.array([[0, 1, 2]], np.int32)
    uvs = np.tile(np.array([[0.999, 0.001]], np.float32), (3, 1))   # -> column 255, bottom row = the last texel in memory
    tex = np.random.RandomState(3).randint(0, 256, (256, 256, 3)).astype(np.uint8)
    tex[-1, -1] = (11, 222, 133)
    m = Mesh(verts, tris, uvs, tex)
    r = HipRenderer3D(n_views=8, verbose=False)
    poses = r.generate_3d_transformations()
    got = r.render_device(m, poses).cpu().numpy()
    want = raster.multiview_render(m.verts, m.tris, m.uvs, m.texture, poses)
    np.testing.assert_array_equal(got, want)
    hit = got[..., 3] != np.float32(1 / 255)
    assert hit.mean() > 0.9
    np.testing.assert_array_equal(got[hit][:, :3], np.tile(np.float32([11, 222, 133]) / np.float32(255), (int(hit.sum()), 1)))


def test_render_degenerate_and_offscreen_triangles():
    from mvlm_amd.utils import HipRenderer3D, Mesh
    from oracle import raster

    verts = np.array([[0, 0, 0], [50, 0, 0], [0, 50, 0],           # ordinary
                      [10, 10, 5], [10, 10, 5], [30, 30, 5],         # degenerate (zero area)
                      [400, 400, 0], [450, 400, 0], [400, 450, 0],   # outside the view box
                      [-200, -200, -10], [200, -200, -10], [0, 250, -10],  # huge, covers the window
                      [0, 0, 0], [0, 50, 0], [50, 0, 0]], np.float32)  # same as #0 with opposite winding
    tris = np.arange(15, dtype=np.int32).reshape(5, 3)
    m = Mesh(verts, tris)
    r = HipRenderer3D(n_views=8, verbose=False)
    poses = r.generate_3d_transformations()
    got = r.render_device(m, poses).cpu().numpy()
    want = raster.multiview_render(m.verts, m.tris, None, None, poses)
    np.testing.assert_array_equal(got, want)


# --------------------------------------------------------------------------------------
def test_heatmap_maxima_golden(ctx, golden):
    g = golden("maxima.npz")
    hm = g["heatmaps"]
    nl, s = hm.shape[0], hm.shape[1]
    hd = dev(hm[None])
    for method, key in ((0, "out_simple"), (1, "out_moment")):
        out = torch.empty((nl, 1, 3), dtype=torch.float32, device="cuda")
        ctx.check(ctx.lib.mvlm_heatmap_maxima(ctx.handle, C.c_void_p(hd.data_ptr()), 1, nl, s, method, C.c_void_p(out.data_ptr())))
        np.testing.assert_array_equal(out.cpu().numpy()[:, 0, :], g[key])


@pytest.mark.parametrize("tag,n", [("8", 8), ("64", 64)])
def test_lines_golden(golden, tag, n):
    from mvlm_amd.utils import HipEstimator3D

    g = golden("estimator.npz")
    e3 = HipEstimator3D(verbose=False)
    s, e = e3.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), g[f"lines_{tag}_lms"], g[f"poses_{n}"])
    # float64 sums of three products; the reference's BLAS may fuse differently: 1e-9 model units
    np.testing.assert_allclose(s, g[f"lines_{tag}_s"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(e, g[f"lines_{tag}_e"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag", ["q64", "q8", "qfail", "abs", "absfew", "q128x478"])
def test_consensus_golden(golden, tag):
    from mvlm_amd.utils import HipEstimator3D

    g = golden("estimator.npz")
    lms, poses = g[f"fuse_{tag}_lms"], g[f"fuse_{tag}_poses"]
    mode = ["quantile", "absolute"][int(g[f"fuse_{tag}_cfg"][0])]
    e3 = HipEstimator3D(mode=mode, threshold_quantile=float(g[f"fuse_{tag}_cfg"][1]),
                        threshold_absolute=float(g[f"fuse_{tag}_cfg"][2]), verbose=False)
    s, e = e3.estimate_landmark_lines(np.zeros((lms.shape[1], 256, 256, 4), np.float32), lms, poses)
    ks = []
    orig = np.random.randint

    def rec(low, high=None, size=None, dtype=int):
        # one call per landmark, or ONE call of size (NL, 8) when every landmark keeps the same number of lines
        ks.extend([high] * (size[0] if isinstance(size, tuple) else 1))
        return orig(low, high, size, dtype)

    np.random.seed(1)  # same seed as the generator: the host draws must coincide
    np.random.randint = rec  # the product draws through randint(0, k, 8) == choice(range(k), 8)
    try:
        out, err = e3.estimate_landmarks_from_lines(lms, s, e)
    finally:
        np.random.randint = orig
    np.testing.assert_array_equal(np.array(ks), g[f"fuse_{tag}_draw_k"])
    np.testing.assert_allclose(out, g[f"fuse_{tag}_out"], rtol=0, atol=1e-8)
    assert abs(err - float(g[f"fuse_{tag}_err"])) <= 1e-9 * max(1.0, abs(float(g[f"fuse_{tag}_err"])))


def test_consensus_errors():
    from mvlm_amd.utils import HipEstimator3D

    e3 = HipEstimator3D(mode="nonsense", verbose=False)
    lms = np.zeros((3, 8, 3), np.float32)
    z = np.zeros((3, 8, 3))
    with pytest.raises(ValueError, match="Unknown mode"):
        e3.estimate_landmarks_from_lines(lms, z, z)


def test_project_to_surface_matches_oracle():
    from mvlm_amd.utils import HipEstimator3D
    from oracle import surface

    m = _mesh(60, 64, 2)
    rs = np.random.RandomState(5)
    pts = rs.uniform(-120, 120, (50, 3))
    pts[:5] = m.verts[rs.randint(0, m.n_verts, 5)]  # exactly on vertices
    got = HipEstimator3D(verbose=False).project_landmarks_to_surface(m, pts)
    want = surface.project_landmarks_to_surface(m.verts, m.tris, pts)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)


@pytest.mark.parametrize("grid,n_points", [(60, 500), (224, 70), (3, 33)])
def test_project_to_surface_bound_never_excludes_the_winner(grid, n_points):
    """The snap tests a triangle exactly only if a sphere bound cannot exclude it (surface.hip: distance to the nearest
    vertex as upper bound, |pa| - max edge as lower bound).  Points where that bound is tight or useless - on vertices, on
    edge midpoints (two triangles tie), on face centres, far outside the mesh, between groups of 16 landmarks (500 = 31 x 16
    + 4) - against the oracle's walk over all triangles."""
    from mvlm_amd.utils import HipEstimator3D
    from oracle import surface

    m = _mesh(grid, 64, 3)
    rs = np.random.RandomState(grid)
    v, t = m.verts.astype(np.float64), m.tris
    k = n_points // 5
    tri = t[rs.randint(0, m.n_tris, k)]
    pts = np.concatenate([
        v[rs.randint(0, m.n_verts, k)],                                   # on vertices (distance 0, many triangles tie)
        0.5 * (v[tri[:, 0]] + v[tri[:, 1]]),                                # on edge midpoints
        (v[tri[:, 0]] + v[tri[:, 1]] + v[tri[:, 2]]) / 3 + rs.normal(0, 1e-3, (k, 3)),  # just off face centres
        rs.uniform(-1000, 1000, (k, 3)),                                    # far outside: the bound excludes nothing much
        rs.uniform(-120, 120, (n_points - 4 * k, 3)),                       # around the surface
    ])
    got = HipEstimator3D(verbose=False).project_landmarks_to_surface(m, pts)
    want = surface.project_landmarks_to_surface(m.verts, m.tris, pts)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    assert np.abs(got[:k] - pts[:k]).max() < 1e-12  # a point on a vertex stays there


# --------------------------------------------------------------------------------------
# the whole network against vectors produced by the reference's own MVLMModel
def _near_tie_ok(heat_plane, got_rc, want_rc, rel=2e-4):
    """Both argmax pixels hold (nearly) the maximum of the oracle's heatmap plane."""
    gv = heat_plane[int(got_rc[0]) + 1, int(got_rc[1] + 0.5)]
    wv = heat_plane[int(want_rc[0]) + 1, int(want_rc[1] + 0.5)]
    return abs(gv - wv) <= rel * max(abs(wv), 1.0)


@pytest.mark.parametrize("nl,mode", [(73, "RGB"), (84, "RGB+depth"), (73, "geometry+depth"), (84, "depth")])
def test_full_network_against_reference_vectors(golden, nl, mode):
    from mvlm_amd import arch, weights
    from mvlm_amd.prediction import BU3DFEPredictor, DTU3DPredictor
    from oracle import cnn as ocnn

    g = golden("cnn_full.npz")
    tag = f"{nl}_{mode}"
    seed, img_seed = (int(v) for v in g[f"{tag}_seed"])
    imgs = seeded_images(img_seed, 2)
    cls = BU3DFEPredictor if nl == 84 else DTU3DPredictor
    pred = cls(image_mode=mode, weights=f"synthetic:{seed}", verbose=False)
    heat = pred.heatmaps_device(dev(imgs)).cpu().numpy()
    ref_sub = g[f"{tag}_heat_sub"]
    scale = np.abs(ref_sub).max()
    # 139 fp32 conv layers with a different (but still exact-fp32) summation order than
    # oneDNN: the reference's own NCHW vs channels-last kernels differ by 2e-5 relative
    assert np.abs(heat[:, :, 5::16, 3::16] - ref_sub).max() < 2e-4 * scale
    lms, valid = pred.predict_landmarks_from_images(imgs)
    assert valid.all() and lms.shape == (nl, 2, 3)
    want = g[f"{tag}_maxima"]
    # the fused-argmax path and the materialised heatmaps agree with each other exactly
    np.testing.assert_array_equal(lms, ocnn.maxima_fast(torch.from_numpy(heat)))
    sd = weights.synthetic_state_dict(nl, MODES[mode], seed=seed)
    _, _, oheat = ocnn.predict_landmarks_from_images(sd, imgs, arch.CHANNEL_SELECT[mode], return_heatmaps=True)
    oheat = oheat.numpy()
    flips = 0
    for lm in range(nl):
        for v in range(2):
            if not np.array_equal(lms[lm, v, :2], want[lm, v, :2]):
                flips += 1
                assert _near_tie_ok(oheat[v, lm], lms[lm, v], want[lm, v]), (lm, v, lms[lm, v], want[lm, v])
            assert abs(lms[lm, v, 2] - want[lm, v, 2]) < 2e-4 * scale
    assert flips <= 0.02 * nl * 2, f"{flips} argmax differences"


def test_moment_selection_runs_and_matches_oracle():
    from mvlm_amd import arch, weights
    from mvlm_amd.prediction import DTU3DPredictor
    from oracle import cnn as ocnn

    imgs = seeded_images(77, 1)
    pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:9", selection_method="moment", verbose=False)
    lms, _ = pred.predict_landmarks_from_images(imgs)
    heat = pred.heatmaps_device(dev(imgs)).cpu().numpy()
    want = ocnn.maxima_from_heatmaps(heat, "moment")  # oracle's moment rule on the same heatmaps
    np.testing.assert_array_equal(lms, want)


# --------------------------------------------------------------------------------------
def _e2e(n_views, grid, nl_cls, seed):
    import tempfile
    from pathlib import Path

    from mvlm_amd import pipeline, weights
    from mvlm_amd.utils.mesh_io import load_obj
    from mvlm_amd.utils.synthetic import write_face_like_obj
    from oracle import pipeline as opipe

    with tempfile.TemporaryDirectory() as td:
        obj = write_face_like_obj(Path(td) / "face.obj", grid=grid, tex_size=128, seed=seed)
        pipe = pipeline.create_pipeline(nl_cls, n_views=n_views, weights=f"synthetic:{seed}", verbose=False)
        np.random.seed(0)
        poses = pipe.renderer_3d.generate_3d_transformations()
        mesh = load_obj(obj)
        np.random.seed(1)
        got, gerr = pipe.predict_mesh_device(mesh, poses)
        nl = pipe.get_lm_count()
        sd = weights.synthetic_state_dict(nl, 4, seed=seed)
        np.random.seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            want, werr, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses, sd, (0, 1, 2, 3))
        gmax = pipe.predictor_2d.predict_device(pipe.renderer_3d.render_device(mesh, poses)).cpu().numpy()
    return got, gerr, want, werr, inter, gmax, pipe, mesh, poses


def test_end_to_end_8_views_against_oracle():
    got, gerr, want, werr, inter, gmax, pipe, mesh, poses = _e2e(8, 51, "dtu3d", 7)
    same = np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=(1, 2))
    # landmarks whose every view picked the oracle's pixel must agree to 1e-3 model units
    # (BASELINE.json north_star); the others differ by an argmax near-tie of one view
    assert same.mean() > 0.9
    assert np.abs(got[same] - want[same]).max() < 1e-3
    # with identical maxima the rest of the path (rays, consensus, snap) is ~1e-9
    e3 = pipe.estimator_3d
    s, e = e3.estimate_landmark_lines(np.zeros((8, 256, 256, 4), np.float32), inter["maxima"], poses)
    np.random.seed(1)
    pts, err = e3.estimate_landmarks_from_lines(inter["maxima"], s, e)
    np.testing.assert_allclose(pts, inter["raw"], rtol=0, atol=1e-8)
    snapped = e3.project_landmarks_to_surface(mesh, pts)
    np.testing.assert_allclose(snapped, want, rtol=0, atol=1e-8)
    assert abs(err - werr) <= 1e-9 * max(1.0, abs(werr))


def test_slot_protocol_equals_fused_path(tmp_path):
    """Driving the three slots through the reference's numpy protocol gives the fused result."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=40, tex_size=64, seed=1)
    pipe = pipeline.create_pipeline("BU3DFE", n_views=8, weights="synthetic:3", verbose=False)
    assert pipe.get_lm_count() == 84
    np.random.seed(1)
    fused = pipe.predict_one_file(obj)
    np.random.seed(1)
    slots = pipe._predict_slots(obj)
    np.testing.assert_array_equal(fused, slots)
    assert pipe.predict_one_file(tmp_path / "missing.obj") is None
    with pytest.raises(ValueError):
        pipe.renderer_3d.multiview_render(tmp_path / "face.jpg")


# --------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE.json sizes)
def test_full_size_round_trip_478_landmarks_128_views():
    """Points on the 100k-triangle mesh -> projected into 128 views -> fused on the GPU:
    the renderer's pose convention and the estimator's inverse round-trip."""
    from mvlm_amd.utils import HipEstimator3D, HipRenderer3D, view_rotations

    m = _mesh(224, 64, 4)
    rs = np.random.RandomState(11)
    pts = m.verts[rs.choice(m.n_verts, 478, replace=False)].astype(np.float64)
    r = HipRenderer3D(n_views=128, verbose=False)
    np.random.seed(0)
    poses = r.generate_3d_transformations()
    rot = view_rotations(poses).reshape(-1, 3, 3)
    lms = np.empty((478, 128, 3), np.float32)
    for v in range(128):
        q = pts @ rot[v].T
        lms[:, v, 1] = (q[:, 0] + 150) / 300 * 256           # col
        lms[:, v, 0] = 255 - (q[:, 1] + 150) / 300 * 256      # row
        lms[:, v, 2] = rs.rand(478)
    e3 = HipEstimator3D(verbose=False)
    s, e = e3.estimate_landmark_lines(np.zeros((128, 256, 256, 4), np.float32), lms, poses)
    np.random.seed(3)
    out, err = e3.estimate_landmarks_from_lines(lms, s, e)
    assert np.abs(out - pts).max() < 1e-3   # float32 pixel coordinates -> ~1e-5 units
    snapped = e3.project_landmarks_to_surface(m, out)
    assert np.abs(snapped - pts).max() < 1e-3
    assert err < 1e-6


def test_full_size_render_is_deterministic_and_view_consistent():
    from mvlm_amd.utils import HipRenderer3D

    m = _mesh(224, 256, 0)
    r = HipRenderer3D(n_views=96, verbose=False)
    np.random.seed(0)
    poses = r.generate_3d_transformations()
    a = r.render_device(m, poses)
    b = r.render_device(m, poses)
    assert torch.equal(a, b)                       # bin order is racy, the image must not be
    c = r.render_device(m, poses[40:41])
    assert torch.equal(a[40:41], c)                # a view does not depend on its batch


# --------------------------------------------------------------------------------------
# view sharding end to end: two ranks (sharing the one GPU of the test box, gloo) must return
# exactly what a single process returns
def _shard_worker(rank, world, port, obj, q, backend="gloo"):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":  # = RCCL; the recipe of INTEGRATION.md "Multi-GPU"
        from mvlm_amd import parallel

        parallel.rccl_environment()  # the same environment bench.py's launcher gives its ranks (this child has not touched the GPU yet)
        os.environ["MVLM_DIST_WORLD_OF_ONE"] = "1"  # a group of one rank still takes the sharded path (parallel.is_distributed)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mvlm_amd import pipeline

    # one rank per GPU under RCCL; the gloo rehearsal keeps every rank on the test box's one GPU
    pipe = pipeline.create_pipeline("dtu3d", n_views=12, weights="synthetic:5", verbose=False, shard_views=True,
                                    device=rank if backend == "nccl" else 0)
    np.random.seed(4 if rank == 0 else 99)  # only rank 0's RNG may matter
    out = pipe.predict_one_file(obj)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", [
    "gloo",
    pytest.param("nccl", marks=pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with more than one rank needs one GPU per rank")),
])
def test_sharded_views_equal_single_process(tmp_path, backend):
    """Two ranks, 6 of the 12 views each, must return exactly what a single process returns: over gloo with both ranks on
    the test box's one GPU, and - on any machine that shows two GPUs - over RCCL ("nccl") with one rank per device, which is
    the form bench.py --gpus N and the driver's 8-GPU run use (the reference's own multi-GPU mechanism is nn.DataParallel,
    paulsenpredictor.py:100-105)."""
    import socket

    import torch.multiprocessing as mp

    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=40, tex_size=64, seed=2)
    pipe = pipeline.create_pipeline("dtu3d", n_views=12, weights="synthetic:5", verbose=False)
    np.random.seed(4)
    want = pipe.predict_one_file(obj)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, obj, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0], want)
    np.testing.assert_array_equal(res[1], want)


def test_sharded_path_on_rccl_world_of_one(tmp_path):
    """The sharded code path with backend "nccl" (= RCCL): the test box has one GPU, so the world is one rank - the
    collectives (pose broadcast, draw broadcast, all-gather of the maxima) still go through RCCL with the tensors,
    dtypes and devices the 8-GPU run uses, and the result must equal the unsharded pipeline's."""
    import socket

    import torch.multiprocessing as mp

    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=40, tex_size=64, seed=2)
    pipe = pipeline.create_pipeline("dtu3d", n_views=12, weights="synthetic:5", verbose=False)
    np.random.seed(4)
    want = pipe.predict_one_file(obj)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_shard_worker, args=(0, 1, port, obj, q, "nccl"))
    p.start()
    try:
        rank, got = q.get(timeout=240)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    np.testing.assert_array_equal(got, want)


def test_cli_writes_landmark_files(tmp_path):
    """python -m mvlm_amd mirrors the reference's main.py output convention (main.py:62)."""
    from mvlm_amd.__main__ import main
    from mvlm_amd.utils.synthetic import write_face_like_obj

    write_face_like_obj(tmp_path / "a.obj", grid=30, tex_size=32, seed=1)
    write_face_like_obj(tmp_path / "b.obj", grid=30, tex_size=32, seed=2)
    out = tmp_path / "out"
    assert main(["-p", str(tmp_path), "-o", str(out), "-n", "8", "--weights", "synthetic:1", "--pipelines", "dtu3d"]) == 0
    for stem in ("a", "b"):
        lm = np.loadtxt(out / f"{stem}_dtu3d.txt", delimiter=",")
        assert lm.shape == (73, 3) and np.isfinite(lm).all()
    assert main(["-p", str(tmp_path / "nope")]) == 1


@pytest.mark.parametrize("n", [512, 1024])
def test_fusion_at_eight_gpu_view_counts_matches_oracle(n):
    """Weak scaling puts 64 views per GPU on one mesh: at 8 GPUs every rank fuses 512 views per landmark
    (1024 = the kernels' documented limit).  Rays, filter, RANSAC draw and refit against the CPU oracle."""
    from mvlm_amd.utils import HipEstimator3D, HipRenderer3D
    from oracle import estimator as oest

    nl = 73
    rs = np.random.RandomState(n)
    np.random.seed(5)
    poses = HipRenderer3D(n_views=n, verbose=False).generate_3d_transformations()
    pts = rs.uniform(-60, 60, (nl, 3))
    from mvlm_amd.utils import view_rotations

    rot = view_rotations(poses).reshape(-1, 3, 3)
    lms = np.empty((nl, n, 3), np.float32)
    for v in range(n):
        q = pts @ rot[v].T
        lms[:, v, 1] = (q[:, 0] + 150) / 300 * 256 + rs.normal(0, 0.7, nl)
        lms[:, v, 0] = 255 - (q[:, 1] + 150) / 300 * 256 + rs.normal(0, 0.7, nl)
        lms[:, v, 2] = rs.rand(nl)
    bad = rs.rand(nl, n) < 0.1                       # gross outliers among the survivors too
    lms[bad, 0] = rs.uniform(0, 255, bad.sum())
    e3 = HipEstimator3D(verbose=False)
    s, e = e3.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), lms, poses)
    os_, oe = oest.estimate_landmark_lines(256, lms, poses)
    np.testing.assert_allclose(s, os_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(e, oe, rtol=0, atol=1e-9)
    np.random.seed(8)
    out, err = e3.estimate_landmarks_from_lines(lms, s, e)
    np.random.seed(8)
    want, werr = oest.estimate_landmarks_from_lines(lms, os_, oe)
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-8)
    assert abs(err - werr) <= 1e-9 * max(1.0, abs(werr))
    with pytest.raises(ValueError, match="1..1024 views"):
        big = np.zeros((2, 1025, 3), np.float32)
        e3.estimate_landmarks_from_lines(big, np.zeros((2, 1025, 3)), np.ones((2, 1025, 3)))


def test_clip_rays_matches_oracle_and_the_depth_buffer(tmp_path):
    """mvlm_clip_rays_to_mesh: (a) equals the brute-force CPU twin, misses included; (b) a ray through a
    pixel centre ends where that view's depth buffer says the surface is (depth-aware unprojection);
    (c) predict_one_file(visualize_rays=True) writes the clipped rays."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils import HipEstimator3D, HipRenderer3D, view_rotations
    from mvlm_amd.utils.synthetic import write_face_like_obj
    from oracle import surface

    m = _mesh(70, 32, 8)
    e3 = HipEstimator3D(verbose=False)
    rs = np.random.RandomState(4)
    n, nl = 12, 9
    np.random.seed(6)
    poses = HipRenderer3D(n_views=n, verbose=False).generate_3d_transformations()
    lms = np.empty((nl, n, 3), np.float32)
    lms[:, :, 0] = rs.uniform(-20, 275, (nl, n))     # some maxima fall outside the silhouette -> misses
    lms[:, :, 1] = rs.uniform(-20, 275, (nl, n))
    lms[:, :, 2] = rs.rand(nl, n)
    s, e = e3.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), lms, poses)
    got, hit = e3.clip_rays_to_mesh(m, s, e)
    want, whit = surface.clip_rays_to_mesh(m.verts, m.tris, s, e)
    np.testing.assert_array_equal(hit, whit)
    assert 0.2 < hit.mean() < 1.0
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(got[~hit], e[~hit])

    # (b) frontal view: the depth plane of pixel (row, col) is the quantised z of the hit of the ray through
    # that pixel's centre (maxima convention: row - 1, col - 0.5 -> centre of pixel (row, col) is (row-0.5, col))
    r = HipRenderer3D(n_views=1, verbose=False)
    pose = np.zeros((1, 6))
    img = r.render_device(m, pose).cpu().numpy()[0]
    rows, cols = np.meshgrid(np.arange(20, 236, 9), np.arange(20, 236, 9), indexing="ij")
    px = np.stack([rows.ravel() - 0.5, cols.ravel() + 0.0, np.ones(rows.size)], axis=1).astype(np.float32)[:, None, :]
    s1, e1 = e3.estimate_landmark_lines(np.zeros((1, 256, 256, 4), np.float32), px, pose)
    ends, h1 = e3.clip_rays_to_mesh(m, s1, e1)
    depth = np.rint(img[rows.ravel(), cols.ravel(), 3] * 255).astype(int)
    zbuf = (500.0 - ends[:, 0, 2]) / 1500.0
    expect = (256 - np.trunc(255.0 * zbuf).astype(int)) % 256
    inside = h1[:, 0]
    assert inside.mean() > 0.3
    assert (np.abs(depth[inside] - expect[inside]) <= 1).mean() > 0.98   # interpolation rounding at a level boundary
    assert (depth[~inside] == 1).all()                                      # background: far plane z = 1 -> 256 - 255
    # (c) the pipeline's ray dump
    obj = write_face_like_obj(tmp_path / "face.obj", grid=40, tex_size=32, seed=1)
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", verbose=False, visualize_rays=True,
                                    screenshot_folder=tmp_path / "rays")
    np.random.seed(1)
    lm = pipe.predict_one_file(obj, landmark_indices=[0, 5, -1], view_indices=[1, 2])
    d = np.load(tmp_path / "rays" / "face_rays.npz")
    assert d["starts"].shape == (3, 2, 3) and d["hit"].shape == (3, 2) and bool(d["clipped"])
    np.testing.assert_array_equal(d["landmarks"], lm)
    np.testing.assert_array_equal(d["landmark_indices"], [0, 5, 72])
    seg = d["ends"] - d["starts"]
    assert (np.linalg.norm(seg, axis=2)[d["hit"]] < 1000.0 - 1e-6).all()   # clipped rays got shorter


@pytest.mark.parametrize("method", ["simple", "moment"])
def test_planted_peaks_take_the_inlier_branch(ctx, method):
    """SURVEY.md 8(d) "planted-peak": heatmaps that are Gaussians at the projections of known
    surface points, weak and misplaced in a third of the views.  Maxima -> rays -> quantile
    filter -> RANSAC inlier refit -> snap must (a) equal the oracle's and (b) recover the points."""
    from mvlm_amd.utils import HipEstimator3D, HipRenderer3D, view_rotations
    from oracle import cnn as ocnn
    from oracle import estimator as oest
    from oracle import surface

    n, nl = 24, 12
    m = _mesh(80, 32, 6)
    rs = np.random.RandomState(21)
    pts = m.verts[rs.choice(m.n_verts, nl, replace=False)].astype(np.float64)
    np.random.seed(2)
    poses = HipRenderer3D(n_views=n, verbose=False).generate_3d_transformations()
    rot = view_rotations(poses).reshape(-1, 3, 3)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    heat = np.empty((n, nl, 256, 256), np.float32)
    for v in range(n):
        q = pts @ rot[v].T
        col = (q[:, 0] + 150) / 300 * 256 + 0.5          # maxima report (row - 1, col - 0.5)
        row = 255 - (q[:, 1] + 150) / 300 * 256 + 1
        for lm in range(nl):
            amp = 0.6 + 0.4 * rs.rand()
            r, c = row[lm], col[lm]
            if rs.rand() < 0.33:                          # a wrong, weak detection: filtered by its score
                amp, r, c = 0.2 * rs.rand(), rs.uniform(40, 215), rs.uniform(40, 215)
            heat[v, lm] = amp * np.exp(-((yy - r) ** 2 + (xx - c) ** 2) / (2 * 3.0 ** 2)) + 1e-3 * rs.rand(256, 256)
    hd = dev(heat)
    lms = torch.empty((nl, n, 3), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.mvlm_heatmap_maxima(ctx.handle, C.c_void_p(hd.data_ptr()), n, nl, 256,
                                          {"simple": 0, "moment": 1}[method], C.c_void_p(lms.data_ptr())))
    lms = lms.cpu().numpy()
    np.testing.assert_array_equal(lms, ocnn.maxima_from_heatmaps(heat, method))

    e3 = HipEstimator3D(verbose=False)
    s, e = e3.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), lms, poses)
    np.random.seed(9)
    out, err = e3.estimate_landmarks_from_lines(lms, s, e)
    snapped = e3.project_landmarks_to_surface(m, out)
    os_, oe = oest.estimate_landmark_lines(256, lms, poses)
    np.random.seed(9)
    want, werr = oest.estimate_landmarks_from_lines(lms, os_, oe)
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-8)
    assert abs(err - werr) <= 1e-9 * max(1.0, abs(werr))
    np.testing.assert_allclose(snapped, surface.project_landmarks_to_surface(m.verts, m.tris, want), rtol=0, atol=1e-8)
    assert err < 10.0                                     # every landmark took the inlier branch (else 1e8 / NL)
    # integer argmax = half a pixel (0.6 model units) per view; the sub-pixel centroid does better
    assert np.linalg.norm(snapped - pts, axis=1).max() < (0.5 if method == "simple" else 0.05)


def test_predict_files_equals_the_sequential_loop(tmp_path):
    """Folder mode (reader thread ahead of the GPU) returns what looping predict_one_file returns."""
    import shutil

    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    files = [write_face_like_obj(tmp_path / f"s{i}.obj", grid=30 + 7 * i, tex_size=64, seed=i) for i in range(3)]
    files.insert(2, tmp_path / "missing.obj")
    shutil.copy(files[0], tmp_path / "notex.obj")  # same geometry, no .jpg next to it -> white mesh
    files.append(tmp_path / "notex.obj")
    pipe = pipeline.create_pipeline("dtu3d", n_views=16, weights="synthetic:2", verbose=False)
    np.random.seed(4)
    want = [pipe.predict_one_file(f) for f in files]
    np.random.seed(4)
    got = list(pipe.predict_files(files))
    assert [f for f, _ in got] == files
    for (f, g), w in zip(got, want):
        if w is None:
            assert g is None and f.name == "missing.obj"
        else:
            np.testing.assert_array_equal(g, w)
    (tmp_path / "bad.obj").write_text("# nothing\n")
    with pytest.raises(ValueError, match="does not contain any points"):
        list(pipe.predict_files([files[0], tmp_path / "bad.obj"]))


def test_predict_files_with_reader_pool_and_recycled_buffers(tmp_path):
    """Several reader threads upload scans of changing size ahead of the GPU: staging slots grow and alternate, device
    buffers of dropped meshes come back from the pool (oldest first, behind their 'freed' event), and every scan must
    still give exactly the sequential loop's landmarks - twice over, so the second pass runs entirely on recycled buffers."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    grids = [28, 61, 33, 75, 30, 52, 90, 31]
    files = [write_face_like_obj(tmp_path / f"s{i}.obj", grid=g, tex_size=32 * (1 + i % 3), seed=i) for i, g in enumerate(grids)]
    pipe = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:6", image_mode="RGB+depth", verbose=False)
    np.random.seed(9)
    want = [pipe.predict_one_file(f) for f in files + files]
    np.random.seed(9)
    got = [lm for _, lm in pipe.predict_files(files + files, readers=3)]
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("n_views", [8, 12])
def test_batched_scans_equal_the_sequential_loop(tmp_path, n_views, monkeypatch):
    """predict_files(batch_scans=3): three scans share one network pass and every scan still gets the sequential loop's
    landmarks - fixed 8-view table and RNG-drawn poses (12 views: poses and RANSAC draws interleave in the global RNG
    stream), a missing file in the middle, a last group that is not full.  A larger device batch may be served by other
    kernel variants (another fp32 summation order), so a near-tied argmax of these random-weight heatmaps may move: a
    landmark whose maxima did not move is bit-identical, and at least 95 % of them must be; the RNG must stand exactly
    where the loop leaves it.  Then the same with one scan's speculative draws made for WRONG survivor counts: its draws
    are repeated from the saved RNG state and the scans after it fall back to the loop."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    files = [write_face_like_obj(tmp_path / f"s{i}.obj", grid=30 + 5 * i, tex_size=64, seed=i) for i in range(7)]
    files.insert(4, tmp_path / "missing.obj")
    pipe = pipeline.create_pipeline("bu3dfe", n_views=n_views, weights="synthetic:8", image_mode="RGB+depth", verbose=False)
    np.random.seed(21)
    want = [pipe.predict_one_file(f) for f in files]
    state_after = np.random.get_state()[1].copy()
    np.random.seed(21)
    got = list(pipe.predict_files(files, batch_scans=3))
    assert [f for f, _ in got] == files

    def same_scans(res):
        for (f, g), w in zip(res, want):
            if w is None:
                assert g is None
            else:
                identical = np.all(g == w, axis=1)
                assert identical.mean() >= 0.95, (f.name, identical.mean())
                assert np.abs(g - w).max() < 60.0

    same_scans(got)
    np.testing.assert_array_equal(np.random.get_state()[1], state_after)   # the RNG stands where the loop leaves it

    e3 = pipe.estimator_3d
    real_plan, calls = e3.plan_draws, [0]

    def tampered(nl, n, draws_fn=None, slot=0):
        plan = real_plan(nl, n, draws_fn, slot=slot)
        calls[0] += 1
        if calls[0] == 2 and plan["expected"] is not None:   # second scan of the first group
            plan["expected"] = plan["expected"].copy()
            plan["expected"][0] += 1
        return plan

    monkeypatch.setattr(e3, "plan_draws", tampered)
    np.random.seed(21)
    again = list(pipe.predict_files(files, batch_scans=3))
    assert calls[0] >= 7
    same_scans(again)
    np.testing.assert_array_equal(np.random.get_state()[1], state_after)


def test_geometry_shading_bit_exact_and_config_driven():
    """The build-defined geometry plane: HIP == CPU restatement, and a geometry+depth config selects it."""
    from mvlm_amd import config
    from mvlm_amd.utils import HipRenderer3D
    from oracle import raster

    m = _mesh(60, 32, 3)
    r = HipRenderer3D(n_views=8, verbose=False, shading="geometry")
    poses = r.generate_3d_transformations()
    got = r.render_device(m, poses).cpu().numpy()
    want = raster.multiview_render(m.verts, m.tris, m.uvs, m.texture, poses, shading="geometry")
    np.testing.assert_array_equal(got, want)
    fg = want[..., 3] != np.float32(1 / 255)
    assert np.array_equal(want[..., 0], want[..., 1]) and want[fg][:, 0].std() > 0.01  # grey, and really shaded
    pipe = config.load_config(config.default_config("DTU3D", "geometry+depth", n_views=8)).build_pipeline(
        weights="synthetic:2", verbose=False)
    assert pipe.renderer_3d.shading == "geometry" and pipe.predictor_2d.in_channels == 2


# --------------------------------------------------------------------------------------
# edge cases of the consensus stage against the oracle (same RNG stream on both sides)
@pytest.mark.parametrize("case", ["ties", "nan_scores", "one_view", "two_views", "absolute_none"])
def test_consensus_edge_cases_match_oracle(case):
    from mvlm_amd.utils import HipEstimator3D
    from oracle import estimator as oest
    from oracle import poses as oposes

    rs = np.random.RandomState(17)
    n = {"one_view": 1, "two_views": 2}.get(case, 16)
    np.random.seed(5)
    poses = oposes.generate_3d_transformations(n) if n != 8 else oposes.generate_3d_transformations(8)
    nl = 6
    lms = np.empty((nl, n, 3), np.float32)
    lms[:, :, 0] = rs.uniform(20, 230, (nl, n))
    lms[:, :, 1] = rs.uniform(20, 230, (nl, n))
    lms[:, :, 2] = rs.rand(nl, n)
    mode, thr = "quantile", 0.5
    if case == "ties":
        lms[:, :, 2] = 0.25          # every score equal -> nothing is strictly above the median
        lms[1, :5, 2] = 0.75         # a few above
    elif case == "nan_scores":
        lms[0, 3, 2] = np.nan        # np.quantile -> nan -> no line survives
    elif case == "absolute_none":
        mode, thr = "absolute", 2.0  # nothing passes
    e3 = HipEstimator3D(mode=mode, threshold_absolute=thr, verbose=False)
    s, e = e3.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), lms, poses)
    os_, oe = oest.estimate_landmark_lines(256, lms, poses)
    np.testing.assert_allclose(s, os_, rtol=0, atol=1e-9)
    np.random.seed(9)
    got, gerr = e3.estimate_landmarks_from_lines(lms, s, e)
    np.random.seed(9)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        want, werr = oest.estimate_landmarks_from_lines(lms, os_, oe, mode, 0.5, thr)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)
    assert abs(gerr - werr) <= 1e-9 * max(1.0, abs(werr))


def test_invalid_views_are_dropped_like_the_reference(tmp_path):
    """A predictor that reports views without a detection (mediapipepredictor.py:38-41): the
    pipeline slices them away (general_pipeline.py:93-95) before rays and consensus."""
    from mvlm_amd import pipeline
    from mvlm_amd.prediction import PrecomputedPredictor
    from mvlm_amd.utils.synthetic import write_face_like_obj
    from oracle import estimator as oest

    obj = write_face_like_obj(tmp_path / "face.obj", grid=30, tex_size=32, seed=1)
    rs = np.random.RandomState(3)
    nl, n = 20, 8
    lms = np.empty((nl, n, 3), np.float32)
    lms[:, :, 0] = rs.uniform(60, 200, (nl, n))
    lms[:, :, 1] = rs.uniform(60, 200, (nl, n))
    lms[:, :, 2] = rs.rand(nl, n)
    valid = np.array([1, 1, 0, 1, 1, 1, 0, 1], bool)
    lms[:, ~valid, :] = np.nan
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:1", verbose=False)
    pipe.predictor_2d = PrecomputedPredictor(nl, lambda images: (lms, valid))
    assert pipe.get_lm_count() == nl
    np.random.seed(2)
    got = pipe.predict_one_file(obj)
    poses = pipe.renderer_3d.generate_3d_transformations()[valid]
    s, e = oest.estimate_landmark_lines(256, lms[:, valid], poses)
    np.random.seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        raw, _ = oest.estimate_landmarks_from_lines(lms[:, valid], s, e)
    from mvlm_amd.utils.mesh_io import load_obj
    from oracle import surface

    m = load_obj(obj)
    want = surface.project_landmarks_to_surface(m.verts, m.tris, raw)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)


def test_end_to_end_64_views_config1_against_oracle():
    """BASELINE configs[1] size: DTU3D, 64 views, ~100k-triangle mesh (RGB+depth net of the live reference
    pipeline).  Landmarks agree to 1e-3 model units wherever no view's argmax sits on a near-tie."""
    got, gerr, want, werr, inter, gmax, pipe, mesh, poses = _e2e(64, 224, "dtu3d", 11)
    diff_views = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)      # [NL, N]
    assert diff_views.mean() < 0.01                                                  # < 1 % of the 4672 planes
    same = ~diff_views.any(axis=1)
    assert same.mean() > 0.6
    assert np.abs(got[same] - want[same]).max() < 1e-3
    assert np.abs(got - want).max() < 2.5   # a flipped ray moves a landmark by a fraction of a pixel (1.17 units)
