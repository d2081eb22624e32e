"""Round-2 GPU parity additions: the renderer against the reference estimator's rays and the analytic
scenes, the end-to-end matrix over the BASELINE configurations' networks (C = 1, 2, 3, 4 through
``pipeline_from_config``), the planted-peak case through all convolutions, rank-deficient LSQ goldens on
the GPU solver, NaN landmarks in the snap, and the conv tiles only big device batches reach.
Everything calls through the C ABI (libmvlm_hip.so).  Run with -m gpu."""
import contextlib
import ctypes as C
import io

import numpy as np
import pytest
import torch

import planted
import test_oracle_pinning as pin

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hip_render(verts, tris, uvs, tex, poses, shading="texture"):
    from mvlm_amd.utils import HipRenderer3D, Mesh

    r = HipRenderer3D(n_views=int(poses.shape[0]), verbose=False, shading=shading)
    m = Mesh(np.asarray(verts, np.float32), np.asarray(tris, np.int32), uvs, tex)
    out = r.render_device(m, np.asarray(poses)).cpu().numpy()
    r.check()
    return out


# ---- renderer: pinned scenes ---------------------------------------------------------------------
def test_hip_renderer_on_reference_estimator_rays(golden):
    """Markers spanned by the REFERENCE estimator's rays (tests/golden/marker_rays.npz) are drawn by the HIP
    rasteriser at exactly their pixels, for 12 random rotations and the fixed 8-view table: rotation order,
    vertical flip and pixel centres of the product renderer are pinned to reference code."""
    from oracle import raster

    g = golden("marker_rays.npz")
    poses = g["poses"]
    for view in range(poses.shape[0]):
        verts, tris = pin.marker_scene(g, view)
        img = hip_render(verts, tris, None, None, poses[view:view + 1])[0]
        pin.check_marker_view(img, g, view)
        np.testing.assert_array_equal(img, raster.multiview_render(verts, tris, None, None, poses[view:view + 1])[0])


def test_hip_renderer_analytic_scenes():
    """The hand-computed scenes of tests/test_oracle_pinning.py through the HIP rasteriser."""
    for z0 in (-103.0, 10.0, 143.0):
        x0, x1, y0, y1 = -50.0, 61.0, -20.0, 33.0
        verts, tris, _ = pin.quad(x0, x1, y0, y1, z0)
        img = hip_render(verts, tris, None, None, pin.IDENTITY)[0]
        want = np.tile(pin.BG, (256, 256, 1))
        rows = [255 - j for j in pin.covered(*pin.window([y0, y1]))]
        want[np.ix_(rows, pin.covered(*pin.window([x0, x1])))] = [1.0, 1.0, 1.0, pin.depth_byte(z0) / 255.0]
        np.testing.assert_array_equal(img, want.astype(np.float32))
    tex = np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 0]]], np.uint8)
    verts, tris, uvs = pin.quad(-60, 60, -60, 60, 0.0, [[0, 0], [2, 0], [2, 2], [0, 2]])
    rgb = np.rint(hip_render(verts, tris, uvs, tex, pin.IDENTITY)[0][..., :3] * 255).astype(int)
    assert rgb[170, 85].tolist() == [0, 0, 255] and rgb[170, 110].tolist() == [255, 255, 0]
    assert rgb[170, 135].tolist() == [0, 0, 255] and rgb[145, 85].tolist() == [255, 0, 0]


def test_depth_plane_is_the_first_ray_hit_for_rotated_poses():
    """mvlm_clip_rays_to_mesh along the rays of ROTATED views ends where each view's depth plane says the surface
    is: ties renderer (pose, flip, depth quantisation) and estimator rays together away from the identity pose."""
    from mvlm_amd.utils import HipEstimator3D, HipRenderer3D, view_rotations
    from mvlm_amd.utils.synthetic import face_like_mesh

    m = face_like_mesh(70, 32, 8)
    poses = np.array([[25.0, -40.0, 15.0, 0, 0, 0], [-35.0, 62.0, -18.0, 0, 0, 0], [30.0, 15.0, 0.0, 0, 0, 0]])
    r = HipRenderer3D(n_views=3, verbose=False)
    e3 = HipEstimator3D(verbose=False)
    imgs = r.render_device(m, poses).cpu().numpy()
    rot = view_rotations(poses).reshape(-1, 3, 3)
    rows, cols = np.meshgrid(np.arange(20, 236, 7), np.arange(20, 236, 7), indexing="ij")
    # centre of pixel (row, col) in the maxima convention (row - 1, col - 0.5): (row - 0.5, col)
    px = np.stack([rows.ravel() - 0.5, cols.ravel() + 0.0, np.ones(rows.size)], axis=1).astype(np.float32)
    lms = np.repeat(px[:, None, :], 3, axis=1)
    s, e = e3.estimate_landmark_lines(np.zeros((3, 256, 256, 4), np.float32), lms, poses)
    ends, hit = e3.clip_rays_to_mesh(m, s, e)
    for v in range(3):
        depth = np.rint(imgs[v, rows.ravel(), cols.ravel(), 3] * 255).astype(int)
        z_view = ends[:, v] @ rot[v].T[:, 2]                                  # (M p).z
        expect = np.array([pin.depth_byte(z) for z in z_view])
        inside = hit[:, v]
        assert inside.mean() > 0.15
        assert (np.abs(depth[inside] - expect[inside]) <= 1).mean() > 0.95     # silhouette pixels: centre vs. sub-pixel ray
        assert (depth[~inside] == 1).mean() > 0.95


# ---- end-to-end matrix ---------------------------------------------------------------------------
def _e2e_config(dataset, mode, n_views, grid, seed):
    """One BASELINE configuration through the pipeline its config file builds (``default_config`` = the file's
    inference keys, pre-align block included) against the oracle.  A config with an active pre-align gets the raw
    scan that block is written for (small, turned, off-centre); the oracle runs the reference's own transform
    sequence (oracle/prealign.py) on the raw points."""
    from mvlm_amd import arch, config, weights
    from mvlm_amd.pipeline import pipeline_from_config
    from mvlm_amd.utils.prealign import aligned, is_active
    from mvlm_amd.utils.synthetic import face_like_mesh, unaligned_copy
    from oracle import pipeline as opipe
    from oracle import prealign as opre

    cfg = config.default_config(dataset, mode, n_views=n_views)
    pipe = pipeline_from_config(cfg, weights=f"synthetic:{seed}", verbose=False)
    nl, c = pipe.get_lm_count(), arch.IMAGE_CHANNELS[mode]
    raw = face_like_mesh(grid, 128, seed)
    block = cfg["pre-align"]
    if is_active(block):
        raw = unaligned_copy(raw, block)
    mesh = aligned(raw, pipe.pre_align)
    assert (mesh is raw) == (not is_active(block))
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    np.random.seed(1)
    got, gerr = pipe.predict_mesh_device(mesh, poses)
    got = pipe._to_original(mesh, got)
    sd = weights.synthetic_state_dict(nl, c, seed=seed)
    overts, t = opre.pre_transformation(raw.verts, block) if is_active(block) else (raw.verts, None)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        want, werr, inter = opipe.predict_mesh(overts, raw.tris, raw.uvs, raw.texture, poses, sd,
                                               arch.CHANNEL_SELECT[mode],
                                               shading="geometry" if "geometry" in mode else "texture")
    if t is not None:
        want = opre.landmarks_to_original_space(want, t)
    images = pipe.renderer_3d.render_device(mesh, poses)
    np.testing.assert_array_equal(images.cpu().numpy(), inter["images"])
    gmax = pipe.predictor_2d.predict_device(images).cpu().numpy()
    unit = 1.0 / float(block["scale"])   # one model unit of the rendered (aligned) space, in the file's coordinates
    return got, gerr, want, werr, inter, gmax, unit


@pytest.mark.parametrize("dataset,mode,n_views,grid", [
    ("BU_3DFE", "depth", 8, 51),              # BASELINE configs[0]: C = 1, the fixed 8-view table, pre-align (centre + scale 20)
    ("DTU3D", "RGB", 64, 224),                # configs[1]: C = 3, 64 views, ~100k triangles
    ("DTU3D", "geometry+depth", 12, 60),      # configs[3]'s per-GPU shard: C = 2, 12 views, geometry shading
    ("BU_3DFE", "RGB+depth", 16, 60),         # configs[2]'s network (C = 4, 84 landmarks)
    ("BU_3DFE", "RGB+depth", 96, 224),        # configs[2] at full size: the configuration bench.py's value is quoted on
    ("DTU3D", "geometry+depth", 96, 224),     # configs[3] at full size on one GPU (its 8-GPU form shards these 96 views)
])
def test_end_to_end_config_matrix_against_oracle(dataset, mode, n_views, grid):
    got, gerr, want, werr, inter, gmax, unit = _e2e_config(dataset, mode, n_views, grid, 13)
    diff_views = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)      # [NL, N]
    # argmax near-ties only (the summation order differs from oneDNN's); measured: 0 of 8 064 planes at the
    # bench size (profiles/r02_parity_stats.txt) - allowed: 0.2 %
    assert diff_views.mean() <= 0.002
    same = ~diff_views.any(axis=1)
    assert same.mean() > 0.9
    assert np.abs(got[same] - want[same]).max() < 1e-3 * unit   # BASELINE north_star: 1e-3 model units
    assert np.abs(got - want).max() < 2.5 * unit
    if same.all():
        assert abs(gerr - werr) <= 1e-6 * max(1.0, abs(werr))


def test_planted_peaks_through_the_network():
    """SURVEY.md 8(d) planted peak THROUGH conv11: the hand-made detector (tests/planted.py) makes every landmark
    channel peak at a known surface point, so render -> 138 convs -> fused argmax -> rays -> quantile filter ->
    RANSAC *inlier* refit -> snap runs end to end on the GPU and is compared with the oracle and the truth."""
    from mvlm_amd import arch, config
    from mvlm_amd.pipeline import pipeline_from_config
    from oracle import pipeline as opipe
    from test_planted_cpu import planted_scene

    mesh, pts, sd, poses = planted_scene(n_views=16)
    pipe = pipeline_from_config(config.default_config("DTU3D", "RGB", n_views=16), weights=sd, verbose=False)
    np.random.seed(1)
    got, gerr = pipe.predict_mesh_device(mesh, poses)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        want, werr, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses, sd,
                                               arch.CHANNEL_SELECT["RGB"])
    assert werr < 10.0 and gerr < 10.0               # inlier branch for every landmark (a fallback adds 1e8 / NL)
    gmax = pipe.predictor_2d.predict_device(pipe.renderer_3d.render_device(mesh, poses)).cpu().numpy()
    same_px = np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)
    assert same_px.mean() > 0.99
    same = same_px.all(axis=1)
    assert same.mean() > 0.8
    assert np.abs(got[same] - want[same]).max() < 1e-3
    assert abs(gerr - werr) < 1e-6 * max(1.0, werr) or not same.all()
    d = np.linalg.norm(got - pts, axis=1)
    assert d.max() < 6.0 and np.median(d) < 4.0      # the planted points are found (offset: see test_planted_cpu)


# ---- consensus: the LSQ goldens (rank-deficient systems) on the GPU solver --------------------------
@pytest.mark.parametrize("tag", ["k0", "k1", "k2", "parallel", "k6"])
def test_lsq_goldens_on_the_gpu_solver(golden, tag):
    """compute_intersection_between_lines (utils3d.py:99-124, pinv with numpy's cutoff) as reproduced by the
    GPU's Jacobi pseudo-inverse: 0 / 1 / 2 lines (k < 3: plain LSQ, estimator3d.py:174-176), three PARALLEL
    lines (rank 2) and a generic bundle (both through the one-shot RANSAC, all lines inliers -> refit on all)."""
    from mvlm_amd.utils import HipEstimator3D

    g = golden("estimator.npz")
    pa, pb, want = g[f"lsq_{tag}_pa"], g[f"lsq_{tag}_pb"], g[f"lsq_{tag}_p"]
    if tag == "parallel":
        # the golden lines are 10 and 7 units apart: with the inlier rule dist^2 < 100 (estimator3d.py:97) the
        # one-shot draw would decide which of them are refitted.  The LSQ point is equivariant under uniform
        # scaling, so the same rank-2 system at half size (5 and 3.5 apart: every line an inlier for any draw)
        # must give half the reference's point.
        pa, pb, want = 0.5 * pa, 0.5 * pb, 0.5 * want
    k = pa.shape[0]
    n = k + 2                                        # two extra views whose score fails the absolute threshold
    s = np.zeros((1, n, 3))
    e = np.ones((1, n, 3))
    s[0, 1:1 + k], e[0, 1:1 + k] = pa, pb
    lms = np.zeros((1, n, 3), np.float32)
    lms[0, 1:1 + k, 2] = 0.9
    lms[0, [0, n - 1], 2] = 0.1
    e3 = HipEstimator3D(mode="absolute", threshold_absolute=0.5, verbose=False)
    np.random.seed(0)
    out, err = e3.estimate_landmarks_from_lines(lms, s, e)
    np.testing.assert_allclose(out[0], want, rtol=0, atol=1e-8)
    if k >= 3:
        assert err < 1e8                             # inlier branch


def test_snap_passes_non_finite_landmarks_through():
    """A NaN landmark (non-finite heatmaps / weights upstream) has no nearest triangle: it is returned as it
    came (the CPU oracle's arithmetic gives NaN as well), never uninitialised memory; its neighbours are unaffected."""
    from mvlm_amd.utils import HipEstimator3D
    from mvlm_amd.utils.synthetic import face_like_mesh
    from oracle import surface

    m = face_like_mesh(40, 16, 2)
    pts = np.array([[0.0, 0.0, 80.0], [np.nan, 1.0, 2.0], [10.0, -20.0, 5.0], [np.inf, 0.0, 0.0]])
    e3 = HipEstimator3D(verbose=False)
    for _ in range(2):                               # twice: a stale value from the first call must not show up
        got = e3.project_landmarks_to_surface(m, pts)
    want = surface.project_landmarks_to_surface(m.verts, m.tris, pts[[0, 2]])
    np.testing.assert_allclose(got[[0, 2]], want, rtol=0, atol=1e-9)
    assert np.isnan(got[1]).any() and not np.isfinite(got[3]).all()


# ---- conv tiles that only device batches > 128 views reach -------------------------------------------
@pytest.mark.parametrize("cin,cout,size,batch", [(256, 128, 8, 129), (64, 64, 8, 130), (256, 128, 4, 513), (128, 64, 16, 33)])
def test_conv_tiles_of_large_device_batches(cin, cout, size, batch):
    """8x8 / 4x4 hourglass levels leave the split-K tiles above 8192 pixels per level (device_batch > 128):
    the image-folding tiles conv3x3_c32_t8x8x2 / t4x4x8 (and t8x16 at 16x16) against torch float64."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    rs = np.random.RandomState(cin + size + batch)
    x = rs.standard_normal((batch, cin, size, size)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32))
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32)
    xd, rd = dev(x), dev(res)
    yd = torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, 3, None,
                                  p(pre[0]), p(pre[1]), None, None, C.c_void_p(rd.data_ptr()), 0, C.c_void_p(yd.data_ptr())))
    t = torch.relu(torch.from_numpy(x).double() * torch.from_numpy(pre[0]).double()[None, :, None, None]
                   + torch.from_numpy(pre[1]).double()[None, :, None, None])
    want = (torch.nn.functional.conv2d(t, torch.from_numpy(w).double(), None, 1, 1) + torch.from_numpy(res).double()).numpy()
    assert np.abs(yd.cpu().numpy() - want).max() < 5e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("variant,name,cin,cout,size,batch", [
    (27, "conv3x3_c32k8_t8x16", 128, 64, 64, 2), (27, "conv3x3_c32k8_t8x16", 256, 128, 16, 5), (27, "conv3x3_c32k8_t8x16", 64, 32, 128, 1),
    (28, "conv3x3_c64k8_t8x16", 256, 128, 64, 1), (28, "conv3x3_c64k8_t8x16", 64, 64, 32, 3), (28, "conv3x3_c64k8_t8x16", 128, 64, 16, 2),
])
def test_round3_tiles_match_torch(variant, name, cin, cout, size, batch):
    """Round 3: the 8x16-pixel tiles with 8-channel chunks (half the LDS: more workgroups per CU at small batches), forced
    onto residual-block shaped layers (pre-BN, residual add) against torch float64."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    rs = np.random.RandomState(cin + size + batch + variant)
    x = rs.standard_normal((batch, cin, size, size)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32))
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32)
    xd, rd = dev(x), dev(res)
    yd = torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert ctx.lib.mvlm_conv_variant_name(variant).decode() == name
    ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, variant))
    try:
        ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, 3, None,
                                      p(pre[0]), p(pre[1]), None, None, C.c_void_p(rd.data_ptr()), 0, C.c_void_p(yd.data_ptr())))
    finally:
        ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, -1))
    t = torch.relu(torch.from_numpy(x).double() * torch.from_numpy(pre[0]).double()[None, :, None, None]
                   + torch.from_numpy(pre[1]).double()[None, :, None, None])
    want = (torch.nn.functional.conv2d(t, torch.from_numpy(w).double(), None, 1, 1) + torch.from_numpy(res).double()).numpy()
    assert np.abs(yd.cpu().numpy() - want).max() < 5e-6 * max(1.0, np.abs(want).max())


def test_visualize_image_stack_writes_the_views(tmp_path):
    """--visualize-method (general_pipeline.py:133-146): one PNG per view, RGB planes * 255 as uint8."""
    from PIL import Image

    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=30, tex_size=32, seed=1)
    out = tmp_path / "png"
    out.mkdir()
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:1", verbose=False, render_image_stack=True,
                                    render_image_folder=out)
    np.random.seed(1)
    assert pipe.predict_one_file(obj) is not None
    stack, _, _ = pipe.renderer_3d.multiview_render(obj)
    files = sorted(out.glob("face_*.png"))
    assert [f.name for f in files] == [f"face_{i:02d}.png" for i in range(8)]
    for i, f in enumerate(files):
        np.testing.assert_array_equal(np.asarray(Image.open(f)), np.uint8(stack[i, :, :, 0:3] * 255))
    pipe.render_image_folder = tmp_path / "does_not_exist"
    with pytest.raises(ValueError, match="does not exist"):
        pipe.predict_one_file(obj)


# ---- execution modes of the network: one stream / two streams / captured graph ---------------------------
@pytest.mark.parametrize("n_views", [8, 12, 40])
def test_execution_modes_give_identical_results(n_views):
    """The forward pass launched kernel by kernel on one stream, with the lower hourglass pyramid on a second
    stream (batches <= 32 views), and replayed from a captured hipGraph must agree bit for bit - maxima and
    full heatmaps - and the replay path must really have been taken."""
    from conftest import seeded_images
    from mvlm_amd.prediction import DTU3DPredictor

    imgs = dev(seeded_images(5 + n_views, n_views))
    pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:4", verbose=False)
    out = torch.empty((73, n_views, 3), dtype=torch.float32, device="cuda")
    # (pairing 0: one launch per convolution in every mode - the two-stream experiment does not pair blocks, and a pair may
    #  run another kernel variant than the single launch; paired execution has its own test in test_gpu_round4.py)
    pred.set_execution(graphs=False, concurrency=False, pairing=0)
    want = pred.predict_device(imgs, out=out).clone()
    want_heat = pred.heatmaps_device(imgs[:2]).clone()
    pred.set_execution(graphs=False, concurrency=True)
    np.testing.assert_array_equal(pred.predict_device(imgs, out=out).cpu().numpy(), want.cpu().numpy())
    assert torch.equal(pred.heatmaps_device(imgs[:2]), want_heat)
    pred.set_execution(graphs=True, concurrency=True)
    before = pred.execution_stats()
    for i in range(4):  # 1st: eager, 2nd: capture + launch, then replays
        out.zero_()
        got = pred.predict_device(imgs, out=out)
        assert torch.equal(got, want), f"pass {i}"
    after = pred.execution_stats()
    assert after["graph_failures"] == before["graph_failures"]
    assert after["graph_captures"] == before["graph_captures"] + 1
    assert after["graph_replays"] == before["graph_replays"] + 3
    # new input content through the same buffers: the replay reads what is in them now
    imgs2 = dev(seeded_images(99, n_views))
    pred.set_execution(graphs=False, concurrency=False)
    want2 = pred.predict_device(imgs2, out=torch.empty_like(out)).clone()
    pred.set_execution(graphs=True, concurrency=True)
    imgs.copy_(imgs2)
    for _ in range(3):
        got = pred.predict_device(imgs, out=out)
    assert torch.equal(got, want2)
    assert pred.execution_stats()["graph_replays"] > after["graph_replays"]


def test_pipeline_steps_replay_the_graph_and_match_eager(tmp_path):
    """predict_mesh_device in a loop (what bench.py times): stable buffers -> graph replay; landmarks equal the eager ones."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import face_like_mesh

    mesh = face_like_mesh(60, 64, 3)
    pipe = pipeline.create_pipeline("bu3dfe", n_views=12, weights="synthetic:6", verbose=False)
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    pipe.predictor_2d.set_execution(graphs=False, concurrency=False)
    np.random.seed(1)
    want, werr = pipe.predict_mesh_device(mesh, poses)
    pipe.predictor_2d.set_execution(graphs=True, concurrency=True)
    for _ in range(4):
        np.random.seed(1)
        got, gerr = pipe.predict_mesh_device(mesh, poses)
        np.testing.assert_array_equal(got, want)
        assert gerr == werr
    st = pipe.predictor_2d.execution_stats()
    assert st["graph_replays"] >= 3 and st["graph_failures"] == 0


# ---- np.argmax order incl. NaN and signed zeros (paulsenpredictor.py:123) -------------------------------------
def test_heatmap_maxima_nan_and_signed_zero_like_numpy():
    from mvlm_amd import _lib
    from oracle import cnn as ocnn

    ctx = _lib.get_context(0)
    rs = np.random.RandomState(3)
    hm = rs.standard_normal((1, 6, 64, 64)).astype(np.float32)
    hm[0, 0, 5, 7] = np.nan            # np.argmax: the first NaN is the maximum
    hm[0, 0, 40, 3] = np.nan
    hm[0, 1, 9, 9] = np.inf            # a NaN later in the plane still wins over +inf
    hm[0, 1, 50, 50] = np.nan
    hm[0, 2] = 0.0
    hm[0, 2, 0, 0] = -0.0              # -0 == +0: the first element wins, not the first +0
    hm[0, 3] = -np.inf                 # constant -inf plane: index 0
    hm[0, 4, 63, 63] = 1e30            # ordinary maximum in the last pixel
    for method in (0, 1):
        out = torch.empty((6, 1, 3), dtype=torch.float32, device="cuda")
        hd = dev(hm)
        ctx.check(ctx.lib.mvlm_heatmap_maxima(ctx.handle, C.c_void_p(hd.data_ptr()), 1, 6, 64, method, C.c_void_p(out.data_ptr())))
        want = ocnn.maxima_from_heatmaps(hm, "simple" if method == 0 else "moment")
        np.testing.assert_array_equal(out.cpu().numpy(), want)      # assert_array_equal treats NaN == NaN
    assert np.isnan(want[0, 0, 2]) and want[0, 0, 0] == 4 and want[0, 0, 1] == 6.5 and want[1, 0, 0] == 49


@pytest.mark.parametrize("family,nl,poisoned", [("dtu3d", 73, (3, 70)), ("bu3dfe", 84, (3, 70, 81, 83))])
def test_fused_argmax_nan_plane_like_numpy(family, nl, poisoned):
    """A NaN bias in conv11 makes one landmark's whole heatmap NaN: np.argmax returns pixel 0 and the value NaN
    (paulsenpredictor.py:123-127); the fused conv11 + argmax kernels must do the same and leave the others alone.
    73 landmarks: a channel of the 32-row tiles and one of the 16-row strip; 84 landmarks (conv2x2_c84_t8x32, round 3):
    also two of the 4-row strip, whose partial maximum is a 64-lane reduction."""
    from conftest import seeded_images
    from mvlm_amd import weights
    from mvlm_amd.prediction import BU3DFEPredictor, DTU3DPredictor

    cls = DTU3DPredictor if family == "dtu3d" else BU3DFEPredictor
    sd = weights.synthetic_state_dict(nl, 3, seed=8)
    imgs = dev(seeded_images(31, 3))
    clean = cls(image_mode="RGB", weights=sd, verbose=False).predict_device(imgs).cpu().numpy()
    sd = dict(sd)
    sd["conv11.bias"] = sd["conv11.bias"].copy()
    sd["conv11.bias"][list(poisoned)] = np.nan
    got = cls(image_mode="RGB", weights=sd, verbose=False).predict_device(imgs).cpu().numpy()
    for lm in poisoned:
        np.testing.assert_array_equal(got[lm, :, :2], np.tile(np.float32([-1.0, -0.5]), (3, 1)))
        assert np.isnan(got[lm, :, 2]).all()
    keep = [i for i in range(nl) if i not in poisoned]
    np.testing.assert_array_equal(got[keep], clean[keep])


def test_fused_argmax_of_the_four_row_strip_against_the_heatmaps():
    """conv11 of the 84-landmark network runs on 84 rows (64 + 16 + 4): the fused maxima of EVERY landmark - the last
    four come out of the 4-row strip's lane reduction - equal the argmax of the materialised heat maps (whose parity
    stores go through the same tile), first maximum in row-major order, value bit for bit."""
    from conftest import seeded_images
    from mvlm_amd.prediction import BU3DFEPredictor

    pred = BU3DFEPredictor(image_mode="RGB+depth", weights="synthetic:21", verbose=False)
    imgs = dev(seeded_images(77, 5))
    maxima = pred.predict_device(imgs).cpu().numpy()                     # [84, 5, 3]
    heat = pred.heatmaps_device(imgs).cpu().numpy()                      # [5, 84, 256, 256]
    flat = heat.reshape(5, 84, -1)
    idx = flat.argmax(axis=2)                                            # np.argmax: first maximum
    np.testing.assert_array_equal(maxima[:, :, 0].T, (idx // 256 - 1).astype(np.float32))
    np.testing.assert_array_equal(maxima[:, :, 1].T, (idx % 256 - 0.5).astype(np.float32))
    np.testing.assert_array_equal(maxima[:, :, 2].T, np.take_along_axis(flat, idx[:, :, None], axis=2)[:, :, 0])


# ---- every split-K tile variant (32-, 16- and 8-channel chunks) against torch -------------------------------
@pytest.mark.parametrize("variant,size", [(12, 16), (13, 8), (14, 4), (15, 32), (18, 16), (19, 8), (20, 4), (21, 32),
                                          (22, 16), (23, 8), (24, 4), (25, 32), (19, 32), (23, 16), (18, 32), (22, 64),
                                          # + 256 log2(parts): the input channels divided over 2 / 4 / 8 workgroups per tile
                                          (256 + 13, 8), (512 + 14, 4), (256 + 19, 8), (512 + 23, 8), (768 + 22, 16),
                                          (768 + 24, 4), (512 + 12, 16), (256 + 21, 32)])
def test_split_k_variants_match_torch(variant, size):
    """The tuned table may pick any split-K tile for a small level; each one forced onto a residual-block shaped
    layer (pre-BN, residual add, ragged batch) against torch float64.  The variants that divide K over workgroups run
    four times on changing inputs: the arrival counters must be back at zero after a launch and no part may come from
    the previous launch's workspace contents."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    cin, cout, batch = 128, 64, 5
    rs = np.random.RandomState(variant * 31 + size)
    x = rs.standard_normal((batch, cin, size, size)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32))
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32)
    xd, rd = dev(x), dev(res)
    yd = torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    name = ctx.lib.mvlm_conv_variant_name(variant).decode()
    assert name.startswith("conv3x3_sk"), name
    ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, variant))
    wd = torch.from_numpy(w).double()

    def want(xin):
        t = torch.relu(torch.from_numpy(xin).double() * torch.from_numpy(pre[0]).double()[None, :, None, None]
                       + torch.from_numpy(pre[1]).double()[None, :, None, None])
        return (torch.nn.functional.conv2d(t, wd, None, 1, 1) + torch.from_numpy(res).double()).numpy()

    try:
        # fresh inputs every pass: a part read from a stale cache line (the previous launch's partial tile) would show
        for k in range(4 if variant >= 256 else 1):
            xk = x if k == 0 else (x * np.float32(1.0 + 0.37 * k) + np.float32(0.11 * k)).astype(np.float32)
            xd.copy_(torch.from_numpy(xk))
            yd.fill_(float("nan"))
            ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, 3, None,
                                          p(pre[0]), p(pre[1]), None, None, C.c_void_p(rd.data_ptr()), 0, C.c_void_p(yd.data_ptr())))
            expect = want(xk)
            assert np.abs(yd.cpu().numpy() - expect).max() < 5e-6 * max(1.0, np.abs(expect).max())
    finally:
        ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, -1))


# ---- opt-in "fast" precision (bf16x3 split on the bf16 matrix cores) -----------------------------------------------------
@pytest.mark.parametrize("cin,cout,size,batch,opts", [
    (256, 128, 64, 2, dict(pre=True, res=True)),       # a residual block's conv1
    (256, 256, 32, 1, dict(bias=True, post=True)),     # conv5 / conv9 shape, two 128-channel tiles
    (128, 64, 32, 3, dict(pre=True, res=True)),        # 64-channel tile
    (64, 64, 64, 1, dict(pre=True)),
    (32, 64, 32, 2, dict(bias=True)),                  # two 16-channel chunks
    (48, 192, 32, 1, dict(pre=True, bias=True)),       # 192 = 3 x 64 tiles, 3 chunks
    (84, 256, 32, 2, dict(bias=True, res=True)),       # conv7: 84 input channels in 6 chunks, the last one partly empty
    (256, 84, 64, 1, dict(bias=True)),                 # conv6 / conv10: 84 output channels in one 128-channel tile
    (84, 84, 32, 1, dict(pre=True, bias=True, post=True)),
])
def test_fast_conv_matches_torch(cin, cout, size, batch, opts):
    """mvlm_conv2d_fast (bf16x3-split operands, 6 cross products, fp32 accumulation) against torch float64: the error
    budget is that of fp32 itself (dropped terms <= 2^-23 per product), nothing like a bf16 convolution's 1e-2."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    rs = np.random.RandomState(cin + cout + size)
    x = rs.standard_normal((batch, cin, size, size)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    bias = rs.standard_normal(cout).astype(np.float32) if opts.get("bias") else None
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32)) if opts.get("pre") else None
    post = (rs.uniform(0.5, 1.5, cout).astype(np.float32), (rs.standard_normal(cout) * 0.3).astype(np.float32)) if opts.get("post") else None
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32) if opts.get("res") else None
    xd = dev(x)
    rd = dev(res) if res is not None else None
    yd = torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    p = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.check(ctx.lib.mvlm_conv2d_fast(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, p(bias),
                                       p(pre[0]) if pre else None, p(pre[1]) if pre else None,
                                       p(post[0]) if post else None, p(post[1]) if post else None,
                                       C.c_void_p(rd.data_ptr()) if rd is not None else None, C.c_void_p(yd.data_ptr())))
    t = torch.from_numpy(x).double()
    if pre:
        t = torch.relu(t * torch.from_numpy(pre[0]).double()[None, :, None, None] + torch.from_numpy(pre[1]).double()[None, :, None, None])
    y = torch.nn.functional.conv2d(t, torch.from_numpy(w).double(), None if bias is None else torch.from_numpy(bias).double(), 1, 1)
    if post:
        y = torch.relu(y * torch.from_numpy(post[0]).double()[None, :, None, None] + torch.from_numpy(post[1]).double()[None, :, None, None])
    if res is not None:
        y = y + torch.from_numpy(res).double()
    want = y.numpy()
    err = np.abs(yd.cpu().numpy() - want).max()
    assert err < 1e-5 * max(1.0, np.abs(want).max()), err


@pytest.mark.parametrize("family", ["bu3dfe", "dtu3d"])
def test_fast_precision_network_close_to_exact(family):
    """precision="fast" on the whole network: heatmaps within 1e-4 of the value range of the exact path's, argmax pixels
    equal except near-ties; switching back restores the exact path bit for bit.  84 landmarks: conv6 / conv10 / conv7 run
    on the bf16x3 kernel with padded channels; 73 landmarks: conv7 does (73 -> 80 input channels), conv6 / conv10 stay
    exact (73 of 128 output channels would waste too much)."""
    from conftest import seeded_images
    from mvlm_amd.prediction import BU3DFEPredictor, DTU3DPredictor

    imgs = dev(seeded_images(41, 4))
    if family == "bu3dfe":
        pred = BU3DFEPredictor(image_mode="RGB+depth", weights="synthetic:3", verbose=False)
    else:
        pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:4", verbose=False)
    exact_heat = pred.heatmaps_device(imgs).clone()
    exact_max = pred.predict_device(imgs).clone()
    pred.set_precision("fast")
    fast_heat = pred.heatmaps_device(imgs)
    fast_max = pred.predict_device(imgs)
    scale = exact_heat.abs().max().item()
    dev_heat = (fast_heat - exact_heat).abs().max().item()
    assert 0 < dev_heat < 1e-4 * scale, (dev_heat, scale)          # different arithmetic, fp32-class accuracy
    flips = (~torch.all(fast_max[:, :, :2] == exact_max[:, :, :2], dim=2)).sum().item()
    assert flips <= 0.03 * pred.get_lm_count() * 4, flips
    pred.set_precision("exact")
    assert torch.equal(pred.predict_device(imgs), exact_max)
