"""Round-4 GPU tests: two independent convolutions in one grid (conv_pair_kernel) against the single launches and torch,
the network with paired residual blocks, threads and ingest outside the pipeline lock, multi-device cases that run
wherever two GPUs are visible.  Run with -m gpu."""
import ctypes as C
import threading
import time
from collections import Counter
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import seeded_images

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# (kernel variant code = base id | lg(kparts0) << 8 | lg(kparts1) << 10, name of the base, cin, cout, size of problem 0, batch)
PAIR_CASES = [
    (0, "conv3x3_c128_t8x32", 256, 128, 64, 2),            # the dominant tile: 64x64 beside 32x32
    (10, "conv3x3_c64_t8x32", 128, 64, 128, 1),
    (28, "conv3x3_c64k8_t8x16", 64, 64, 32, 3),
    (27, "conv3x3_c32k8_t8x16", 128, 64, 32, 2),
    (19, "conv3x3_sk16_t4x8", 256, 128, 16, 5),            # split-K tiles: 16x16 beside 8x8
    (13, "conv3x3_sk_t4x8", 128, 64, 32, 1),
    (19 | (1 << 10), "conv3x3_sk16_t4x8", 256, 128, 16, 3),  # ... the 8x8 problem's input channels over two workgroups
    (20 | (1 << 8) | (2 << 10), "conv3x3_sk16_t4x4x2", 256, 128, 8, 5),  # 8x8 beside 4x4, K parts 2 and 4, odd image count
    (24, "conv3x3_sk8_t4x4x2", 64, 64, 8, 12),
]


@pytest.mark.parametrize("code,name,cin,cout,size,batch", PAIR_CASES)
def test_pair_launch_equals_single_launches(code, name, cin, cout, size, batch):
    """conv j of a skip block and of the next level's first block in ONE grid: each problem's result (raw copy and
    residual sum) equals, bit for bit, the single launch of the same kernel variant, and torch float64 within 5e-6."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    assert ctx.lib.mvlm_conv_variant_name(code & 255).decode() == name
    assert ctx.lib.mvlm_conv_variant_name(code | 0x1000).decode().startswith(name + "_pair")
    rs = np.random.RandomState(code + cin + size + batch)
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32))
    xs, ws, rs_, ys, raws = [], [], [], [], []
    for s in (size, size // 2):
        xs.append(dev(rs.standard_normal((batch, cin, s, s)).astype(np.float32)))
        ws.append((rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
        rs_.append(dev(rs.standard_normal((batch, cout, s, s)).astype(np.float32)))
        ys.append(torch.full((batch, cout, s, s), np.nan, dtype=torch.float32, device="cuda"))
        raws.append(torch.full((batch, cout, s, s), np.nan, dtype=torch.float32, device="cuda"))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    ctx.check(ctx.lib.mvlm_conv2d_pair(ctx.handle, batch, cin, cout, ptr(xs[0]), size, p(ws[0]), ptr(rs_[0]), ptr(raws[0]), ptr(ys[0]),
                                       ptr(xs[1]), size // 2, p(ws[1]), ptr(rs_[1]), ptr(raws[1]), ptr(ys[1]), p(pre[0]), p(pre[1]), code))
    for i, s in enumerate((size, size // 2)):
        single = torch.empty_like(ys[i])
        lg = (code >> (8 + 2 * i)) & 3
        ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, (code & 255) | (lg << 8)))
        try:
            ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, ptr(xs[i]), batch, cin, s, s, p(ws[i]), cout, 3, None, p(pre[0]), p(pre[1]),
                                          None, None, ptr(rs_[i]), 0, ptr(single)))
        finally:
            ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, -1))
        assert torch.equal(ys[i], single), f"problem {i}"
        assert torch.equal(raws[i] + rs_[i], ys[i])              # raw copy = the value before the residual add
        t = torch.relu(xs[i].cpu().double() * torch.from_numpy(pre[0]).double()[None, :, None, None]
                       + torch.from_numpy(pre[1]).double()[None, :, None, None])
        want = (torch.nn.functional.conv2d(t, torch.from_numpy(ws[i]).double(), None, 1, 1) + rs_[i].cpu().double()).numpy()
        assert np.abs(ys[i].cpu().numpy() - want).max() < 5e-6 * max(1.0, np.abs(want).max())


def test_pair_launch_refuses_what_it_cannot_serve():
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    x = torch.zeros((1, 64, 16, 16), device="cuda")
    w = np.zeros((64, 64, 3, 3), np.float32)
    y0, y1 = torch.empty((1, 64, 16, 16), device="cuda"), torch.empty((1, 64, 8, 8), device="cuda")
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def call(code, size1=8):
        return ctx.lib.mvlm_conv2d_pair(ctx.handle, 1, 64, 64, ptr(x), 16, p(w), None, None, ptr(y0), ptr(x), size1, p(w), None, None,
                                        ptr(y1), None, None, code)

    assert call(16) != 0 and b"two-problem" in ctx.lib.mvlm_last_error(ctx.handle)      # the 80-row tile has no pair form
    assert call(0) != 0 and b"multiple of the tile" in ctx.lib.mvlm_last_error(ctx.handle)  # 8x32-pixel tiles on 16x16
    assert call(19 | (3 << 8)) != 0                                                          # 64 input channels do not divide into 8 parts of 16-channel chunks
    assert call(19) == 0


@pytest.mark.parametrize("n_views", [2, 12])
def test_network_with_paired_residual_blocks(n_views):
    """The forward pass with the hourglass's independent blocks sharing launches (pairing 2: wherever one kernel variant can
    serve both; 1: the measured table) against the pass with one launch per convolution: the same network, so heatmaps agree
    to fp32 rounding (another kernel variant = another summation order, like another device batch), the argmax pixel of every
    plane is the same or a near-tie, and the launch graph is replayed in every mode."""
    from mvlm_amd.prediction import DTU3DPredictor

    imgs = dev(seeded_images(40 + n_views, n_views))
    pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:4", verbose=False)
    out = torch.empty((73, n_views, 3), dtype=torch.float32, device="cuda")
    pred.set_execution(graphs=False, pairing=0)
    want = pred.predict_device(imgs, out=out).clone()
    want_heat = pred.heatmaps_device(imgs[:2]).clone()
    scale = float(want_heat.abs().max())
    for mode in (2, 1):
        pred.set_execution(graphs=False, pairing=mode)
        eager = pred.predict_device(imgs, out=out).clone()
        heat = pred.heatmaps_device(imgs[:2])
        assert float((heat - want_heat).abs().max()) < 2e-5 * scale
        same = torch.all(eager[:, :, :2] == want[:, :, :2], dim=2)
        assert float(same.float().mean()) >= 0.99
        assert float((eager[:, :, 2] - want[:, :, 2]).abs().max()) < 2e-5 * scale
        pred.set_execution(graphs=True, pairing=mode)
        before = pred.execution_stats()
        for i in range(3):
            out.zero_()
            assert torch.equal(pred.predict_device(imgs, out=out), eager), f"mode {mode} pass {i}"
        after = pred.execution_stats()
        assert after["graph_failures"] == before["graph_failures"] and after["graph_replays"] >= before["graph_replays"] + 1
    # the paired launches really ran in mode 2
    ctx = pred.ctx
    pred.set_execution(graphs=False, pairing=2)
    ctx.check(ctx.lib.mvlm_cnn_set_profiling(ctx.handle, 1))
    try:
        pred.predict_device(imgs, out=out)
        cap = 512
        slot, var = (C.c_int32 * cap)(), (C.c_int32 * cap)()
        fl, ms = (C.c_double * cap)(), (C.c_float * cap)()
        n = ctx.lib.mvlm_cnn_get_profile(ctx.handle, slot, var, fl, ms, cap)
    finally:
        ctx.check(ctx.lib.mvlm_cnn_set_profiling(ctx.handle, 0))
    convs = [var[i] for i in range(n) if slot[i] >= 0]           # (slot -1 = the pool kernel behind a block whose tiles cannot pool)
    pairs = [v for v in convs if v & 0x1000]
    assert len(convs) == 138 - len(pairs) and len(pairs) >= 24, (n, len(convs), len(pairs))
    assert all(b"_pair" in ctx.lib.mvlm_conv_variant_name(v) for v in pairs)
    pred.set_execution(graphs=True, pairing=1)


# ---- cases that need two GPUs: collected everywhere, run wherever two devices are visible ---------------------------------
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")


@two_gpus
def test_n_gpus_replicas_on_two_devices():
    """``n_gpus=2`` with the replica on a SECOND device (paulsenpredictor.py:100-105, nn.DataParallel): device-to-device
    scatter of the views, the replica's pass on its own stream and launch graph, the peer write of its maxima slice - bit
    for bit the single-device result, launch by launch and replayed.  (The one-GPU box only rehearses this with every replica
    on device 0: tests/test_gpu_round3.py.)"""
    from mvlm_amd.prediction import DTU3DPredictor

    x = torch.from_numpy(seeded_images(3, 9)).cuda()
    single = DTU3DPredictor(image_mode="RGB", weights="synthetic:5", verbose=False)
    multi = DTU3DPredictor(image_mode="RGB", weights="synthetic:5", verbose=False, n_gpus=2)
    assert [r.ctx.device for r in multi._replicas] == [1]
    for n in (9, 4, 2):
        # every device's share has its own batch size: compare with the single device run on the same shares
        from mvlm_amd.parallel import shard_range

        want = torch.cat([single.predict_device(x[slice(*shard_range(n, r, 2))].contiguous()) for r in range(2)], dim=1).cpu().numpy()
        out = torch.empty((73, n, 3), device="cuda:0")
        for _ in range(3):  # launch by launch, capture, replay - on both devices
            got = multi.predict_device(x[:n].contiguous(), out=out).cpu().numpy()
            np.testing.assert_array_equal(got, want)


@two_gpus
def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` exactly as the driver starts it, one RCCL rank per GPU: the JSON line carries the
    scaling breakdown (per-rank times, the two collectives' times, the shard run unsharded)."""
    import json
    import os
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MVLM_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-views", "0",
                        "--no-fast-mode", "--views-total", "24"], capture_output=True, text=True, timeout=900, cwd=str(REPO),
                       env=dict(env, MVLM_BENCH_NO_INGEST="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert rec["n_gpus"] == 2 and "nccl world_size 2" in rec["config"]["parallelism"]
    sb = rec["scaling_breakdown"]
    assert len(sb["per_rank_ms_per_step"]["all"]) == 2 and sb["single_gpu_shard_ms"] > 0
    assert sb["all_gather_ms_per_step"] is not None and sb["collective_clock"].startswith("hip events")


# ---- threads on one pipeline --------------------------------------------------------------------------------------------
def test_threads_at_twelve_views_follow_the_admission_order(tmp_path):
    """12 views: the poses come from the global RNG too (render3d.py:79-89), so a threaded run equals the sequential run IN
    THE ORDER THE PIPELINE LOCK ADMITTED THE CALLS.  The order is recorded under the lock (ingest runs outside it since round
    4, so it is not the order of the calls), then replayed sequentially from the same seed: bit for bit the same."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    files = [write_face_like_obj(tmp_path / f"s{i}.obj", grid=36 + 8 * i, tex_size=64, seed=i + 1) for i in range(3)]
    pipe = pipeline.create_pipeline("dtu3d", n_views=12, weights="synthetic:3", verbose=False)
    admitted = []
    fused = pipe._predict_fused

    def recording(file_name, mesh=None):
        admitted.append(Path(file_name).name)   # called with the pipeline lock held
        return fused(file_name, mesh=mesh)

    pipe._predict_fused = recording
    for attempt in range(3):
        admitted.clear()
        out, errors = {}, []
        start = threading.Barrier(3)

        def work(f):
            try:
                start.wait()
                out[f.name] = pipe.predict_one_file(f)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        np.random.seed(70 + attempt)
        threads = [threading.Thread(target=work, args=(f,)) for f in files]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        order = list(admitted)
        assert sorted(order) == sorted(f.name for f in files)
        np.random.seed(70 + attempt)
        for name in order:
            want = pipe.predict_one_file(tmp_path / name)
            np.testing.assert_array_equal(out[name], want)


def test_ingest_runs_outside_the_pipeline_lock(tmp_path):
    """predict_one_file parses the OBJ and decodes the JPEG BEFORE it takes the pipeline lock (general_pipeline.py: load ->
    lock -> GPU section): two server threads (3DMD_server.py:26-31) x two RGB scans with a 2048x2048 texture finish in less
    than 1.6 x the time one thread needs for its two scans (2.0 x if the ingest were serialised with the GPU section), and
    give the sequential results."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    files = [write_face_like_obj(tmp_path / f"rgb{i}.obj", grid=224, tex_size=2048, seed=i) for i in range(4)]
    pipe = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:3", verbose=False, image_mode="RGB+depth")
    # (round 5: the JPEG is decoded on the device by default and the whole ingest is 4 ms; the property tested here is WHERE the
    # ingest runs, so it is made heavy again - libjpeg on the host, 16 ms per texture)
    def seeded(f):
        np.random.seed(5)  # (the RANSAC draws)
        return pipe.predict_one_file(f)

    device_decoded = {f.name: seeded(f) for f in files}
    pipe.renderer_3d.texture_decode = "host"
    for f in files:
        assert np.array_equal(seeded(f), device_decoded[f.name])   # page cache, launch graph, buffers
    torch.cuda.synchronize()
    want = {f.name: pipe.predict_one_file(f) for f in files}   # the fixed 8-view table: no RNG in the poses, draws only with >= 3 survivors

    def one_thread():
        t0 = time.perf_counter()
        for f in files[:2]:
            pipe.predict_one_file(f)
        return time.perf_counter() - t0

    def two_threads():
        got = {}

        def work(mine):
            for f in mine:
                got[f.name] = pipe.predict_one_file(f)

        threads = [threading.Thread(target=work, args=(files[:2],)), threading.Thread(target=work, args=(files[2:],))]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return time.perf_counter() - t0, got

    t1 = min(one_thread() for _ in range(3))
    best, got = None, None
    for _ in range(3):
        t2, got = two_threads()
        best = t2 if best is None else min(best, t2)
    assert set(got) == set(want)
    assert best < 1.6 * t1, f"two threads x two scans {1e3 * best:.1f} ms, one thread x two scans {1e3 * t1:.1f} ms"
    assert pipe.timings["load"] > 0 and pipe.timings["total"] > pipe.timings["load"]


# ---- "moment" selection out of the fused path (paulsenpredictor.py:129-156) ----------------------------------------------
@pytest.mark.parametrize("cls_name,mode,n_views,device_batch", [("dtu3d", "RGB", 5, 2), ("bu3dfe", "RGB+depth", 3, None), ("dtu3d", "geometry+depth", 2, None)])
def test_fused_moment_equals_the_materialised_heatmaps(monkeypatch, cls_name, mode, n_views, device_batch):
    """selection_method="moment" without the [N,NL,256,256] tensor: the 31x31 window around each fused-argmax peak is
    recomputed from conv10's output with conv11's own arithmetic (80-row tiles with the 16-row strip for 73 landmarks, 84
    rows with the 4-row strip for 84).  Bit for bit (a) the same network's heatmaps through HBM + mvlm_heatmap_maxima and
    (b) the oracle's moment rule (the reference's find_heat_map_maxima restated) on those heatmaps; device batches smaller
    than the view count (view offsets) included; the launch graph is replayed."""
    from mvlm_amd.prediction import BU3DFEPredictor, DTU3DPredictor
    from oracle import cnn as ocnn

    cls = DTU3DPredictor if cls_name == "dtu3d" else BU3DFEPredictor
    imgs = dev(seeded_images(60 + n_views, n_views))
    pred = cls(image_mode=mode, weights="synthetic:9", selection_method="moment", verbose=False, device_batch=device_batch)
    nl = pred.get_lm_count()
    out = torch.empty((nl, n_views, 3), dtype=torch.float32, device="cuda")
    before = pred.execution_stats()
    fused = [pred.predict_device(imgs, out=out).clone() for _ in range(3)]   # launch by launch, capture, replay
    after = pred.execution_stats()
    assert after["graph_replays"] > before["graph_replays"] and after["graph_failures"] == before["graph_failures"]
    assert torch.equal(fused[0], fused[1]) and torch.equal(fused[0], fused[2])
    monkeypatch.setenv("MVLM_MOMENT_MATERIALISED", "1")
    materialised = pred.predict_device(imgs).cpu().numpy()
    monkeypatch.delenv("MVLM_MOMENT_MATERIALISED")
    got = fused[0].cpu().numpy()
    np.testing.assert_array_equal(got, materialised)
    heat = pred.heatmaps_device(imgs).cpu().numpy()
    np.testing.assert_array_equal(got, ocnn.maxima_from_heatmaps(heat, "moment"))
    pred.selection_method = "simple"                    # a plain attribute in the reference (paulsenpredictor.py:57): switchable
    simple = pred.predict_device(imgs).cpu().numpy()
    np.testing.assert_array_equal(simple, ocnn.maxima_from_heatmaps(heat, "simple"))
    refined = np.any(got[:, :, :2] != simple[:, :, :2], axis=2)
    assert refined.mean() > 0.2                         # peaks more than 15 px inside are refined; the others keep the simple form
    np.testing.assert_array_equal(got[:, :, 2], simple[:, :, 2])


def test_fused_moment_on_planted_peaks_end_to_end():
    """The planted detector (tests/planted.py: the final heatmaps peak at known surface points) with "moment" through
    predict_mesh_device - render, network, fused argmax, window recomputation, centroid, rays, RANSAC inlier refit, snap -
    equals the oracle's pipeline run with the same selection method, and finds the planted points."""
    import contextlib
    import io

    from mvlm_amd import arch, config
    from mvlm_amd.pipeline import pipeline_from_config
    from oracle import estimator as oest
    from oracle import cnn as ocnn
    from test_planted_cpu import planted_scene

    mesh, pts, sd, poses = planted_scene(n_views=16)
    pipe = pipeline_from_config(config.default_config("DTU3D", "RGB", n_views=16), weights=sd, verbose=False)
    pipe.predictor_2d.selection_method = "moment"
    np.random.seed(1)
    got, err = pipe.predict_mesh_device(mesh, poses)
    assert err < 10.0
    images = pipe.renderer_3d.render_device(mesh, poses)
    gmax = pipe.predictor_2d.predict_device(images).cpu().numpy()
    heat = pipe.predictor_2d.heatmaps_device(images).cpu().numpy()
    np.testing.assert_array_equal(gmax, ocnn.maxima_from_heatmaps(heat, "moment"))
    s, e = oest.estimate_landmark_lines(256, gmax, poses)
    np.random.seed(1)
    want, werr = oest.estimate_landmarks_from_lines(gmax, s, e)
    from oracle import surface

    np.testing.assert_allclose(got, surface.project_landmarks_to_surface(mesh.verts, mesh.tris, want), rtol=0, atol=1e-8)
    d = np.linalg.norm(got - pts, axis=1)
    # (the planted heatmaps are hat functions on a sloped background: their 31x31 centroid sits a little further from the
    #  knot than the argmax pixel does - 6.9 units at worst against 6.0 with "simple")
    assert d.max() < 8.0 and np.median(d) < 4.0


def test_batched_scans_write_the_pre_aligned_mesh_too(tmp_path):
    """pre-align.write_pre_aligned (utils3d.py:489-494) on the grouped path: predict_files(batch_scans=2) writes
    <stem>_pre_transform_mesh.vtk for every scan, as the one-by-one loop does."""
    from mvlm_amd import config
    from mvlm_amd.pipeline import pipeline_from_config
    from mvlm_amd.utils.mesh_io import write_obj
    from mvlm_amd.utils.synthetic import face_like_mesh, unaligned_copy

    cfg = config.default_config("DTU3D-RGB_Artec3D", n_views=8)
    assert cfg["pre-align"]["write_pre_aligned"]
    cfg["process_3d"]["write_renderings"] = False
    files = []
    for sd in (1, 2, 3):
        raw = unaligned_copy(face_like_mesh(40, 64, sd), cfg["pre-align"])
        path = tmp_path / f"scan{sd}.obj"
        write_obj(path, raw.verts, raw.tris, raw.uvs, raw.texture)
        files.append(path)
    pipe = pipeline_from_config(cfg, weights="synthetic:9", verbose=False)
    out_dir = tmp_path / "aligned"
    out_dir.mkdir()
    pipe.write_pre_aligned_folder = out_dir
    assert pipe._groupable(2)
    np.random.seed(2)
    res = [lm for _, lm in pipe.predict_files(files, batch_scans=2)]
    assert all(lm is not None for lm in res)
    for f in files:
        vtk = out_dir / f"{f.stem}_pre_transform_mesh.vtk"
        assert vtk.exists() and "POLYDATA" in vtk.read_text()[:200]


# ---- the end-to-end matrix over more than one draw of weights and scan -----------------------------------------------------
@pytest.mark.parametrize("seed", [11, 17])
@pytest.mark.parametrize("dataset,mode,n_views,grid", [("BU_3DFE", "depth", 8, 51), ("DTU3D", "geometry+depth", 12, 60),
                                                       ("BU_3DFE", "RGB+depth", 16, 60)])
def test_end_to_end_matrix_over_seeds(dataset, mode, n_views, grid, seed):
    """test_end_to_end_config_matrix_against_oracle runs one seed (13) of weights and scan per configuration; the small
    cases again with seeds 11 and 17 (render bit-identical, argmax planes, landmarks within BASELINE's 1e-3 model units
    wherever every view picked the oracle's pixel, equal RANSAC error when none differs)."""
    from test_gpu_e2e_matrix import _e2e_config

    got, gerr, want, werr, inter, gmax, unit = _e2e_config(dataset, mode, n_views, grid, seed)
    diff_views = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)
    assert diff_views.mean() <= 0.002, f"{int(diff_views.sum())} of {diff_views.size} argmax planes differ"
    same = ~diff_views.any(axis=1)
    assert same.mean() > 0.9
    assert np.abs(got[same] - want[same]).max() < 1e-3 * unit
    # a landmark with differing views moves by at most one pixel (1.17 units) per differing view over its >= 3 inliers
    moved = np.abs(got - want).max(axis=1)
    assert np.all(moved[~same] <= 1.2 * unit * diff_views.sum(axis=1)[~same])
    if same.all():
        assert abs(gerr - werr) <= 1e-6 * max(1.0, abs(werr))


# ---- second opt-in precision: f16x2-split operands ("fast16") --------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,size,batch,opts", [
    (256, 128, 64, 2, dict(pre=True, res=True)),       # a residual block's conv1
    (256, 256, 32, 1, dict(bias=True, post=True)),     # conv5 / conv9 shape, two 128-channel tiles
    (128, 64, 32, 3, dict(pre=True, res=True)),        # 64-channel tile (16 rows: three staging items)
    (64, 64, 64, 1, dict(pre=True)),
    (32, 64, 32, 2, dict(bias=True)),                  # two 16-channel chunks
    (84, 256, 32, 2, dict(bias=True, res=True)),       # conv7: the last chunk partly empty
    (256, 84, 64, 1, dict(bias=True)),                 # conv6 / conv10: 84 output channels in one 128-channel tile
    (128, 128, 32, 1, dict(pre=True, scale=1e-3)),     # small weights: the per-layer power-of-two scale
    (64, 128, 32, 1, dict(pre=True, scale=300.0, xscale=50.0)),   # large weights and activations, still inside fp16
    (64, 32, 64, 2, dict(pre=True, res=True)),         # the 32-channel tile (1 x 8 waves, 18-slot staging): conv2.conv2 / conv3.conv2
    (32, 32, 32, 3, dict(pre=True, res=True)),         # ... conv2.conv3 / conv3.conv3: two chunks
    (48, 24, 32, 1, dict(bias=True)),                  # ... 24 of its 32 channels
])
def test_fast16_conv_matches_torch(cin, cout, size, batch, opts):
    """mvlm_conv2d_fast16 (two fp16 terms per operand, 3 cross products, fp32 accumulation, weights scaled by the layer's
    power of two) against torch float64: fp32-class accuracy (per product <= 3 x 2^-22), nothing like an fp16 convolution's 1e-3."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    rs = np.random.RandomState(cin + cout + size)
    x = (rs.standard_normal((batch, cin, size, size)) * opts.get("xscale", 1.0)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9) * opts.get("scale", 1.0)).astype(np.float32)
    bias = rs.standard_normal(cout).astype(np.float32) if opts.get("bias") else None
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32)) if opts.get("pre") else None
    post = (rs.uniform(0.5, 1.5, cout).astype(np.float32), (rs.standard_normal(cout) * 0.3).astype(np.float32)) if opts.get("post") else None
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32) if opts.get("res") else None
    xd = dev(x)
    rd = dev(res) if res is not None else None
    yd = torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    q = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
    ctx.check(ctx.lib.mvlm_conv2d_fast16(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, q(w), cout, q(bias),
                                         q(pre[0]) if pre else None, q(pre[1]) if pre else None,
                                         q(post[0]) if post else None, q(post[1]) if post else None,
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
def test_fast16_network_close_to_exact(family):
    """precision="fast16" on the whole network: heatmaps within 1e-4 of the value range of the exact path's, argmax pixels
    equal except near-ties; switching back restores the exact path bit for bit."""
    from mvlm_amd.prediction import BU3DFEPredictor, DTU3DPredictor

    imgs = dev(seeded_images(41, 4))
    if family == "bu3dfe":
        pred = BU3DFEPredictor(image_mode="RGB+depth", weights="synthetic:3", verbose=False)
    else:
        pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:4", verbose=False)
    exact_heat = pred.heatmaps_device(imgs).clone()
    exact_max = pred.predict_device(imgs).clone()
    pred.set_precision("fast16")
    heat = pred.heatmaps_device(imgs)
    fmax = pred.predict_device(imgs)
    assert torch.isfinite(heat).all()
    scale = exact_heat.abs().max().item()
    dev_heat = (heat - exact_heat).abs().max().item()
    assert 0 < dev_heat < 1e-4 * scale, (dev_heat, scale)
    flips = (~torch.all(fmax[:, :, :2] == exact_max[:, :, :2], dim=2)).sum().item()
    assert flips <= 0.03 * pred.get_lm_count() * 4, flips
    pred.set_precision("exact")
    assert torch.equal(pred.predict_device(imgs), exact_max)
    assert pred.fast16_fallbacks == 0


def test_fast16_serves_the_up_path_blocks():
    """The last block of a hourglass level on the way up (2x2 scatter into the skip tensor, paulsenpredictor.py:334-359) runs
    on the split kernel in the f16x2 form only: its slots report the split variant under "fast16", an exact tile under
    "fast" (conv_fast.hip: mvlm_conv_fast_ok)."""
    from mvlm_amd import arch
    from mvlm_amd.prediction import DTU3DPredictor

    pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:4", verbose=False)
    imgs = dev(seeded_images(43, 4))
    lib, h = pred.ctx.lib, pred.ctx.handle
    names = [sl.name for sl in arch.conv_slots(73, 3)]
    cap = 1024
    slot, var = (C.c_int32 * cap)(), (C.c_int32 * cap)()
    fl, ms = (C.c_double * cap)(), (C.c_float * cap)()

    def variants(precision):
        pred.set_precision(precision)
        pred.predict_device(imgs)
        lib.mvlm_cnn_set_profiling(h, 1)
        pred.predict_device(imgs)
        torch.cuda.synchronize()
        n = lib.mvlm_cnn_get_profile(h, slot, var, fl, ms, cap)
        lib.mvlm_cnn_set_profiling(h, 0)
        assert n > 0
        return {names[slot[i]]: lib.mvlm_conv_variant_name(var[i]).decode() for i in range(n) if slot[i] >= 0}

    try:
        v16, v3 = variants("fast16"), variants("fast")
    finally:
        pred.set_precision("exact")
    up = [f"hg{g}.rb{b}.conv{c}" for g in (1, 2) for b in (18, 20) for c in (1, 2, 3)]  # the 32x32 and 64x64 levels' last blocks
    assert all("f16x2" in v16[k] for k in up), {k: v16[k] for k in up}
    assert not any("bf16x3" in v3[k] for k in up), {k: v3[k] for k in up}
    assert "f16x2" in v16["hg1.rb19.conv1"] and "bf16x3" in v3["hg1.rb19.conv1"]  # the plain blocks: both forms
    stem32 = ["conv2.conv2", "conv2.conv3", "conv3.conv2", "conv3.conv3"]  # 64 -> 32 and 32 -> 32 channels: the f16x2 form's 32-channel tile
    assert all("f16x2" in v16[k] for k in stem32), {k: v16[k] for k in stem32}
    assert not any("bf16x3" in v3[k] for k in stem32), {k: v3[k] for k in stem32}


@pytest.mark.parametrize("name,mode,n_views", [("dtu3d", "RGB", 16), ("bu3dfe", "RGB+depth", 12)])
def test_fast16_against_the_oracle(name, mode, n_views):
    """precision="fast16" against the CPU ORACLE (never the default, never bench.py's value): the same render, at most
    0.2 % of the argmax planes move (near-ties), every landmark whose views all picked the oracle's pixel within 1e-3."""
    import contextlib
    import io

    from mvlm_amd import arch, pipeline, weights
    from mvlm_amd.utils.synthetic import face_like_mesh
    from oracle import pipeline as opipe

    pipe = pipeline.create_pipeline(name, n_views=n_views, weights="synthetic:11", verbose=False, image_mode=mode, precision="fast16")
    assert pipe.predictor_2d.precision == "fast16"
    mesh = face_like_mesh(60, 128, 11)
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    np.random.seed(1)
    got, _ = pipe.predict_mesh_device(mesh, poses)
    assert pipe.predictor_2d.precision == "fast16" and pipe.predictor_2d.fast16_fallbacks == 0
    gmax = pipe.predictor_2d.predict_device(pipe.renderer_3d.render_device(mesh, poses)).cpu().numpy()
    nl = pipe.get_lm_count()
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        want, _, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses,
                                            weights.synthetic_state_dict(nl, arch.IMAGE_CHANNELS[mode], seed=11), arch.CHANNEL_SELECT[mode])
    diff = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)
    assert diff.mean() <= 0.002, f"{int(diff.sum())} of {diff.size} argmax planes differ from the oracle"
    scores = np.abs(gmax[:, :, 2] - inter["maxima"][:, :, 2])[~diff]
    assert scores.max() < 1e-4 * max(1.0, np.abs(inter["maxima"][:, :, 2]).max())
    # Downstream of the maxima the pass is the exact one: the oracle's estimator fed with THESE maxima (same RNG seed) gives
    # the product's landmarks to 1e-8.
    from oracle import estimator as oest
    from oracle import surface

    s_, e_ = oest.estimate_landmark_lines(256, gmax, poses)
    np.random.seed(1)
    fed, _ = oest.estimate_landmarks_from_lines(gmax, s_, e_)
    np.testing.assert_allclose(got, surface.project_landmarks_to_surface(mesh.verts, mesh.tris, fed), rtol=0, atol=1e-8)
    # Against the oracle's OWN maxima the comparison is rank-sensitive: the view filter keeps the views whose score exceeds
    # the landmark's median (estimator3d.py:140-147), so scores that differ in the 6th digit can swap two views around the
    # median (another line in the bundle), and a landmark whose median pair ties changes its survivor COUNT, which shifts the
    # global RNG's draws of every landmark after it (estimator3d.py:105).  So EVERY landmark with the oracle's pixels and
    # the oracle's surviving views is compared with the oracle's result for the draw the product made for it
    # (tests/parity_helpers.py), and the landmarks before the first shift of the RNG stream with the oracle's run itself.
    from parity_helpers import compare_with_the_oracle_landmark_by_landmark, survivors

    same, _, worst = compare_with_the_oracle_landmark_by_landmark(got, gmax, inter, mesh, pipe.estimator_3d, seed=1)
    assert same.mean() > 0.8, same.mean()
    assert worst < 1e-3, worst
    sg, so = survivors(gmax), survivors(inter["maxima"])
    unshifted = same & ~(np.cumsum(sg.sum(axis=1) != so.sum(axis=1)) > 0)
    assert np.abs(got[unshifted] - want[unshifted]).max() < 1e-3


def test_fast16_overflow_falls_back_to_bf16x3(capsys):
    """fp16 ends at 65504: with activations beyond that the f16x2 pass returns non-finite maxima - the predictor (numpy slot)
    and the pipeline (fused path) notice, say so, repeat the pass on the bf16x3 form and stay there."""
    from mvlm_amd import pipeline, weights
    from mvlm_amd.utils.synthetic import face_like_mesh

    sd = weights.synthetic_state_dict(73, 3, seed=5)
    sd["conv1.weight"] = (sd["conv1.weight"] * 3.0e6).astype(np.float32)      # conv2.conv1's input leaves fp16's range
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights=sd, verbose=False, image_mode="RGB", precision="fast")
    mesh = face_like_mesh(40, 64, 3)
    poses = pipe.renderer_3d.generate_3d_transformations()
    np.random.seed(1)
    want, _ = pipe.predict_mesh_device(mesh, poses)
    assert np.isfinite(want).all()
    pipe.predictor_2d.set_precision("fast16")
    np.random.seed(1)
    got, _ = pipe.predict_mesh_device(mesh, poses)
    assert pipe.predictor_2d.precision == "fast" and pipe.predictor_2d.fast16_fallbacks == 1
    assert "fp16 range" in capsys.readouterr().out
    np.testing.assert_array_equal(got, want)
    pipe.predictor_2d.set_precision("fast16")
    imgs = pipe.renderer_3d.render_device(mesh, poses).cpu().numpy()
    lms, valid = pipe.predictor_2d.predict_landmarks_from_images(imgs)
    assert np.isfinite(lms).all() and pipe.predictor_2d.precision == "fast" and pipe.predictor_2d.fast16_fallbacks == 2


def test_draw_table_of_a_new_landmark_count_is_not_overwritten():
    """Regression (round 4): the RANSAC draw table lives on the device and is written by the estimator's UPLOAD stream.  It used
    to be allocated from the compute stream's pool of torch's caching allocator; a torch op issued right before the first plan
    of a NEW landmark count could leave a freed temporary there whose kernel was still queued, the table took that block, and
    the late kernel overwrote the draws (the fused path then disagreed with the slot estimator on the same maxima).  Two
    pipelines with different landmark counts in one process, torch temporaries in flight before the second one's first plan."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import face_like_mesh

    mesh = face_like_mesh(60, 128, 11)
    for name, mode, nv in (("dtu3d", "RGB", 16), ("bu3dfe", "RGB+depth", 12)):
        pipe = pipeline.create_pipeline(name, n_views=nv, weights="synthetic:11", verbose=False, image_mode=mode)
        np.random.seed(0)
        poses = pipe.renderer_3d.generate_3d_transformations()
        junk = [torch.rand((84, 12, 3), device="cuda") for _ in range(8)]
        _ = [(~torch.isfinite(j[:, :, 2])).any() for j in junk]      # temporaries allocated and freed on the compute stream
        del junk
        np.random.seed(1)
        got, _ = pipe.predict_mesh_device(mesh, poses)
        gmax = pipe._buffers["maxima"].cpu().numpy()
        e3 = pipe.estimator_3d
        s2, e2 = e3.estimate_landmark_lines(np.zeros((nv, 256, 256, 4), np.float32), gmax, poses)
        np.random.seed(1)
        out2, _ = e3.estimate_landmarks_from_lines(gmax, s2, e2)
        np.testing.assert_array_equal(got, e3.project_landmarks_to_surface(mesh, out2))


# ---- split-K tiles with two 32-pixel columns per workgroup ---------------------------------------------------------------
@pytest.mark.parametrize("variant,name,cin,cout,size,batch,scatter", [
    (30, "conv3x3_sk16_t8x8", 256, 128, 16, 3, False), (30, "conv3x3_sk16_t8x8", 128, 64, 32, 2, False), (30, "conv3x3_sk16_t8x8", 64, 64, 8, 5, True),
    (31, "conv3x3_sk8_t8x8", 256, 128, 32, 1, True), (31, "conv3x3_sk8_t8x8", 64, 64, 16, 7, False), (31, "conv3x3_sk8_t8x8", 128, 64, 8, 2, False),
    (32, "conv3x3_sk16_t4x16", 256, 128, 16, 5, False), (32, "conv3x3_sk16_t4x16", 128, 64, 32, 1, True),
])
def test_two_column_split_k_tiles_match_torch(variant, name, cin, cout, size, batch, scatter):
    """Round 4: split-K tiles whose four waves multiply TWO 32-pixel columns on one staged weight slice (the small levels are
    bound by staging weights for 32 pixels of matrix work), forced onto residual-block shaped layers (pre-BN, residual) and,
    through the network's own executor, onto the scatter form - against torch float64."""
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    assert ctx.lib.mvlm_conv_variant_name(variant).decode() == name
    rs = np.random.RandomState(cin + size + batch + variant)
    x = rs.standard_normal((batch, cin, size, size)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    pre = (rs.uniform(0.5, 1.5, cin).astype(np.float32), (rs.standard_normal(cin) * 0.3).astype(np.float32))
    res = rs.standard_normal((batch, cout, size, size)).astype(np.float32)
    xd, rd = dev(x), dev(res)
    yd = torch.empty((batch, cout, size, size), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, variant))
    try:
        ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, 3, None,
                                      p(pre[0]), p(pre[1]), None, None, C.c_void_p(rd.data_ptr()), 0, C.c_void_p(yd.data_ptr())))
        # K parts have no form on these tiles: refused, not mis-run
        ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, variant + 256))
        assert ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(xd.data_ptr()), batch, cin, size, size, p(w), cout, 3, None,
                                   p(pre[0]), p(pre[1]), None, None, C.c_void_p(rd.data_ptr()), 0, C.c_void_p(yd.data_ptr())) != 0
    finally:
        ctx.check(ctx.lib.mvlm_conv_force_variant(ctx.handle, -1))
    t = torch.relu(torch.from_numpy(x).double() * torch.from_numpy(pre[0]).double()[None, :, None, None]
                   + torch.from_numpy(pre[1]).double()[None, :, None, None])
    want = (torch.nn.functional.conv2d(t, torch.from_numpy(w).double(), None, 1, 1) + torch.from_numpy(res).double()).numpy()
    assert np.abs(yd.cpu().numpy() - want).max() < 5e-6 * max(1.0, np.abs(want).max())


def test_two_column_split_k_tiles_in_the_network():
    """The same tiles serving every layer they can (plain, scatter and pooled kinds at <= 32x32) inside a forward pass: the
    heatmaps stay within fp32 rounding of the default dispatch."""
    from mvlm_amd.prediction import DTU3DPredictor

    imgs = dev(seeded_images(77, 3))
    pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:6", verbose=False)
    pred.set_execution(graphs=False, pairing=0)
    want = pred.heatmaps_device(imgs).clone()
    ctx, lib = pred.ctx, pred.ctx.lib
    n = 0
    for cin, cout in ((256, 128), (128, 64), (64, 64)):
        for size in (32, 16, 8):
            for kind in (0, 1, 2):
                for v in (30, 31, 32):
                    if lib.mvlm_conv_variant_serves(v, 3, cin, cout, size, kind):
                        ctx.check(lib.mvlm_conv_set_override(ctx.handle, 3, cin, cout, size, kind, v))
                        n += 1
                        break
    assert n >= 20
    try:
        got = pred.heatmaps_device(imgs)
        assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
        assert not torch.equal(got, want)        # other tiles, another summation order
    finally:
        ctx.check(lib.mvlm_conv_set_override(ctx.handle, 0, 0, 0, 0, 0, -1))
    assert torch.equal(pred.heatmaps_device(imgs), want)


def test_one_context_under_two_streams_stays_ordered():
    """Two users of the process-wide renderer context under different torch streams (two pipelines of one device): the
    context's scratch (bins, keys, transformed vertices) is reused from call to call, so mvlm_set_stream makes the new stream
    wait for the work enqueued on the previous one.  Renders issued alternately on two streams, without any host wait in
    between, equal the renders issued one after the other."""
    from mvlm_amd.utils import HipRenderer3D
    from mvlm_amd.utils.synthetic import face_like_mesh

    r = HipRenderer3D(n_views=24, verbose=False)
    meshes = [face_like_mesh(120, 64, 3), face_like_mesh(90, 64, 4)]
    np.random.seed(3)
    poses = r.generate_3d_transformations()
    want = [r.render_device(m, poses).clone() for m in meshes]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(6):
        for m, st in zip(meshes, (s1, s2)):
            with torch.cuda.stream(st):
                outs.append((rep, r.render_device(m, poses)))
    torch.cuda.synchronize()
    for k, (rep, o) in enumerate(outs):
        assert torch.equal(o, want[k % 2]), (rep, k % 2)
    r.check()


def test_new_entry_points_refuse_bad_arguments():
    """The round's C entry points report misuse through the status code and mvlm_last_error (nothing throws, nothing is launched)."""
    from mvlm_amd import _lib, weights
    from mvlm_amd.prediction import DTU3DPredictor

    fresh = _lib.Context(0)
    lib = fresh.lib
    assert lib.mvlm_cnn_set_selection(fresh.handle, 1) != 0 and b"mvlm_cnn_load comes first" in lib.mvlm_last_error(fresh.handle)
    assert lib.mvlm_cnn_set_pairing(fresh.handle, 3) != 0 and lib.mvlm_cnn_set_pairing(fresh.handle, -1) != 0
    assert lib.mvlm_cnn_set_precision(fresh.handle, 2) != 0 and b"load_fast16" in lib.mvlm_last_error(fresh.handle)
    v = C.c_int(7)
    assert lib.mvlm_cnn_fast16_overflowed(fresh.handle, C.byref(v)) == 0 and v.value == 0   # nothing loaded: nothing overflowed
    assert lib.mvlm_cnn_fast16_overflowed(fresh.handle, None) != 0
    # overrides: a variant that cannot serve the shape is refused, a serving one accepted, ksize 0 clears
    assert lib.mvlm_conv_set_override(fresh.handle, 3, 256, 128, 16, 0, 0) != 0            # 8x32-pixel tiles on a 16x16 map
    assert lib.mvlm_conv_set_override(fresh.handle, 3, 256, 128, 16, 0, 30) == 0
    assert lib.mvlm_conv_set_override(fresh.handle, 3, 256, 128, 16, 0, 30 + 256) != 0     # no K parts on the two-column tiles
    assert lib.mvlm_conv_set_override(fresh.handle, 0, 0, 0, 0, 0, -1) == 0
    assert lib.mvlm_conv_variant_serves(16, 3, 256, 80, 128, 0) == 0                       # the 80-row tile is conv6 / conv10's alone
    assert lib.mvlm_conv_variant_serves(0, 3, 256, 128, 128, 2) == 1 and lib.mvlm_conv_variant_serves(0, 1, 64, 128, 256, 0) == 0
    fresh.close()
    pred = DTU3DPredictor(image_mode="RGB", weights="synthetic:2", verbose=False)
    ctx = pred.ctx
    assert lib.mvlm_cnn_set_selection(ctx.handle, 2) != 0 and lib.mvlm_cnn_set_selection(ctx.handle, 1) == 0 and lib.mvlm_cnn_set_selection(ctx.handle, 0) == 0
    blob, off, unscale = weights.pack_fast16_for_device(pred._state_dict, 73, 3, pred._desc)
    bad = unscale.copy()
    bad[int(np.argmax(off >= 0))] = 0.0
    q16 = blob.ctypes.data_as(C.POINTER(C.c_uint16))
    assert lib.mvlm_cnn_load_fast16(ctx.handle, q16, blob.size, off.ctypes.data_as(C.POINTER(C.c_int64)), p(bad), off.shape[0]) != 0
    assert b"inverse scale" in lib.mvlm_last_error(ctx.handle)
    assert lib.mvlm_cnn_load_fast16(ctx.handle, q16, blob.size - 8, off.ctypes.data_as(C.POINTER(C.c_int64)), p(unscale), off.shape[0]) != 0
    assert lib.mvlm_cnn_load_fast16(ctx.handle, q16, blob.size, off.ctypes.data_as(C.POINTER(C.c_int64)), p(unscale), off.shape[0]) == 0
    # the packer refuses non-finite weights (their scale would be meaningless)
    w = np.ones((64, 16, 3, 3), np.float32)
    w[0, 0, 0, 0] = np.inf
    out = np.empty(int(lib.mvlm_pack_fast_weights16(p(w), 64, 16, 64, 16, None, None)), np.uint16)
    assert lib.mvlm_pack_fast_weights16(p(w), 64, 16, 64, 16, out.ctypes.data_as(C.POINTER(C.c_uint16)), None) == 0
