"""Round-3 GPU tests: the config files' pre-align block through the live paths (fused and slot protocol) against
the oracle, write_renderings / write_pre_aligned, one pipeline shared by threads, the predictor's ``n_gpus``
replicas, and ``bench.py --gpus N`` started exactly as the driver starts it.  Run with -m gpu."""
import contextlib
import io
import json
import os
import subprocess
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]


def _raw_scan(tmp_path, block, grid=40, seed=5, name="scan.obj"):
    from mvlm_amd.utils.mesh_io import write_obj
    from mvlm_amd.utils.synthetic import face_like_mesh, unaligned_copy

    raw = unaligned_copy(face_like_mesh(grid, 64, seed), block)
    path = tmp_path / name
    write_obj(path, raw.verts, raw.tris, raw.uvs, raw.texture)
    return path


@pytest.mark.parametrize("stem,n_views", [("BU_3DFE-depth", 8), ("DTU3D-RGB_BU3DFE_RAW", 12), ("DTU3D-depth-MRI", 8)])
def test_pre_align_block_through_predict_one_file(tmp_path, stem, n_views):
    """An off-centre, small or turned scan + the pre-align block of the reference's config file
    (utils3d.py:465-527): predict_one_file renders the ALIGNED mesh and returns landmarks in the FILE's coordinates.
    Oracle: the reference's transform sequence on the same points, the whole CPU path, the inverse transform.
    The slot protocol (general_pipeline.py:83-108) gives the same result as the fused path."""
    from mvlm_amd import arch, config, weights
    from mvlm_amd.pipeline import pipeline_from_config
    from mvlm_amd.utils.mesh_io import load_obj
    from oracle import pipeline as opipe
    from oracle import prealign as opre

    cfg = config.default_config(stem, n_views=n_views)
    block = cfg["pre-align"]
    cfg["process_3d"]["write_renderings"] = False
    mode = cfg["arch"]["args"]["image_channels"]
    path = _raw_scan(tmp_path, block)
    pipe = pipeline_from_config(cfg, weights="synthetic:9", verbose=False)
    pipe.write_pre_aligned_folder = tmp_path
    np.random.seed(4)
    got = pipe.predict_one_file(path)
    np.random.seed(4)
    slots = pipe._predict_slots(path)
    np.testing.assert_array_equal(got, slots)

    raw = load_obj(path)
    nl, c = pipe.get_lm_count(), arch.IMAGE_CHANNELS[mode]
    verts, t = opre.pre_transformation(raw.verts, block)
    np.random.seed(4)
    poses = pipe.renderer_3d.generate_3d_transformations()
    # a depth model's pipeline does not decode the JPEG (nothing reads the colour planes): the oracle renders untextured too
    texture = raw.texture if pipe._texture_needed() else None
    assert pipe._texture_needed() == (mode != "depth")
    with contextlib.redirect_stdout(io.StringIO()):
        want, _, inter = opipe.predict_mesh(verts, raw.tris, raw.uvs, texture, poses,
                                            weights.synthetic_state_dict(nl, c, seed=9), arch.CHANNEL_SELECT[mode])
    want = opre.landmarks_to_original_space(want, t)
    mesh = pipe.renderer_3d.load_mesh(path, load_texture=pipe._texture_needed())
    assert mesh.to_original is not None
    images = pipe.renderer_3d.render_device(mesh, poses)
    np.testing.assert_array_equal(images.cpu().numpy(), inter["images"])
    assert (inter["images"][..., 3] < 1.0).mean() > 0.1          # the aligned scan really is in the view box
    gmax = pipe.predictor_2d.predict_device(images).cpu().numpy()
    same = np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=(1, 2))
    assert same.mean() > 0.95
    unit = 1.0 / float(block["scale"])
    assert np.abs(got[same] - want[same]).max() < 1e-3 * unit
    # the landmarks lie on the RAW surface (file coordinates), not on the aligned one
    lo, hi = raw.verts.min(0) - 1e-3, raw.verts.max(0) + 1e-3
    assert np.all(got >= lo) and np.all(got <= hi)
    if block["write_pre_aligned"]:
        out = tmp_path / "scan_pre_transform_mesh.vtk"
        assert out.exists() and "POLYDATA" in out.read_text()[:200]
        from mvlm_amd.utils.mesh_io import load_mesh

        back = load_mesh(out)
        np.testing.assert_allclose(back.verts, mesh.verts, rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(back.tris, mesh.tris)


def test_predict_files_applies_pre_align_once(tmp_path):
    """predict_files (reader threads, batched scans) honours the block too and equals the one-by-one loop."""
    from mvlm_amd import config
    from mvlm_amd.pipeline import pipeline_from_config

    cfg = config.default_config("BU_3DFE-depth", n_views=8)
    files = [_raw_scan(tmp_path, cfg["pre-align"], seed=s, name=f"scan{s}.obj") for s in (1, 2, 3)]
    pipe = pipeline_from_config(cfg, weights="synthetic:9", verbose=False)
    np.random.seed(2)
    loop = [pipe.predict_one_file(f) for f in files]
    np.random.seed(2)
    piped = [lm for _, lm in pipe.predict_files(files)]
    np.random.seed(2)
    batched = [lm for _, lm in pipe.predict_files(files, batch_scans=3)]
    for a, b, c in zip(loop, piped, batched):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, c, rtol=0, atol=1e-9)
        assert np.abs(a).max() < 12.0                      # file coordinates: a scan of +-5 units around (3, -2, 1.5)


def test_depth_models_skip_the_texture_decode(tmp_path, monkeypatch):
    """A depth (or geometry) model never reads the colour planes: its pipeline asks load_mesh to leave the JPEG alone (17 of
    20 ms of a 2048x2048-texture scan's ingest) and the landmarks are those of the run that decoded it.  Writing the views
    out (render_image_stack) or an RGB model brings the texture back.  The choice is an argument of each load (round 4), not
    renderer state: the renderer's own entry points always load the texture, whatever pipeline used the renderer before."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils import render3d
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=40, tex_size=64, seed=1)
    asked = []  # per load: did anything touch the texture file (libjpeg at load time, or - round 5 - the device decoder)?
    real_load, real_ahead = render3d.load_obj, render3d.decode_texture_ahead
    touched = []

    def spy(path, load_texture=True, **kw):
        mesh = real_load(path, load_texture=load_texture, **kw)
        touched.append(bool(load_texture) and (mesh.texture_jpeg is not None or mesh._texture is not None))
        return mesh

    def spy_ahead(ctx, data):
        touched.append(True)
        return real_ahead(ctx, data)

    real_load_mesh = render3d.HipRenderer3D.load_mesh

    def spy_load_mesh(self, file_name, load_texture=True):
        touched.clear()
        mesh = real_load_mesh(self, file_name, load_texture=load_texture)
        asked.append(any(touched))
        assert asked[-1] == bool(load_texture)
        return mesh

    monkeypatch.setattr(render3d, "load_obj", spy)
    monkeypatch.setattr(render3d, "decode_texture_ahead", spy_ahead)
    monkeypatch.setattr(render3d.HipRenderer3D, "load_mesh", spy_load_mesh)
    pipe = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:3", image_mode="depth", verbose=False)
    np.random.seed(5)
    lean = pipe.predict_one_file(obj)
    assert pipe._texture_needed() is False and asked == [False]
    # direct use of the same renderer afterwards: texture loaded (the reference's behaviour, utils3d.py:26-36)
    assert pipe.renderer_3d.load_mesh(obj).texture is not None
    asked.clear()
    stack, _, handle = pipe.renderer_3d.multiview_render(obj)
    assert asked == [True]
    assert handle.texture is not None and len(np.unique(stack[..., :3])) > 2
    asked.clear()
    pipe.render_image_stack, pipe.render_image_folder = True, tmp_path
    np.random.seed(5)
    full = pipe.predict_one_file(obj)
    assert pipe._texture_needed() is True and asked == [True]
    np.testing.assert_array_equal(lean, full)
    np.random.seed(5)
    np.testing.assert_array_equal(pipe._predict_slots(obj), full)     # the slot protocol takes the same decision per call
    rgb = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:3", image_mode="RGB+depth", verbose=False)
    assert rgb.predict_one_file(obj) is not None and rgb._texture_needed() is True
    geo = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="geometry+depth", verbose=False)
    geo.renderer_3d.shading = "geometry"
    asked.clear()
    assert geo.predict_one_file(obj) is not None and geo._texture_needed() is False and asked == [False]
    asked.clear()
    assert [lm is not None for _, lm in geo.predict_files([obj, obj])] == [True, True] and asked == [False, False]


@pytest.mark.parametrize("name,mode,n_views", [("dtu3d", "RGB", 16), ("bu3dfe", "RGB+depth", 12)])
def test_fast_precision_against_the_oracle(name, mode, n_views):
    """precision="fast" (opt-in bf16x3 arithmetic, never the default, never bench.py's value) against the CPU ORACLE, not
    just against the exact path: the same render, at most 0.1 % of the argmax planes move (near-ties; measured at full size:
    1 of 8 064 and 0 of 4 672, profiles/r03_fast_vs_oracle_*), and every landmark whose views all picked the oracle's
    pixel lands within 1e-3 model units."""
    from mvlm_amd import arch, pipeline, weights
    from mvlm_amd.utils.synthetic import face_like_mesh
    from oracle import pipeline as opipe

    pipe = pipeline.create_pipeline(name, n_views=n_views, weights="synthetic:11", verbose=False, image_mode=mode,
                                    precision="fast")
    assert pipe.predictor_2d.precision == "fast"
    mesh = face_like_mesh(60, 128, 11)
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    np.random.seed(1)
    got, _ = pipe.predict_mesh_device(mesh, poses)
    gmax = pipe.predictor_2d.predict_device(pipe.renderer_3d.render_device(mesh, poses)).cpu().numpy()
    nl = pipe.get_lm_count()
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        want, _, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses,
                                            weights.synthetic_state_dict(nl, arch.IMAGE_CHANNELS[mode], seed=11),
                                            arch.CHANNEL_SELECT[mode])
    diff = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)
    assert diff.mean() <= 0.001, f"{int(diff.sum())} of {diff.size} argmax planes differ from the oracle"
    same = ~diff.any(axis=1)
    assert same.mean() > 0.95
    assert np.abs(got[same] - want[same]).max() < 1e-3
    scores = np.abs(gmax[:, :, 2] - inter["maxima"][:, :, 2])[~diff]
    assert scores.max() < 1e-4 * max(1.0, np.abs(inter["maxima"][:, :, 2]).max())


def test_cli_runs_a_config_file(tmp_path):
    """`python -m mvlm_amd -p folder -c BU_3DFE-depth -n 8`: one pipeline built from the config (here by the stem of the
    reference's file; a path to a JSON works the same), pre-align block included - the landmark files hold what
    pipeline_from_config + predict_one_file give."""
    from mvlm_amd import config
    from mvlm_amd.__main__ import main
    from mvlm_amd.pipeline import pipeline_from_config

    cfg = config.default_config("BU_3DFE-depth", n_views=8)
    files = [_raw_scan(tmp_path, cfg["pre-align"], seed=s, name=f"scan{s}.obj") for s in (1, 2)]
    out = tmp_path / "out"
    assert main(["-p", str(tmp_path), "-o", str(out), "-c", "BU_3DFE-depth", "-n", "8", "--weights", "synthetic:9", "--seed", "3"]) == 0
    pipe = pipeline_from_config(cfg, weights="synthetic:9", verbose=False)
    np.random.seed(3)
    for f in files:
        want = pipe.predict_one_file(f)
        got = np.loadtxt(out / f"{f.stem}_BU_3DFE-depth.txt", delimiter=",")
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    cfg_path = tmp_path / "my.json"
    cfg_path.write_text(json.dumps(config.default_config("DTU3D-RGB", n_views=8)))
    assert main(["-p", str(files[0]), "-o", str(out), "-c", str(cfg_path), "--weights", "synthetic:9"]) == 0
    assert np.loadtxt(out / "scan1_my.txt", delimiter=",").shape == (73, 3)


def test_write_renderings_key_dumps_the_views(tmp_path):
    """process_3d.write_renderings (configs/DTU3D-RGB_Artec3D.json ...) -> general_pipeline.py:133-146's PNG dump."""
    from PIL import Image

    from mvlm_amd import config
    from mvlm_amd.pipeline import pipeline_from_config

    cfg = config.default_config("DTU3D-RGB_Artec3D", n_views=8)
    assert cfg["process_3d"]["write_renderings"] is True
    path = _raw_scan(tmp_path, cfg["pre-align"])
    pipe = pipeline_from_config(cfg, weights="synthetic:9", verbose=False)
    assert pipe.render_image_stack
    assert pipe.predict_one_file(path) is not None
    pngs = sorted(tmp_path.glob("scan_*.png"))
    assert len(pngs) == 8
    views = pipe.renderer_3d.render_device(pipe.renderer_3d.load_mesh(path), pipe.renderer_3d.generate_3d_transformations())
    first = np.asarray(Image.open(pngs[0]))
    np.testing.assert_array_equal(first, np.uint8(views[0, :, :, :3].cpu().numpy() * 255))
    assert (tmp_path / "scan_pre_transform_mesh.vtk").exists()   # that file also sets write_pre_aligned


def test_one_pipeline_shared_by_two_threads(tmp_path):
    """3DMD_server.py:26-31 calls predict_one_file from a thread pool: two threads x different scans x ONE pipeline
    give, bit for bit, what the sequential calls give (calls are serialised per pipeline; the global RNG is
    consumed call by call, so the threaded run equals one of the two sequential orders)."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    a = write_face_like_obj(tmp_path / "a.obj", grid=40, tex_size=64, seed=1)
    b = write_face_like_obj(tmp_path / "b.obj", grid=56, tex_size=64, seed=2)
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", verbose=False)  # the fixed 8-view table: no RNG in the poses

    def sequential(order):
        np.random.seed(7)
        return {f.name: pipe.predict_one_file(f) for f in order}

    ab, ba = sequential([a, b]), sequential([b, a])
    assert not np.array_equal(ab["a.obj"], ab["b.obj"])
    for _ in range(4):
        out, errors = {}, []
        start = threading.Barrier(2)

        def work(f):
            try:
                start.wait()
                out[f.name] = pipe.predict_one_file(f)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        np.random.seed(7)
        threads = [threading.Thread(target=work, args=(f,)) for f in (a, b)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        # the RANSAC draws come from the global RNG in the order the lock admitted the two calls
        matches = [all(np.array_equal(out[k], ref[k]) for k in ("a.obj", "b.obj")) for ref in (ab, ba)]
        assert any(matches)


def test_threads_with_planted_peaks_match_one_sequential_order():
    """The same on the RANSAC inlier branch (planted peaks: draw, inlier count, refit, snap all run): the threaded
    calls give the sequential calls' results, in whatever order the lock admitted them."""
    from mvlm_amd import config
    from mvlm_amd.pipeline import pipeline_from_config
    from test_planted_cpu import planted_scene

    mesh, _, sd, poses = planted_scene(n_views=16)
    pipe = pipeline_from_config(config.default_config("DTU3D", "RGB", n_views=16), weights=sd, verbose=False)

    def run_sequential(k):
        np.random.seed(21)
        return [pipe.predict_mesh_device(mesh, poses)[0] for _ in range(k)]

    seq = run_sequential(4)   # (the inlier refit uses every inlier, so the four may well be equal: then all calls must give that)
    results, lock = {}, threading.Lock()

    def work(tag):
        for i in range(2):
            lm, _ = pipe.predict_mesh_device(mesh, poses)
            with lock:
                results[(tag, i)] = lm

    # order of admission = order of RNG consumption: the four results are the four sequential ones in some order
    np.random.seed(21)
    threads = [threading.Thread(target=work, args=(t,)) for t in ("x", "y")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    got = list(results.values())
    assert len(got) == 4
    # a permutation of the sequential results, bit for bit: every threaded result IS a sequential one, every sequential one
    # occurs, and equal results occur equally often (the two multisets are the same)
    from collections import Counter

    key = lambda a: np.ascontiguousarray(a).tobytes()
    assert Counter(key(g) for g in got) == Counter(key(s) for s in seq)


def test_n_gpus_replicas_match_single_device(monkeypatch):
    """``n_gpus`` > 1 (paulsenpredictor.py:100-105, nn.DataParallel): replicas of the weights, views split
    contiguously, maxima gathered in view order.  Rehearsed on this one-GPU box with both replicas on device 0
    (MVLM_REPLICA_DEVICES); uneven splits and more devices than views included."""
    import torch

    from conftest import seeded_images
    from mvlm_amd.prediction import DTU3DPredictor

    x = torch.from_numpy(seeded_images(3, 7)).cuda()
    single = DTU3DPredictor(image_mode="RGB", weights="synthetic:5", verbose=False)
    want = single.predict_device(x).cpu().numpy()
    monkeypatch.setenv("MVLM_REPLICA_DEVICES", "0,0,0")
    multi = DTU3DPredictor(image_mode="RGB", weights="synthetic:5", verbose=False, n_gpus=3)
    assert len(multi._replicas) == 2
    for n in (7, 3, 2, 1):
        got = multi.predict_device(x[:n]).cpu().numpy()
        np.testing.assert_array_equal(got[:, :, :2], want[:, :n, :2])
        np.testing.assert_allclose(got[:, :, 2], want[:, :n, 2], rtol=1e-5)
    out = torch.empty((73, 7, 3), device="cuda")
    for _ in range(3):                                           # stable buffers: capture + replay on every replica
        multi.predict_device(x, out=out)
    np.testing.assert_array_equal(out.cpu().numpy()[:, :, :2], want[:, :, :2])
    lms, valid = multi.predict_landmarks_from_images(x.cpu().numpy())
    assert valid.all() and np.array_equal(lms[:, :, :2], want[:, :, :2])
    monkeypatch.delenv("MVLM_REPLICA_DEVICES")
    clamped = DTU3DPredictor(image_mode="RGB", weights="synthetic:5", verbose=False, n_gpus=8)
    assert len(clamped._replicas) == torch.cuda.device_count() - 1


def _run_bench(args, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    env["MVLM_BENCH_NO_INGEST"] = "1"
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=str(REPO))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` as a plain process (no RANK in the environment - how the driver starts it):
    the parent spawns the two ranks, relays rank 0's JSON line and the exit code.  One GPU here, so the two ranks
    share it over gloo (MVLM_BENCH_SHARE_GPU=1); on a node with >= 2 GPUs the same command runs one rank per GPU
    over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--cpu-views", "0", "--no-fast-mode", "--views-total", "16"],
                       capture_output=True, text=True, timeout=900, cwd=str(REPO),
                       env=dict(env, MVLM_BENCH_SHARE_GPU="1", MVLM_BENCH_NO_INGEST="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                              # ONE JSON line on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert "gloo world_size 2" in rec["config"]["parallelism"]
    assert rec["config"]["views_total"] == 16 and rec["config"]["views_per_gpu"] == "8"
    assert rec["value"] > 0 and rec["unit"] == "views/s" and rec["scaling"] == "strong"
    # round 4: the line explains its own scaling - per-rank step times, the two collectives, the shard run unsharded
    sb = rec["scaling_breakdown"]
    assert len(sb["per_rank_ms_per_step"]["all"]) == 2 and sb["per_rank_ms_per_step"]["max"] == pytest.approx(rec["ms_per_step"], rel=1e-3)
    assert sb["all_gather_ms_per_step"] > 0 and sb["draws_broadcast_ms_per_step"] > 0 and sb["collective_clock"] == "host wall clock"
    assert sb["shard_views"] == 8 and 0 < sb["single_gpu_shard_ms"] < 2 * rec["ms_per_step"]


def test_bench_refuses_more_gpus_than_visible():
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MVLM_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=str(REPO), env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert f"needs {n} visible GPUs" in r.stderr
