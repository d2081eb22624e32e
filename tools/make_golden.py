#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own importable code.

Runs only in the build container (needs /root/reference); never on the GPU box.
The reference cannot be imported as a package here (``import mvlm`` pulls dlib /
mediapipe / vtk, all absent), so the two numeric modules on the hot path are
loaded by file path, exactly as SURVEY.md 8c describes:

  * prediction/predictor2d.py + prediction/paulsenpredictor.py under a synthetic
    parent package (so ``from .predictor2d import ...`` resolves),
  * utils/utils3d.py + utils/estimator3d.py + utils/render3d.py with an inert
    module object named ``vtk`` in sys.modules: those files ``import vtk`` at the
    top but the functions captured here (rays, filters, LSQ, one-shot RANSAC,
    pose table) never call into it.  The renderer and the surface snap DO live in
    vtk and are therefore not captured (parity unpinned).

Only data (inputs / seeds and the reference's outputs) is written.  Weights come
from mvlm_amd.weights.synthetic_state_dict (numpy RandomState over sorted keys),
so tests rebuild them without the reference.

usage: python tools/make_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import types
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
REF = Path("/root/reference/src/mvlm")

from mvlm_amd import arch, weights  # noqa: E402


class _Inert:
    """Attribute sink used only so ``import vtk`` / ``vtk.vtkActor`` annotations evaluate."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Inert

    def __call__(self, *a, **k):
        return _Inert()


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    import matplotlib

    matplotlib.use("Agg")
    for pkg in ("refmvlm", "refmvlm.prediction", "refmvlm.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    vtk = types.ModuleType("vtk")
    vtk.__getattr__ = lambda name: _Inert  # type: ignore[attr-defined]
    vtk_util = types.ModuleType("vtk.util")
    vtk_np = types.ModuleType("vtk.util.numpy_support")
    vtk_np.vtk_to_numpy = _Inert()
    sys.modules.update({"vtk": vtk, "vtk.util": vtk_util, "vtk.util.numpy_support": vtk_np})
    _load("refmvlm.prediction.predictor2d", REF / "prediction/predictor2d.py")
    pp = _load("refmvlm.prediction.paulsenpredictor", REF / "prediction/paulsenpredictor.py")
    u3 = _load("refmvlm.utils.utils3d", REF / "utils/utils3d.py")
    e3 = _load("refmvlm.utils.estimator3d", REF / "utils/estimator3d.py")
    r3 = _load("refmvlm.utils.render3d", REF / "utils/render3d.py")
    return pp, u3, e3, r3


def torch_sd(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def seeded_images(seed: int, n: int) -> np.ndarray:
    """u8-quantised [n,256,256,4] f32 in [0,1], like the renderer's output (render3d.py:191)."""
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(n, 32, 32, 4)).astype(np.float32)
    img = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    img += rs.randint(-20, 21, size=img.shape)
    return (np.clip(img, 0, 255) / 255).astype(np.float32)


MODES = {"RGB": 3, "depth": 1, "RGB+depth": 4, "geometry+depth": 2}


def gen_keys(pp, out: Path):
    table = {}
    for nl, mode in ((73, "RGB"), (84, "RGB+depth"), (73, "geometry+depth"), (84, "depth")):
        m = pp.MVLMModel(n_landmarks=nl, n_features=256, dropout_rate=0.2, image_channels=mode)
        table[f"{nl}_{mode}"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    (out / "state_dict_keys.json").write_text(json.dumps(table, indent=0, sort_keys=True))


def gen_blocks(pp, out: Path):
    """ResidualBlock / HourGlassModule on small inputs with non-trivial BN stats."""
    res = {}
    rs = np.random.RandomState(11)
    for tag, cin, cout, size in (("rb_64_128", 64, 128, 24), ("rb_256_256", 256, 256, 12)):
        sd_full = weights.synthetic_state_dict(73, 3, seed=5)
        prefix = "conv2." if cin == 64 else "hg1.rb1."
        sub = {k[len(prefix):]: v for k, v in sd_full.items() if k.startswith(prefix)}
        blk = pp.ResidualBlock(cin, cout)
        blk.load_state_dict(torch_sd(sub))
        blk.eval()
        x = rs.standard_normal((2, cin, size, size)).astype(np.float32)
        with torch.no_grad():
            y = blk(torch.from_numpy(x)).numpy()
        res[f"{tag}_x"] = x
        res[f"{tag}_y"] = y
    sd_full = weights.synthetic_state_dict(73, 3, seed=5)
    sub = {k[len("hg2."):]: v for k, v in sd_full.items() if k.startswith("hg2.")}
    hg = pp.HourGlassModule(256)
    hg.load_state_dict(torch_sd(sub))
    hg.eval()
    x = rs.standard_normal((1, 256, 32, 32)).astype(np.float32)
    with torch.no_grad():
        y = hg(torch.from_numpy(x)).numpy()
    res["hg_x"] = x
    res["hg_y"] = y
    np.savez_compressed(out / "cnn_blocks.npz", **res)


def gen_full_model(pp, out: Path):
    """Whole MVLMModel through the reference's predict_landmarks_from_images."""
    res = {}
    for nl, mode, seed in ((73, "RGB", 1), (84, "RGB+depth", 2), (73, "geometry+depth", 3), (84, "depth", 4)):
        c = MODES[mode]
        sd = weights.synthetic_state_dict(nl, c, seed=seed)
        model = pp.MVLMModel(n_landmarks=nl, n_features=256, dropout_rate=0.2, image_channels=mode)
        model.load_state_dict(torch_sd(sd))
        model.eval()
        cls = pp.BU3DFEPredictor if nl == 84 else pp.DTU3DPredictor
        pred = cls.__new__(cls)  # bypass __init__'s URL download (:101)
        pred.model, pred.device = model, torch.device("cpu")
        pred.batch_size, pred.selection_method = 2, "simple"
        imgs = seeded_images(100 + seed, 2)
        # the reference feeds all planes of the stack to the net; give it the
        # planes the model was built for (MVLMModel.in_channels)
        sel = list(arch.CHANNEL_SELECT[mode])
        lms, valid = pred.predict_landmarks_from_images(np.ascontiguousarray(imgs[..., sel]))
        with torch.no_grad():
            heat = model(torch.from_numpy(np.ascontiguousarray(imgs[..., sel])).permute(0, 3, 1, 2))[-1]
        flat = heat.reshape(2, nl, -1)
        top2 = torch.topk(flat, 2, dim=2).values.numpy()
        tag = f"{nl}_{mode}"
        res[f"{tag}_seed"] = np.array([seed, 100 + seed])
        res[f"{tag}_maxima"] = lms
        res[f"{tag}_valid"] = valid
        res[f"{tag}_top2"] = top2
        res[f"{tag}_heat_sub"] = heat.numpy()[:, :, 5::16, 3::16].copy()
        print(tag, "heat range", float(heat.min()), float(heat.max()), "min top2 gap", float((top2[..., 0] - top2[..., 1]).min()))
    np.savez_compressed(out / "cnn_full.npz", **res)


def gen_maxima(pp, out: Path):
    rs = np.random.RandomState(21)
    hm = rs.standard_normal((12, 64, 64)).astype(np.float32)
    hm[0, 10, 20] = hm[0, 30, 5] = 9.0          # tie -> first in row-major order
    hm[1, 0, 0] = 9.0                            # corner
    hm[2, 63, 63] = 9.0
    hm[3, 15, 40] = 9.0                          # exactly sz from the border: no moment refinement
    hm[4, 16, 40] = 9.0                          # first refined row
    hm[5, 47, 16] = 9.0                          # hm_size - px = 17 > sz
    hm[6, 48, 30] = 9.0                          # hm_size - px = 16 > sz
    hm[7, 49, 30] = 9.0                          # hm_size - px = 15: not refined
    yy, xx = np.mgrid[0:64, 0:64]
    hm[8] = np.exp(-((yy - 30.3) ** 2 + (xx - 25.8) ** 2) / 18.0).astype(np.float32)
    hm[9] = hm[9] * 0 + 1.5                      # constant plane: argmax = 0
    p = pp.BU3DFEPredictor.__new__(pp.BU3DFEPredictor)
    res = {"heatmaps": hm}
    for method in ("simple", "moment"):
        p.selection_method = method
        res[f"out_{method}"] = p.find_heat_map_maxima(hm)
    np.savez_compressed(out / "maxima.npz", **res)


def gen_estimator(u3, e3, r3, out: Path):
    res = {}
    est = e3.Estimator3D()
    # --- pose tables (render3d.py:79-112) ---
    for n in (8, 64, 96, 128):
        r = r3.ObjVTKRenderer3D.__new__(r3.ObjVTKRenderer3D)
        r.__dict__.update(dict(n_views=n, min_x_angle=-40, max_x_angle=40, min_y_angle=-80, max_y_angle=80,
                               min_z_angle=-20, max_z_angle=20, min_scale=1.4, max_scale=1.9,
                               min_tx=-20, max_tx=20, min_ty=-20, max_ty=20))
        np.random.seed(0)
        res[f"poses_{n}"] = r.generate_3d_transformations()
    # --- rays (estimator3d.py:31-90) ---
    rs = np.random.RandomState(31)
    for tag, n, nl in (("8", 8, 10), ("64", 64, 6)):
        poses = res[f"poses_{n}"]
        lms = np.empty((nl, n, 3), np.float32)
        lms[:, :, 0] = rs.randint(0, 256, (nl, n)) - 1
        lms[:, :, 1] = rs.randint(0, 256, (nl, n)) - 0.5
        lms[:, :, 2] = rs.rand(nl, n)
        if tag == "64":  # sub-pixel ("moment"-style) coordinates too
            lms[:3, :, :2] += rs.rand(3, n, 2).astype(np.float32)
        s, e = est.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), lms, poses)
        res[f"lines_{tag}_lms"], res[f"lines_{tag}_s"], res[f"lines_{tag}_e"] = lms, s, e
    # --- LSQ intersection (utils3d.py:99-124) ---
    target = np.array([12.5, -30.0, 41.0])
    dirs = rs.standard_normal((6, 3))
    pa = target + 200 * dirs + rs.standard_normal((6, 3)) * 0.5
    pb = target - 200 * dirs
    cases = {
        "k0": (np.zeros((0, 3)), np.zeros((0, 3))),
        "k1": (pa[:1], pb[:1]),
        "k2": (pa[:2], pb[:2]),
        "k6": (pa, pb),
        "parallel": (np.array([[0.0, 0, 0], [10, 0, 0], [0, 7, 0]]), np.array([[0.0, 0, 100], [10, 0, 100], [0, 7, 100]])),
    }
    for tag, (a, b) in cases.items():
        res[f"lsq_{tag}_pa"], res[f"lsq_{tag}_pb"] = a, b
        with np.errstate(all="ignore"):
            res[f"lsq_{tag}_p"] = u3.compute_intersection_between_lines(a, b)
    # --- per-landmark consensus (estimator3d.py:158-183) ---
    def scene(n, nl, noise_px, outlier_frac, score_mode):
        poses = res[f"poses_{n}"]
        pts = rs.uniform(-60, 60, (nl, 3))
        lms = np.empty((nl, n, 3), np.float32)
        for v in range(n):
            rx, ry, rz = poses[v, :3]
            M = (e3.rotation_matrix_y(np.deg2rad(ry)) @ e3.rotation_matrix_x(np.deg2rad(rx))) @ e3.rotation_matrix_z(np.deg2rad(rz))
            q = pts @ M.T
            col = (q[:, 0] + 150) / 300 * 256 + rs.standard_normal(nl) * noise_px
            row = 255 - (q[:, 1] + 150) / 300 * 256 + rs.standard_normal(nl) * noise_px
            bad = rs.rand(nl) < outlier_frac
            col[bad] = rs.uniform(0, 256, bad.sum())
            row[bad] = rs.uniform(0, 256, bad.sum())
            lms[:, v, 0], lms[:, v, 1] = row, col
            lms[:, v, 2] = rs.rand(nl) if score_mode == "rand" else np.where(bad, 0.05, 0.9) + rs.rand(nl) * 0.05
        return lms, poses

    scenes = {
        "q64": (scene(64, 12, 0.4, 0.1, "rand"), "quantile", 0.5, 0.5),
        "q8": (scene(8, 9, 0.3, 0.0, "rand"), "quantile", 0.5, 0.5),
        "qfail": (scene(96, 7, 0.3, 0.9, "rand"), "quantile", 0.5, 0.5),     # mostly outliers -> 1e8 branch
        "abs": (scene(64, 10, 0.4, 0.2, "good"), "absolute", 0.5, 0.5),
        "absfew": (scene(8, 8, 0.3, 0.8, "good"), "absolute", 0.5, 0.5),       # k < 3 branch
        "q128x478": (scene(128, 478, 0.5, 0.1, "rand"), "quantile", 0.5, 0.5),  # mediapipe-sized stress
    }
    for tag, ((lms, poses), mode, q, thr) in scenes.items():
        est = e3.Estimator3D(mode=mode, threshold_quantile=q, threshold_absolute=thr)
        s, e = est.estimate_landmark_lines(np.zeros((lms.shape[1], 256, 256, 4), np.float32), lms, poses)
        log = []
        orig = np.random.choice

        def rec(a, size=None, replace=True, p=None, _o=orig, _log=log):
            r = _o(a, size, replace, p)
            _log.append((len(a), np.array(r)))
            return r

        np.random.seed(1)
        np.random.choice = rec
        try:
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                out_lm, err = est.estimate_landmarks_from_lines(lms, s, e)
        finally:
            np.random.choice = orig
        res[f"fuse_{tag}_lms"] = lms
        res[f"fuse_{tag}_poses"] = poses
        res[f"fuse_{tag}_cfg"] = np.array([{"quantile": 0, "absolute": 1}[mode], q, thr])
        res[f"fuse_{tag}_out"] = out_lm
        res[f"fuse_{tag}_err"] = np.array(err)
        res[f"fuse_{tag}_draw_k"] = np.array([k for k, _ in log], np.int64)
        res[f"fuse_{tag}_draws"] = np.array([d for _, d in log], np.int64).reshape(len(log), 8)
        print(tag, "err", err, "draws", len(log))
    np.savez_compressed(out / "estimator.npz", **res)


def gen_marker_rays(e3, r3, out: Path):
    """Ties the RENDERER's pose / pixel convention to the reference ESTIMATOR (estimator3d.py:31-90):
    for seeded rotated poses and argmax pixels (row R, col C) the reference's rays through the four
    sub-pixel corners (R-1 +- 0.75, C-0.5 +- 0.75) of a 1.5-pixel square around the maximum's ray.
    A quad spanned by points on those four rays must be rasterised at exactly the pixels the square
    covers (tests/test_oracle_pinning.py): rotation order, y flip and pixel centres are then pinned
    to reference code, not to this repo's own restatement."""
    res = {}
    rs = np.random.RandomState(77)
    r = r3.ObjVTKRenderer3D.__new__(r3.ObjVTKRenderer3D)
    r.__dict__.update(dict(n_views=12, min_x_angle=-40, max_x_angle=40, min_y_angle=-80, max_y_angle=80,
                           min_z_angle=-20, max_z_angle=20, min_scale=1.4, max_scale=1.9,
                           min_tx=-20, max_tx=20, min_ty=-20, max_ty=20))
    np.random.seed(5)
    poses = r.generate_3d_transformations()          # seeded random rotations about all three axes
    poses8 = r3.ObjVTKRenderer3D.__new__(r3.ObjVTKRenderer3D)
    poses8.__dict__.update(dict(n_views=8))
    poses = np.concatenate([poses, poses8.generate_3d_transformations().astype(np.float64)])  # + the fixed 8-view table
    n = poses.shape[0]
    k = 6                                             # markers per view
    rows = rs.randint(40, 216, (k, n))
    cols = rs.randint(40, 216, (k, n))
    corners = np.array([[-0.75, -0.75], [-0.75, 0.75], [0.75, 0.75], [0.75, -0.75]])
    lms = np.empty((k * 4, n, 3), np.float32)
    for c in range(4):
        lms[c::4, :, 0] = rows - 1 + corners[c, 0]    # the maxima convention (row - 1, col - 0.5), paulsenpredictor.py:127
        lms[c::4, :, 1] = cols - 0.5 + corners[c, 1]
    lms[:, :, 2] = 1.0
    est = e3.Estimator3D()
    s, e = est.estimate_landmark_lines(np.zeros((n, 256, 256, 4), np.float32), lms, poses)
    res["poses"], res["rows"], res["cols"] = poses, rows, cols
    res["corner_lms"], res["corner_s"], res["corner_e"] = lms, s, e
    np.savez_compressed(out / "marker_rays.npz", **res)
    print("marker rays", s.shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(REPO / "tests/golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    pp, u3, e3, r3 = load_reference()
    only = set(filter(None, args.only.split(",")))
    want = lambda k: not only or k in only
    if want("keys"):
        gen_keys(pp, out)
    if want("maxima"):
        gen_maxima(pp, out)
    if want("estimator"):
        gen_estimator(u3, e3, r3, out)
    if want("blocks"):
        gen_blocks(pp, out)
    if want("full"):
        gen_full_model(pp, out)
    if want("markers"):
        gen_marker_rays(e3, r3, out)
    print("wrote", sorted(p.name for p in out.iterdir()))


if __name__ == "__main__":
    main()
