"""Why do fast16 landmarks of bu3dfe / 12 views differ from the oracle estimator fed with the same maxima?"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import face_like_mesh
from oracle import estimator as oest, surface

import contextlib, io
from mvlm_amd import arch, weights
from oracle import pipeline as opipe
for prec, name, mode, nv in (("fast16", "dtu3d", "RGB", 16), ("fast16", "bu3dfe", "RGB+depth", 12), ("exact", "bu3dfe", "RGB+depth", 12)):
    pipe = pipeline.create_pipeline(name, n_views=nv, weights="synthetic:11", verbose=False, image_mode=mode, precision=prec)
    mesh = face_like_mesh(60, 128, 11)
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    np.random.seed(1)
    got, _ = pipe.predict_mesh_device(mesh, poses)
    inner = pipe._buffers["maxima"].cpu().numpy().copy()
    gmax = pipe.predictor_2d.predict_device(pipe.renderer_3d.render_device(mesh, poses)).cpu().numpy()
    nl = pipe.get_lm_count()
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        want, _, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses,
                                            weights.synthetic_state_dict(nl, arch.IMAGE_CHANNELS[mode], seed=11), arch.CHANNEL_SELECT[mode])
    print(prec, name, "pixels differing from the oracle:", int((~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)).sum()))
    print(prec, "inner maxima == second pass maxima:", np.array_equal(inner, gmax), "max |d|", np.abs(inner - gmax).max())
    s_, e_ = oest.estimate_landmark_lines(256, gmax, poses)
    np.random.seed(1)
    fed, _ = oest.estimate_landmarks_from_lines(gmax, s_, e_)
    fed = surface.project_landmarks_to_surface(mesh.verts, mesh.tris, fed)
    d = np.abs(got - fed).max(axis=1)
    bad = np.where(d > 1e-6)[0]
    v = gmax[:, :, 2]
    q = np.quantile(v, 0.5, axis=1, keepdims=True)
    counts = (v > q).sum(axis=1)
    print(prec, "landmarks off:", bad.tolist())
    print(prec, "survivor counts != nv // 2 at:", np.where(counts != nv // 2)[0].tolist(), counts[counts != nv // 2].tolist())
    e3 = pipe.estimator_3d
    # the product's own slot-protocol estimator on the same maxima
    s2, e2 = e3.estimate_landmark_lines(np.zeros((nv, 256, 256, 4), np.float32), gmax, poses)
    np.random.seed(1)
    out2, _ = e3.estimate_landmarks_from_lines(gmax, s2, e2)
    out2 = e3.project_landmarks_to_surface(mesh, out2)
    print(prec, "slot estimator vs oracle estimator max |d|:", np.abs(out2 - fed).max(), " fused vs slot:", np.abs(got - out2).max())
