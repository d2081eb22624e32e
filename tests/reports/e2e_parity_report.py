#!/usr/bin/env python3
"""End-to-end parity report at a BASELINE size: GPU path vs CPU oracle on the same mesh, poses,
weights and RNG seed.  Test infrastructure (imports oracle/); run on the GPU box.
usage: tests/reports/e2e_parity_report.py [n_views] [grid] [dtu3d|bu3dfe] [exact|fast|fast16] [RGB+depth|RGB|...]
(73 / 84 landmarks; "fast" / "fast16" = the opt-in bf16x3 / f16x2 precisions against the SAME oracle)"""
import contextlib
import io
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mvlm_amd import pipeline, weights  # noqa: E402
from mvlm_amd.utils.mesh_io import load_obj  # noqa: E402
from mvlm_amd.utils.synthetic import write_face_like_obj  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 64
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 224
name = sys.argv[3] if len(sys.argv) > 3 else "dtu3d"
precision = sys.argv[4] if len(sys.argv) > 4 else "exact"
mode = sys.argv[5] if len(sys.argv) > 5 else "RGB+depth"
from mvlm_amd import arch  # noqa: E402
nl = {"dtu3d": 73, "bu3dfe": 84}[name]
with tempfile.TemporaryDirectory() as td:
    obj = write_face_like_obj(Path(td) / "face.obj", grid=grid, tex_size=256, seed=11)
    pipe = pipeline.create_pipeline(name, n_views=n_views, weights="synthetic:11", verbose=False, image_mode=mode,
                                    precision=precision)
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    mesh = load_obj(obj)
    np.random.seed(1)
    got, gerr = pipe.predict_mesh_device(mesh, poses)
    gmax = pipe.predictor_2d.predict_device(pipe.renderer_3d.render_device(mesh, poses)).cpu().numpy()
    sd = weights.synthetic_state_dict(nl, arch.IMAGE_CHANNELS[mode], seed=11)
    np.random.seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        want, werr, inter = opipe.predict_mesh(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses, sd, arch.CHANNEL_SELECT[mode])
img_equal = np.array_equal(pipe.renderer_3d.render_device(mesh, poses).cpu().numpy(), inter["images"])
diff = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)
same = ~diff.any(axis=1)
dev = np.linalg.norm(got - want, axis=1)
print(f"{name} ({mode}, {nl} landmarks, precision {precision}), views {n_views}, triangles {mesh.n_tris}: rendered stack bit-identical to the CPU restatement: {img_equal}")
print(f"argmax planes that differ: {int(diff.sum())} of {diff.size} ({100 * diff.mean():.3f} %)")
print(f"landmarks with every view identical: {int(same.sum())} of {same.size}; max deviation among them {dev[same].max():.3e} model units")
if (~same).any():
    print(f"landmarks with a near-tie flip somewhere: {int((~same).sum())}; deviation median {np.median(dev[~same]):.3e}, max {dev[~same].max():.3e}")
print(f"all landmarks: max deviation {dev.max():.3e} model units; mean RANSAC error gpu {gerr:.6f} / oracle {werr:.6f}")
