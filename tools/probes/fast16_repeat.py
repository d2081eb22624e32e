"""Is a fast16 pass deterministic?  Two passes over the same views, maxima and heatmaps compared (also exact / fast)."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import face_like_mesh

for name, mode, n in (("bu3dfe", "RGB+depth", 12), ("dtu3d", "RGB", 16)):
    pipe = pipeline.create_pipeline(name, n_views=n, weights="synthetic:11", verbose=False, image_mode=mode)
    mesh = face_like_mesh(60, 128, 11)
    np.random.seed(0)
    poses = pipe.renderer_3d.generate_3d_transformations()
    images = pipe.renderer_3d.render_device(mesh, poses)
    p2 = pipe.predictor_2d
    for prec in ("exact", "fast", "fast16"):
        p2.set_precision(prec)
        p2.set_execution(graphs=False)
        m = [p2.predict_device(images).clone() for _ in range(4)]
        h = [p2.heatmaps_device(images[:4]).clone() for _ in range(3)]
        print(name, prec, "maxima equal:", [bool(torch.equal(m[0], x)) for x in m[1:]], "heat equal:", [bool(torch.equal(h[0], x)) for x in h[1:]],
              "max |dheat|", float(max((h[0] - x).abs().max() for x in h[1:])), "nan", bool(torch.isnan(m[0]).any()), flush=True)
        if prec == "fast16":
            d = (m[0] != m[1]).any(dim=2)
            print("   planes differing between pass 0 and 1:", int(d.sum()), "of", d.numel())
