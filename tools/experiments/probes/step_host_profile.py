#!/usr/bin/env python3
"""cProfile of the host side of predict_mesh_device at 12 views (run on the GPU box): where the Python time between two
steps' GPU work goes."""
import cProfile, pstats, sys, io
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from mvlm_amd import config
from mvlm_amd.utils.synthetic import face_like_mesh
cfg = config.load_config(config.default_config("DTU3D", "geometry+depth", n_views=12))
pipe = cfg.build_pipeline(weights="synthetic:0", verbose=False)
mesh = face_like_mesh(224, 2048, seed=0)
np.random.seed(0)
poses = pipe.renderer_3d.generate_3d_transformations()
for _ in range(5):
    np.random.seed(1); pipe.predict_mesh_device(mesh, poses)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    np.random.seed(1); pipe.predict_mesh_device(mesh, poses)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
