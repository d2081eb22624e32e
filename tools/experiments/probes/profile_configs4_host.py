#!/usr/bin/env python3
"""Where does the host spend a configs[4] step (render + 478-landmark fusion + snap, no network)?  cProfile over 300 steps."""
import cProfile, pstats, sys, io, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
sys.argv = ["bench.py"]
import bench
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import face_like_mesh
mesh = face_like_mesh(224, 2048, seed=0)
pipe = pipeline.Pipeline(n_views=128, device=0, verbose=False)
np.random.seed(0)
poses = pipe.renderer_3d.generate_3d_transformations()
pred, state, _ = bench.synthetic_landmark_predictor(mesh, poses, 478, torch.device("cuda", 0))
pipe.predictor_2d = pred
def step():
    np.random.seed(1)
    return pipe.predict_mesh_device(mesh, poses)
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
