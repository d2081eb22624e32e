import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, "/root/repo")
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
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    np.random.seed(1)
    pipe.predict_mesh_device(mesh, poses)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{t:.2f}" for t in ts))
