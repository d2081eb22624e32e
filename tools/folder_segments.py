#!/usr/bin/env python3
"""Folder mode at the reference's default 8 views per scan (run on the GPU box): wall time per scan, scans sharing a
network pass (predict_files(batch_scans=k)), and the device step on one / on different uploaded meshes."""
import sys, time, tempfile, shutil
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import write_face_like_obj
d = Path(tempfile.mkdtemp())
first = write_face_like_obj(d / "scan0.obj", grid=224, tex_size=2048, seed=0)
files = [first]
for i in range(1, 33):
    f = d / f"scan{i}.obj"; shutil.copy(first, f); shutil.copy(first.with_suffix(".jpg"), f.with_suffix(".jpg")); files.append(f)
pipe = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:0", image_mode="depth", verbose=False)
for _ in range(3): pipe.predict_one_file(first)
print(pipe.predictor_2d.execution_stats())
t_prev = time.perf_counter(); ts = []
for f, lm in pipe.predict_files(files):
    t = time.perf_counter(); ts.append(1e3 * (t - t_prev)); t_prev = t
print("per scan ms:", " ".join(f"{v:.1f}" for v in ts))
print(pipe.predictor_2d.execution_stats(), pipe.timings)
for bs, rd in ((1, None), (2, None), (4, None), (8, None), (16, None), (1, 8), (4, 8), (8, 8), (4, 2), (4, 6)):
    list(pipe.predict_files(files[:bs * 2], batch_scans=bs))  # capture the graph of this batch size
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_done = sum(1 for _ in pipe.predict_files(files, batch_scans=bs, readers=rd))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch_scans {bs:2d} readers {rd}: {1e3 * dt / n_done:6.2f} ms per scan  ({8 * n_done / dt:7.1f} views/s, ingest included)")
# same mesh repeatedly through predict_mesh_device (the bench's step)
from mvlm_amd.utils.mesh_io import load_obj
mesh = load_obj(first); poses = pipe.renderer_3d.generate_3d_transformations()
for _ in range(3): pipe.predict_mesh_device(mesh, poses)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): pipe.predict_mesh_device(mesh, poses)
torch.cuda.synchronize(); print("same mesh step ms", 1e3 * (time.perf_counter() - t0) / 10)
meshes = [load_obj(f) for f in files[:8]]
for m in meshes: pipe.predict_mesh_device(m, poses)
torch.cuda.synchronize(); t0 = time.perf_counter()
for m in meshes: pipe.predict_mesh_device(m, poses)
torch.cuda.synchronize(); print("different uploaded meshes step ms", 1e3 * (time.perf_counter() - t0) / 8)

for k in (2, 4, 8):
    group = meshes[:k]
    for _ in range(3): pipe.predict_meshes_device(group)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): pipe.predict_meshes_device(group)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"predict_meshes_device, {k} resident scans: {1e3 * dt:.2f} ms per group = {1e3 * dt / k:.2f} ms per scan", {n: round(1e3 * v, 2) for n, v in pipe.timings.items()})
