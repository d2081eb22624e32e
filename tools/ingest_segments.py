#!/usr/bin/env python3
"""Wall-time segments of one scan at the reference's default 8 views (run on the GPU box): parse, upload call, device
pass, post-step, dropping the mesh.  usage: tools/ingest_segments.py [mallopt]   (mallopt: apply the allocator hint by hand
first - Pipeline applies it anyway unless MVLM_HOST_MALLOC_TUNING=0)"""
import sys, time, tempfile
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import write_face_like_obj
from mvlm_amd.utils.mesh_io import load_obj
from mvlm_amd.utils.render3d import upload_mesh
if len(sys.argv) > 1 and sys.argv[1] == "mallopt":
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    print("mallopt mmap", libc.mallopt(-3, 32 << 20), "trim", libc.mallopt(-1, 1 << 30))
d = Path(tempfile.mkdtemp())
obj = write_face_like_obj(d / "face.obj", grid=224, tex_size=2048, seed=0)
pipe = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:0", image_mode="depth", verbose=False)
for _ in range(3):
    pipe.predict_one_file(obj)
ctx = pipe.renderer_3d.ctx
def T(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(8):
    mode = "device" if it < 4 else "host"  # where the JPEG texture is decoded
    t0 = T(); mesh = load_obj(obj, decode=mode); t1 = T()
    upload_mesh(ctx, mesh); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    poses = pipe.renderer_3d.generate_3d_transformations()
    t4 = T(); lm, _ = pipe.predict_mesh_device(mesh, poses); t5 = T()
    pipe._after_prediction(obj, lm); t6 = T()
    del mesh; t7 = T()
    print(f"[jpeg on the {mode}] load {1e3*(t1-t0):.1f}  upload call {1e3*(t2-t1):.1f} (+{1e3*(t3-t2):.1f} until done)  predict_mesh_device {1e3*(t5-t4):.1f}  after {1e3*(t6-t5):.2f}  del {1e3*(t7-t6):.2f}")
t0 = T()
for _ in range(5): pipe.predict_one_file(obj)
print("predict_one_file avg", 1e3*(T()-t0)/5, pipe.timings)
# the parts of load_obj, and the native reader with 1 thread vs its default (chunk-parallel parse)
import os
from mvlm_amd.utils import mesh_io
def best(fn, n=7):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return 1e3 * min(ts)
print(f"native OBJ reader, default threads: {best(lambda: mesh_io._read_obj_native(obj)):.2f} ms")
for n in ("1", "2", "4", "8", "12", "16"):
    os.environ["MVLM_OBJ_THREADS"] = n
    print(f"native OBJ reader, {n} thread(s): {best(lambda: mesh_io._read_obj_native(obj)):.2f} ms")
del os.environ["MVLM_OBJ_THREADS"]
print(f"JPEG decode (2048x2048): {best(lambda: mesh_io._read_texture(obj.with_suffix('.jpg'))):.2f} ms")
print(f"load_obj (both, JPEG on a second thread): {best(lambda: load_obj(obj, decode='host')):.2f} ms")
print(f"load_obj (JPEG bytes only, decoded by the upload): {best(lambda: load_obj(obj)):.2f} ms")
def load_and_upload(mode):
    m = load_obj(obj, decode=mode); upload_mesh(ctx, m); torch.cuda.synchronize()
print(f"load_obj + upload until done, JPEG on the host: {best(lambda: load_and_upload('host')):.2f} ms")
print(f"load_obj + upload until done, JPEG on the device: {best(lambda: load_and_upload('device')):.2f} ms")
def load_ahead_and_upload():
    m = pipe.renderer_3d.load_mesh(obj); upload_mesh(ctx, m); torch.cuda.synchronize()
print(f"HipRenderer3D.load_mesh + upload until done (JPEG decoded on the device by a second thread beside the OBJ parse): {best(load_ahead_and_upload):.2f} ms")
print("texture file:", obj.with_suffix('.jpg').stat().st_size, "bytes")
# ---- the same scan with one of the REFERENCE'S OWN scanner textures (assets/Normal-Emotion/angry_01.jpg, committed as data:
# 3546 x 2282, another encoder's tables, 2.2 MB, no restart markers; utils3d.py:26-36 is its consumer) ----------------------
real = Path(__file__).resolve().parents[1] / "tests" / "golden" / "jpeg" / "real" / "angry_01.jpg"
if real.exists():
    import shutil
    obj2 = d / "real" / "face.obj"
    obj2.parent.mkdir()
    shutil.copyfile(obj, obj2)
    shutil.copyfile(real, obj2.with_suffix(".jpg"))
    rgb = pipeline.create_pipeline("bu3dfe", n_views=96, weights="synthetic:0", image_mode="RGB+depth", verbose=False)
    for _ in range(3):
        rgb.predict_one_file(obj2)
    print(f"real texture ({real.name}, {real.stat().st_size} bytes): libjpeg on one core {best(lambda: mesh_io._read_texture(obj2.with_suffix('.jpg')), 5):.2f} ms")
    def load_up2(mode):
        m = load_obj(obj2, decode=mode); upload_mesh(ctx, m); torch.cuda.synchronize()
    print(f"real texture: load_obj + upload until done, JPEG on the host {best(lambda: load_up2('host'), 5):.2f} ms, on the device {best(lambda: load_up2('device'), 5):.2f} ms")
    def ahead2():
        m = rgb.renderer_3d.load_mesh(obj2); upload_mesh(ctx, m); torch.cuda.synchronize()
    print(f"real texture: HipRenderer3D.load_mesh + upload until done (device decode beside the OBJ parse): {best(ahead2, 5):.2f} ms")
    for mode in ("device", "host"):
        rgb.renderer_3d.texture_decode = mode
        ts = []
        for _ in range(5):
            t = T(); rgb.predict_one_file(obj2); ts.append(T() - t)
        print(f"real texture: predict_one_file from disk, 96 views RGB+depth, JPEG on the {mode}: {1e3 * min(ts):.2f} ms = {96 / min(ts):.1f} views/s with ingest")
