"""Device and host memory over 1 500 predict_one_file calls on RGB scans (texture decoded ahead on the device, mesh pool)."""
import resource, sys, tempfile, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import write_face_like_obj
d = Path(tempfile.mkdtemp())
files = [write_face_like_obj(d / f"s{i}.obj", grid=61 + 10 * i, tex_size=512 * (1 + i % 3), seed=i) for i in range(4)]
pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="RGB", verbose=False)
def free_mb(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
def rss_mb(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for i in range(40): pipe.predict_one_file(files[i % 4])
f0, r0, t0 = free_mb(), rss_mb(), time.perf_counter()
for rnd in range(5):
    for i in range(300): assert pipe.predict_one_file(files[i % 4]) is not None
    print(f"after {300 * (rnd + 1)} calls: device free {free_mb():.0f} MB (start {f0:.0f}), host max rss {rss_mb():.0f} MB (start {r0:.0f}), {1e3 * (time.perf_counter() - t0) / (300 * (rnd + 1)):.2f} ms per call", flush=True)
