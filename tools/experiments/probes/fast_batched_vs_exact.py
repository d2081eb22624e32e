#!/usr/bin/env python3
"""Opt-in modes against the default path on the same seeded RNG (run on the GPU box): precision="fast" and
predict_files(batch_scans=4), separately and together, 6 scans at 8 views - how many landmarks stay bit-identical."""
import sys, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from mvlm_amd import pipeline
from mvlm_amd.utils.synthetic import write_face_like_obj
d = Path(tempfile.mkdtemp())
files = [write_face_like_obj(d / f"s{i}.obj", grid=40 + 3 * i, tex_size=64, seed=i) for i in range(6)]
def run(precision, bs):
    pipe = pipeline.create_pipeline("bu3dfe", n_views=8, weights="synthetic:1", verbose=False, precision=precision)
    np.random.seed(5)
    return [lm for _, lm in pipe.predict_files(files, batch_scans=bs)]
base = run("exact", 1)
for precision, bs in (("exact", 4), ("fast", 1), ("fast", 4)):
    got = run(precision, bs)
    same = [int(np.all(a == b, axis=1).sum()) for a, b in zip(base, got)]
    print(f"precision {precision:5s} batch_scans {bs}: identical landmarks per scan {same} of 84; max |diff| {max(float(np.abs(a - b).max()) for a, b in zip(base, got)):.3f}")
