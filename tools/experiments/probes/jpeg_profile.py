"""Decode the bench's 2048 x 2048 texture ten times (run under rocprofv3 --kernel-trace --stats)."""
import ctypes as C, sys, tempfile, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from mvlm_amd import _lib
from mvlm_amd.utils.synthetic import write_face_like_obj
ctx = _lib.Context(0); lib = ctx.lib
d = Path(tempfile.mkdtemp())
obj = write_face_like_obj(d / "face.obj", grid=64, tex_size=2048, seed=0)
raw = np.frombuffer(obj.with_suffix(".jpg").read_bytes(), np.uint8)
out = torch.empty((2048, 2048, 3), dtype=torch.uint8, device="cuda")
rounds = C.c_int()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    rc = lib.mvlm_jpeg_decode(ctx.handle, _lib.as_ptr(raw, C.c_uint8), raw.size, C.c_void_p(out.data_ptr()), C.byref(rounds))
    ts.append(time.perf_counter() - t0)
print("rc", rc, "bytes", raw.size, "rounds", rounds.value, "ms", [round(1e3 * t, 3) for t in ts])
