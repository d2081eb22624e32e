"""Phases of the native OBJ reader on the bench's 6.4 MB file (MVLM_OBJ_TIMING=1)."""
import os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from mvlm_amd.utils.synthetic import write_face_like_obj
from mvlm_amd.utils import mesh_io
d = Path(tempfile.mkdtemp())
obj = write_face_like_obj(d / "face.obj", grid=224, tex_size=64, seed=0)
for _ in range(5): mesh_io._read_obj_native(obj)
os.environ["MVLM_OBJ_TIMING"] = "1"
for _ in range(3):
    t = time.perf_counter(); mesh_io._read_obj_native(obj); print(f"whole call from Python {1e3 * (time.perf_counter() - t):.3f} ms", file=sys.stderr)
