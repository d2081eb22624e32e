"""Oracle: CPU multi-view renderer, ctypes wrapper around oracle/raster.c.
TEST INFRASTRUCTURE - see oracle/__init__.py.  PARITY UNPINNED against VTK (absent); held against a real OpenGL's
rendering of the reference's GL work (tests/golden/gl_raster.npz, tests/test_gl_contract.py)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from .estimator import view_rotation

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "liboracle_raster.so"
_lib = None


def build() -> Path:
    r = subprocess.run(["make", "-C", str(_HERE)], capture_output=True, text=True)
    if r.returncode != 0 or not _SO.exists():
        raise RuntimeError(f"oracle build failed:\n{r.stdout}\n{r.stderr}")
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        _lib = C.CDLL(str(_SO))
        _lib.oracle_render.restype = C.c_int
        _lib.oracle_render_bits.restype = C.c_int
    return _lib


def multiview_render(verts, tris, uvs, texture, transform_stack, shading: str = "texture", subpixel_bits: int = 8) -> np.ndarray:
    """-> image_stack [N,256,256,4] float32 in [0,1] (render3d.py:179-193 output).  `subpixel_bits`: see oracle_render_bits
    (8 = the contract; 4 only for the comparison with the software OpenGL of tests/golden/gl_raster.npz)."""
    lib = _load()
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.int32)
    n = int(np.asarray(transform_stack).shape[0])
    rot = np.ascontiguousarray(np.stack([view_rotation(*transform_stack[i, :3]).ravel() for i in range(n)]), np.float64)
    out = np.empty((n, 256, 256, 4), np.float32)
    use_tex = uvs is not None and texture is not None
    uv = np.ascontiguousarray(uvs, np.float32) if uvs is not None else None
    tex = np.ascontiguousarray(texture, np.uint8) if use_tex else None
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t)) if a is not None else None
    rc = lib.oracle_render_bits(p(verts, C.c_float), p(uv, C.c_float), C.c_int(verts.shape[0]), p(tris, C.c_int32),
                           C.c_int(tris.shape[0]), p(tex, C.c_uint8), C.c_int(tex.shape[0] if use_tex else 0),
                           C.c_int(tex.shape[1] if use_tex else 0), p(rot, C.c_double), C.c_int(n),
                           C.c_int(1 if shading == "geometry" else 0), C.c_int(subpixel_bits), p(out, C.c_float))
    if rc != 0:
        raise MemoryError("oracle_render failed")
    return out
