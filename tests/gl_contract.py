"""Comparison of a rasteriser's image stack with tests/golden/gl_raster.npz (test infrastructure, shared by the CPU test of
oracle/raster.c and the -m gpu test of the HIP rasteriser).

gl_raster.npz holds what a real OpenGL implementation (tools/make_gl_golden.py: SwiftShader's OpenGL ES 3.0, headless) draws
when it is given the reference renderer's GL work (render3d.py:53-77, :136-177; utils3d.py:26-64), with the reference's own
read-back post-processing applied.  A disagreement must fall into one of the classes below, each tied to something OpenGL
leaves to the implementation; anything else fails.

  clip     coverage differs, the scene has vertices outside the window: this GL clips triangles geometrically at the window
           and snaps the new vertices, which moves the clipped edge by up to 1/16 pixel (GPUs use a guard band instead)
  texel    same triangle coverage, the colour is that of a texel adjacent to ours: the pixel's (u, v) lies within the GL's
           interpolation error (~1e-4 texel here) of a texel boundary
  depth1   depth byte off by one: 255 z lies within the last bits of z of an integer (float rounding of the interpolation),
           or the GL's float depth differs from the 24-bit value a depth buffer would hold (`d24` equal)
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden" / "gl_raster.npz"


def load():
    g = np.load(GOLDEN)
    meta = json.loads(str(g["meta"]))
    scenes = {}
    for name in g["scenes"]:
        name = str(name)
        sc = {k: g[f"{name}.{k}"] for k in ("verts", "tris", "poses", "image_u8", "z")}
        sc["uvs"] = g[f"{name}.uvs"] if f"{name}.uvs" in g else None
        sc["tex"] = g[f"{name}.tex"] if f"{name}.tex" in g else None
        sc["lattice"] = bool(g[f"{name}.lattice"])
        scenes[name] = sc
    return meta, scenes


def depth_byte(z: np.ndarray) -> np.ndarray:
    """render3d.py:73-77 on a float Z read-back: x (-255), C cast to unsigned char"""
    return (np.trunc(z.astype(np.float64) * -255.0).astype(np.int64) & 255).astype(np.uint8)


def quantise24(z: np.ndarray) -> np.ndarray:
    """the float a Z read-back returns from a 24-bit fixed-point depth buffer holding z"""
    return (np.floor(z.astype(np.float64) * 16777215.0 + 0.5) / 16777215.0).astype(np.float32)


def _adjacent_texel(tex: np.ndarray, a: np.ndarray, b: np.ndarray) -> bool:
    """is some texel of colour a adjacent (8-neighbourhood, GL_REPEAT wrap) to some texel of colour b?"""
    h, w = tex.shape[:2]
    pa = np.argwhere((tex == a).all(-1))
    pb = np.argwhere((tex == b).all(-1))
    for ya, xa in pa:
        dy = np.minimum((pb[:, 0] - ya) % h, (ya - pb[:, 0]) % h)
        dx = np.minimum((pb[:, 1] - xa) % w, (xa - pb[:, 1]) % w)
        if ((dy <= 1) & (dx <= 1)).any():
            return True
    return False


def compare(scene: dict, stack: np.ndarray) -> dict:
    """stack: float32 [N,256,256,4] in [0,1] (the renderer's output) -> counts per class + `unexplained`."""
    got = np.round(stack * 255.0).astype(np.uint8)
    assert np.array_equal(got.astype(np.float32) / np.float32(255), stack), "the stack's values are not k / 255"
    gl = scene["image_u8"]
    assert got.shape == gl.shape
    # background = white with depth byte 1; a covered pixel may be white too, so coverage comes from the depth plane
    cov_got = got[..., 3] != 1
    cov_ref = gl[..., 3] != 1
    out = {"pixels": int(cov_ref.size), "covered": int(cov_ref.sum()), "clip": 0, "texel": 0, "depth1": 0, "d24": 0, "unexplained": 0}
    from oracle.estimator import view_rotation

    cov_diff = cov_got != cov_ref
    both = cov_got & cov_ref
    d = np.abs(got[..., 3].astype(np.int32) - gl[..., 3].astype(np.int32))
    d = np.minimum(d, 256 - d)
    d24 = depth_byte(quantise24(scene["z"]))
    other_triangle = np.zeros_like(both)
    for v in range(gl.shape[0]):
        xy = (scene["verts"].astype(np.float64) @ view_rotation(*scene["poses"][v, :3]).T)[:, :2]
        clipped = bool((np.abs(xy) > 150.0).any())      # something leaves the window: this GL clips it geometrically
        n = int(cov_diff[v].sum())
        out["clip" if clipped else "unexplained"] += n
        for y, x in zip(*np.nonzero(both[v] & (got[v, ..., :3] != gl[v, ..., :3]).any(-1))):
            if scene["tex"] is not None and _adjacent_texel(scene["tex"], gl[v, y, x, :3], got[v, y, x, :3]):
                out["texel"] += 1
            elif clipped and d[v, y, x] > 1:             # another triangle is seen: a clipped edge moved across this centre
                out["clip"] += 1
                other_triangle[v, y, x] = True
            else:
                out["unexplained"] += 1
    depth_diff = both & (d > 0) & ~other_triangle
    out["depth1"] = int((depth_diff & (d == 1)).sum())
    out["d24"] = int((depth_diff & (got[..., 3] == d24)).sum())
    out["unexplained"] += int((depth_diff & (d > 1) & (got[..., 3] != d24)).sum())
    return out
