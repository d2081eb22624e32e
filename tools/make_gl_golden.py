"""Write tests/golden/gl_raster.npz: the reference renderer's OpenGL work, done by a real OpenGL implementation.

BUILD-CONTAINER TOOL.  The reference renders through VTK, and VTK only drives OpenGL: per pose it hands OpenGL pre-rotated
float vertices, one orthographic matrix, an unlit NEAREST/REPEAT texture and a LEQUAL depth test, and reads RGB bytes and float
Z back (render3d.py:53-77, :136-177; utils3d.py:26-64).  VTK itself is absent here, but a conformant OpenGL ES 3.0 is present
(SwiftShader, tools/gl_reference.py).  This script issues those GL calls for a set of scenes, applies the reference's own
post-processing literally (gl_reference.reference_postprocess) and stores inputs + outputs, so that
  * tests/test_gl_contract.py (CPU) holds oracle/raster.c against them, and
  * tests/test_gpu_gl_contract.py (-m gpu) holds the HIP rasteriser against them.
What this pins: pixel centres, fill rule, window mapping, depth mapping and byte conversion, LEQUAL in draw order, near/far
clipping, texel addressing (NEAREST, REPEAT, bottom-up rows), row flip.  What it cannot pin: choices OpenGL leaves to the
implementation and that differ between this one and the GPU a user's VTK would run on - listed in DESIGN.md section 5.1.

    python tools/make_gl_golden.py [--out tests/golden/gl_raster.npz]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tools"))

from gl_reference import GLReference, reference_postprocess  # noqa: E402

EIGHT_VIEWS = np.array([[30, 15, 0], [30, -15, 0], [30, 45, 0], [30, -45, 0], [-30, 15, 0], [-30, -15, 0], [-30, 45, 0],
                        [-30, -45, 0]], np.float64)   # render3d.py:94-111


def rotation(rx, ry, rz) -> np.ndarray:
    """vtkTransform.RotateY(ry); RotateX(rx); RotateZ(rz) (pre-multiply mode, render3d.py:140-144) = Ry Rx Rz."""
    a, b, c = np.deg2rad([rx, ry, rz])
    mx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    my = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    mz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return (my @ mx) @ mz


def px(p):
    """window coordinate in pixels (GL: origin bottom left) -> model coordinate; exact for multiples of 1/16 pixel"""
    return np.asarray(p, np.float64) * 300.0 / 256.0 - 150.0


def colour_texture(h, w, seed):
    """every texel its own colour, so that a wrong texel shows"""
    rs = np.random.RandomState(seed)
    t = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    t[..., 0] = (np.arange(w)[None, :] * 255 // max(w - 1, 1)).astype(np.uint8)
    t[..., 1] = (np.arange(h)[:, None] * 255 // max(h - 1, 1)).astype(np.uint8)
    return t


def quad(x0, y0, x1, y1, z=7.0, uv=((0, 0), (1, 0), (1, 1), (0, 1))):
    v = np.array([[px(x0), px(y0), z], [px(x1), px(y0), z], [px(x1), px(y1), z], [px(x0), px(y1), z]], np.float32)
    return v, np.array([[0, 1, 2], [0, 2, 3]], np.int32), np.array(uv, np.float32)


def merge(parts):
    verts, tris, uvs, base = [], [], [], 0
    for v, t, u in parts:
        verts.append(v)
        tris.append(t + base)
        uvs.append(u)
        base += len(v)
    return np.concatenate(verts), np.concatenate(tris), np.concatenate(uvs)


def scenes():
    from mvlm_amd.utils.synthetic import face_like_mesh

    out = {}
    # 1. the 40-grid face, fixed 8-view table: sub-pixel-free triangles of ~4 px, silhouettes, self-occlusion at +-45 deg
    m = face_like_mesh(grid=40, tex_size=64, seed=1)
    out["face40"] = dict(verts=m.verts, tris=m.tris, uvs=m.uvs, tex=m.texture, poses=EIGHT_VIEWS, lattice=False)
    # 1b. the bench's own regime: the 224-grid face (99 458 triangles, most of them smaller than a pixel), three poses
    m = face_like_mesh(grid=224, tex_size=256, seed=0)
    out["face224"] = dict(verts=m.verts, tris=m.tris, uvs=m.uvs, tex=m.texture,
                          poses=np.array([[30, 15, 0], [-30, -45, 0], [-17, 33, 9]], np.float64), lattice=False)
    # 2. a coarse mesh: 3x3 vertices over +-120 -> triangles of ~100 px (the rasteriser's "big triangle" path), odd-sized texture
    lin = np.linspace(-120.0, 120.0, 3)
    x, y = np.meshgrid(lin, lin)
    z = 40.0 * np.cos(x / 90.0) * np.cos(y / 70.0) - 10.0
    idx = np.arange(9).reshape(3, 3)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    u, v = np.meshgrid(np.linspace(0, 1, 3), np.linspace(0, 1, 3))
    out["coarse"] = dict(verts=np.stack([x.ravel(), y.ravel(), z.ravel()], 1).astype(np.float32),
                         tris=np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32),
                         uvs=np.stack([u.ravel(), v.ravel()], 1).astype(np.float32), tex=colour_texture(23, 37, 2),
                         poses=np.array([[0, 0, 0], [-17, 33, 9], [25, -61, -14]], np.float64), lattice=False)
    # 3. edges and vertices through pixel centres (every vertex on k + 0.5 pixels): axis-aligned and diagonal shared edges, a
    #    fan around a centre vertex, both windings, one-pixel slivers - the fill rule and nothing else: every triangle has ONE
    #    colour (its three vertices carry the centre of one texel), so a pixel's colour names the triangle that owns it
    flat_tex = colour_texture(8, 8, 3)
    parts = []

    def flat(p0, p1, p2, z, texel):
        uv = [((texel % 8) + 0.5) / 8.0, ((texel // 8) + 0.5) / 8.0]
        parts.append((np.array([[px(p[0]), px(p[1]), z] for p in (p0, p1, p2)], np.float32), np.array([[0, 1, 2]], np.int32),
                      np.array([uv] * 3, np.float32)))

    flat((10.5, 10.5), (40.5, 10.5), (40.5, 30.5), 7.0, 0)          # two quads side by side, split along a diagonal
    flat((10.5, 10.5), (40.5, 30.5), (10.5, 30.5), 7.0, 1)
    flat((40.5, 10.5), (60.5, 10.5), (60.5, 30.5), 7.0, 2)
    flat((60.5, 30.5), (40.5, 30.5), (40.5, 10.5), 7.0, 3)          # (other winding, other starting vertex)
    ring = [np.round(30.0 * np.array([np.cos(k * np.pi / 4), np.sin(k * np.pi / 4)])) for k in range(8)]
    for k in range(8):                                               # a fan around a vertex on a pixel centre, windings mixed
        a, b = ring[k], ring[(k + 1) % 8]
        pa, pb = (128.5 + a[0], 128.5 + a[1]), (128.5 + b[0], 128.5 + b[1])
        flat((128.5, 128.5), pa, pb, 5.0, 8 + k) if k % 2 == 0 else flat((128.5, 128.5), pb, pa, 5.0, 8 + k)
    flat((180.5, 20.5), (240.5, 20.5), (240.5, 80.5), -3.0, 16)       # 45-degree shared edge through pixel centres
    flat((240.5, 80.5), (180.5, 80.5), (180.5, 20.5), -3.0, 17)
    flat((20.5, 200.5), (21.5, 200.5), (20.5, 201.5), 1.0, 18)        # slivers: the three vertices are the only centres touched
    flat((30.5, 200.5), (32.5, 200.5), (31.5, 201.5), 1.0, 19)
    flat((60.5, 200.5), (90.5, 200.5), (75.5, 200.5), 1.0, 20)        # zero area: nothing
    flat((100.5, 180.5), (160.5, 240.5), (101.5, 180.5), 1.0, 21)     # a needle
    v_, t_, u_ = merge(parts)
    out["centres"] = dict(verts=v_, tris=t_, uvs=u_, tex=flat_tex, poses=np.zeros((1, 3)), lattice=True)
    # 4. texture coordinates outside [0, 1): GL_REPEAT in both directions, negative and > 2, on a 7 x 5 texture
    v_, t_, u_ = quad(16.0, 32.0, 240.0, 224.0, uv=((-1.3, -0.7), (2.4, -0.7), (2.4, 1.9), (-1.3, 1.9)))
    out["uv_wrap"] = dict(verts=v_, tris=t_, uvs=u_, tex=colour_texture(5, 7, 4), poses=np.array([[0, 0, 0], [12, -20, 30]], np.float64),
                          lattice=False)
    # 5. the last texel and u = 1 exactly: 64-pixel quads over an 8-texel row, vertices on pixel centres, u running both ways
    parts = [quad(10.5, 10.5, 74.5, 42.5), quad(100.5, 10.5, 164.5, 42.5, uv=((1, 1), (0, 1), (0, 0), (1, 0))),
             quad(10.0, 100.0, 74.0, 132.0), quad(100.0, 100.0, 228.0, 164.0, uv=((0, 0), (3, 0), (3, 2), (0, 2)))]
    v_, t_, u_ = merge(parts)
    out["last_texel"] = dict(verts=v_, tris=t_, uvs=u_, tex=colour_texture(4, 8, 5), poses=np.zeros((1, 3)), lattice=True)
    # 6. coplanar overlapping triangles, screen-parallel (equal depth bit for bit): GL_LEQUAL -> the later one is seen
    parts = [quad(40.0, 40.0, 160.0, 160.0, z=20.0, uv=((0, 0), (0.5, 0), (0.5, 0.5), (0, 0.5))),
             quad(100.0, 100.0, 220.0, 220.0, z=20.0, uv=((0.5, 0.5), (1, 0.5), (1, 1), (0.5, 1))),
             quad(20.0, 180.0, 84.0, 244.0, z=-40.0), quad(50.0, 170.0, 114.0, 234.0, z=-40.5)]     # and a plain occlusion
    v_, t_, u_ = merge(parts)
    out["coplanar"] = dict(verts=v_, tris=t_, uvs=u_, tex=colour_texture(8, 8, 6), poses=np.zeros((1, 3)), lattice=True)
    # 7. the clip range: a slab running from in front of the near plane (z > 500) to behind the far one (z < -1000)
    v_ = np.array([[px(16.0), px(64.0), 620.0], [px(240.0), px(64.0), -1180.0], [px(240.0), px(192.0), -1180.0], [px(16.0), px(192.0), 620.0]],
                  np.float32)
    out["clip"] = dict(verts=v_, tris=np.array([[0, 1, 2], [0, 2, 3]], np.int32),
                       uvs=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), tex=colour_texture(16, 16, 7), poses=np.zeros((1, 3)),
                       lattice=True)
    # 8. triangles leaving the window on every side (no clipping in x / y may move an edge)
    v_ = np.array([[-400.0, -100.0, 10.0], [120.0, -300.0, -30.0], [20.0, 380.0, 60.0], [300.0, 260.0, -80.0], [-60.0, 140.0, 30.0],
                   [330.0, -200.0, 3.0]], np.float32)
    out["offscreen"] = dict(verts=v_, tris=np.array([[0, 1, 2], [3, 4, 5]], np.int32),
                            uvs=np.array([[0, 0], [4, 0], [2, 4], [0, 3], [4, 3], [1, 1]], np.float32), tex=colour_texture(9, 11, 8),
                            poses=np.array([[0, 0, 0], [-8, 14, 100]], np.float64), lattice=False)
    # 9. the plane z = 0: window depth 1/3, and 255 / 3 = 85 exactly - the depth byte hangs on the last bit of the depth value
    v_, t_, u_ = quad(32.0, 32.0, 224.0, 224.0, z=0.0)
    out["third"] = dict(verts=v_, tris=t_, uvs=None, tex=None, poses=np.zeros((1, 3)), lattice=True)
    # 10. a depth ramp: every byte value of the depth plane and the bytes' boundaries, on a slab tilted in x and in y
    #     (z values chosen so that no pixel centre falls exactly on a byte boundary)
    v_ = np.array([[px(8.0), px(8.0), 481.3], [px(248.0), px(8.0), -301.7], [px(248.0), px(248.0), -959.9], [px(8.0), px(248.0), -177.1]],
                  np.float32)
    out["depth_ramp"] = dict(verts=v_, tris=np.array([[0, 1, 2], [0, 2, 3]], np.int32), uvs=None, tex=None, poses=np.zeros((1, 3)),
                             lattice=True)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(REPO / "tests" / "golden" / "gl_raster.npz"))
    args = ap.parse_args()
    gl = GLReference(256)
    store = {"meta": np.array(json.dumps({"gl": gl.info, "generator": "tools/make_gl_golden.py"}))}
    names = []
    for name, sc in scenes().items():
        gl.set_mesh(sc["uvs"], sc["tris"], sc["tex"])
        images, zs = [], []
        for rx, ry, rz in sc["poses"]:
            m = rotation(rx, ry, rz)
            v = sc["verts"].astype(np.float64)
            vv = np.stack([(m[k, 0] * v[:, 0] + m[k, 1] * v[:, 1]) + m[k, 2] * v[:, 2] for k in range(3)], 1).astype(np.float32)
            rgb, z = gl.draw(vv)
            img = reference_postprocess(rgb, z)
            images.append(np.round(img * 255.0).astype(np.uint8))          # exact: every value is k / 255
            assert np.array_equal(images[-1].astype(np.float32) / np.float32(255), img)
            zs.append(np.flip(z, 0).copy())                                # image row order, like the stack
        store[f"{name}.verts"] = sc["verts"].astype(np.float32)
        store[f"{name}.tris"] = sc["tris"].astype(np.int32)
        if sc["uvs"] is not None:
            store[f"{name}.uvs"] = sc["uvs"].astype(np.float32)
            store[f"{name}.tex"] = sc["tex"]
        store[f"{name}.poses"] = np.asarray(sc["poses"], np.float64)
        store[f"{name}.lattice"] = np.array(bool(sc["lattice"]))
        store[f"{name}.image_u8"] = np.stack(images)                       # the reference's image_stack x 255
        store[f"{name}.z"] = np.stack(zs)                                  # float window depth per pixel (1.0 = background)
        names.append(name)
        print(f"{name}: {len(sc['poses'])} views, {int((np.stack(zs) < 1).sum())} covered pixels")
    store["scenes"] = np.array(names)
    np.savez_compressed(args.out, **store)
    print(f"wrote {args.out} ({Path(args.out).stat().st_size} bytes) with {gl.info}")


if __name__ == "__main__":
    main()
