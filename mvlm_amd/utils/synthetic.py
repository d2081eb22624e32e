"""Synthetic inputs for benchmarks and tests (no datasets are reachable offline):
a ~100k-triangle face-like height field with a planar-UV texture, centred at the
origin inside the renderer's +-150 view box (the live reference path renders the
mesh as-is), as BASELINE.json's metric asks for.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .mesh_io import Mesh, write_obj


def face_like_mesh(grid: int = 224, tex_size: int = 512, seed: int = 0) -> Mesh:
    """(grid x grid) vertex height field -> 2*(grid-1)^2 triangles (224 -> 99 458)."""
    rs = np.random.RandomState(seed)
    lin = np.linspace(-100.0, 100.0, grid)
    x, y = np.meshgrid(lin, lin)
    z = 60.0 * np.exp(-(x ** 2 + y ** 2) / (2 * 70.0 ** 2))
    z += 25.0 * np.exp(-((x) ** 2 + (y + 10) ** 2) / (2 * 12.0 ** 2))          # nose
    z -= 8.0 * np.exp(-((x - 35) ** 2 + (y - 30) ** 2) / (2 * 10.0 ** 2))       # eye sockets
    z -= 8.0 * np.exp(-((x + 35) ** 2 + (y - 30) ** 2) / (2 * 10.0 ** 2))
    z -= 5.0 * np.exp(-((x) ** 2 / (2 * 25.0 ** 2) + (y + 45) ** 2 / (2 * 6.0 ** 2)))  # mouth
    z -= z.mean()
    verts = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1).astype(np.float32)
    u, v = np.meshgrid(np.linspace(0.0, 1.0, grid), np.linspace(0.0, 1.0, grid))
    uvs = np.stack([u.ravel(), v.ravel()], axis=1).astype(np.float32)
    idx = np.arange(grid * grid).reshape(grid, grid)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32)
    yy, xx = np.mgrid[0:tex_size, 0:tex_size]
    tex = np.empty((tex_size, tex_size, 3), np.float32)
    tex[..., 0] = 170 + 50 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    tex[..., 1] = 130 + 40 * np.cos(xx / 19.0 + yy / 31.0)
    tex[..., 2] = 110 + 30 * np.sin((xx + yy) / 11.0)
    tex += rs.randint(-12, 13, size=tex.shape)
    texture = np.clip(tex, 0, 255).astype(np.uint8)
    return Mesh(verts, tris, uvs, texture, None)


def write_face_like_obj(path, grid: int = 224, tex_size: int = 512, seed: int = 0) -> Path:
    """Write the synthetic mesh as ``<path>`` (+ same-stem .jpg texture)."""
    path = Path(path)
    m = face_like_mesh(grid, tex_size, seed)
    write_obj(path, m.verts, m.tris, m.uvs, m.texture)
    return path


def unaligned_copy(mesh: Mesh, pre_align: dict, offset=(3.0, -2.0, 1.5)) -> Mesh:
    """The "raw scan" a config's ``pre-align`` block is written for: a copy of ``mesh`` in the coordinates from which
    that block (utils/prealign.py) brings it back into the renderer's view box - shrunk by the block's scale, turned
    by its inverse rotation and, when the block centres the scan, moved off the origin by ``offset``."""
    from .prealign import prealign_matrix

    m = prealign_matrix(np.zeros((1, 3)), dict(pre_align, align_center_of_mass=False))
    v = mesh.verts.astype(np.float64) @ np.linalg.inv(m[:3, :3]).T
    if pre_align.get("align_center_of_mass", False):
        v = v + np.asarray(offset, dtype=np.float64)
    return Mesh(v.astype(np.float32), mesh.tris, mesh.uvs, mesh.texture, None)
