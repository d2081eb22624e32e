"""Oracle: camera pose table.  Restates src/mvlm/utils/render3d.py:79-112.
TEST INFRASTRUCTURE - see oracle/__init__.py."""
from __future__ import annotations

import numpy as np


def random_transform(size, min_x=-40, max_x=40, min_y=-80, max_y=80, min_z=-20, max_z=20,
                     min_scale=1.4, max_scale=1.9, min_tx=-20, max_tx=20, min_ty=-20, max_ty=20):
    """render3d.py:79-89 - global numpy RNG, draw order rx, ry, rz, scale, tx, ty."""
    rx = np.random.randint(min_x, max_x, size=size)
    ry = np.random.randint(min_y, max_y, size=size)
    rz = np.random.randint(min_z, max_z, size=size)
    scale = np.random.uniform(min_scale, max_scale, size=size)
    tx = np.random.randint(min_tx, max_tx, size=size)
    ty = np.random.randint(min_ty, max_ty, size=size)
    return np.stack((rx, ry, rz, scale, tx, ty), axis=1)


def generate_3d_transformations(n_views: int, **angle_kw) -> np.ndarray:
    """render3d.py:92-112 - fixed 8-view table (float32) or random poses (float64)."""
    if n_views == 8:
        rows = [[rx, ry, 0, 0, 0, 0] for rx in (30, -30) for ry in (15, -15, 45, -45)]
        return np.array(rows, dtype=np.float32)
    return random_transform(n_views, **angle_kw)
