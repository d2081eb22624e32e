"""Oracle: snap landmarks to the closest point of the triangle surface.

Stands in for src/mvlm/utils/estimator3d.py:252-285, whose arithmetic lives in
the absent third-party ``vtk`` (vtkCellLocator.FindClosestPoint).  The closest
point on a triangle mesh is mathematically unique up to ties, so an exact
brute-force search is a sound oracle; PARITY UNPINNED against VTK itself.
TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
from __future__ import annotations

import numpy as np


def closest_point_on_triangles(p: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray):
    """Closest point on each triangle (a,b,c)[T,3] to p[3] (region walk over the
    Voronoi regions of the triangle).  Returns (points [T,3], dist2 [T])."""
    ab, ac, ap = b - a, c - a, p - a
    d1 = np.einsum("ij,ij->i", ab, ap)
    d2 = np.einsum("ij,ij->i", ac, ap)
    bp = p - b
    d3 = np.einsum("ij,ij->i", ab, bp)
    d4 = np.einsum("ij,ij->i", ac, bp)
    cp = p - c
    d5 = np.einsum("ij,ij->i", ab, cp)
    d6 = np.einsum("ij,ij->i", ac, cp)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    out = np.empty_like(a)
    done = np.zeros(len(a), bool)

    def put(mask, pts):
        m = mask & ~done
        out[m] = pts[m]
        done[m] = True

    with np.errstate(divide="ignore", invalid="ignore"):
        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        # d1 - d3 = |ab|^2: a triangle with a == b is the segment ac (vtkCleanPolyData makes it a line cell) - edge ac below
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0) & (d1 - d3 > 0), a + (d1 / (d1 - d3))[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), c)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + w[:, None] * (c - b))
        denom = 1.0 / (va + vb + vc)
        v, w2 = vb * denom, vc * denom
        put(np.ones(len(a), bool), a + ab * v[:, None] + ac * w2[:, None])
    d = out - p
    return out, np.einsum("ij,ij->i", d, d)


def project_landmarks_to_surface(verts: np.ndarray, tris: np.ndarray, landmarks: np.ndarray) -> np.ndarray:
    """[V,3] f32 vertices, [T,3] int triangles, [NL,3] f64 -> [NL,3] f64."""
    v = verts.astype(np.float64)
    a, b, c = v[tris[:, 0]], v[tris[:, 1]], v[tris[:, 2]]
    out = np.copy(landmarks)
    for i in range(landmarks.shape[0]):
        pts, d2 = closest_point_on_triangles(landmarks[i], a, b, c)
        d2 = np.where(np.isfinite(d2), d2, np.inf)  # a triangle the walk cannot evaluate (a == b == c with p = nan ...) never wins
        if np.isfinite(d2).any():                   # (a non-finite landmark passes through unchanged)
            out[i] = pts[int(np.argmin(d2))]
    return out


def clip_rays_to_mesh(verts: np.ndarray, tris: np.ndarray, starts: np.ndarray, ends: np.ndarray):
    """First intersection of each segment starts[i]->ends[i] with the triangle surface.

    Stands in for ``RayVisualizer._clip_rays_to_mesh``
    (src/mvlm/visualization/ray_visualizer.py:172-192: vtkOBBTree.IntersectWithLine, first point);
    PARITY UNPINNED against VTK itself - the first hit of a segment with a triangle soup is unique
    up to ties, so brute-force Moeller-Trumbore in float64 is a sound oracle.
    starts/ends [...,3] f64 -> (new_ends [...,3] f64, hit [...] bool); misses keep their end.
    """
    v = verts.astype(np.float64)
    a = v[tris[:, 0]]
    e1, e2 = v[tris[:, 1]] - a, v[tris[:, 2]] - a

    def dot(p, q):
        return (p[..., 0] * q[..., 0] + p[..., 1] * q[..., 1]) + p[..., 2] * q[..., 2]

    def cross(p, q):
        return np.stack([p[..., 1] * q[..., 2] - p[..., 2] * q[..., 1], p[..., 2] * q[..., 0] - p[..., 0] * q[..., 2],
                         p[..., 0] * q[..., 1] - p[..., 1] * q[..., 0]], axis=-1)

    shape = starts.shape[:-1]
    s, e = starts.reshape(-1, 3), ends.reshape(-1, 3)
    new_ends = e.copy()
    hit = np.zeros(len(s), bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(len(s)):
            d = e[i] - s[i]
            pvec = cross(d[None, :], e2)
            det = dot(e1, pvec)
            inv = 1.0 / det
            tvec = s[i][None, :] - a
            u = dot(tvec, pvec) * inv
            qvec = cross(tvec, e1)
            vv = dot(d[None, :], qvec) * inv
            t = dot(e2, qvec) * inv
            ok = (det != 0.0) & (u >= 0.0) & (u <= 1.0) & (vv >= 0.0) & (u + vv <= 1.0) & (t >= 0.0) & (t <= 1.0)
            if ok.any():
                tt = np.where(ok, t, np.inf)
                k = int(np.argmin(tt))  # first minimum = lowest triangle id
                new_ends[i] = s[i] + tt[k] * d
                hit[i] = True
    return new_ends.reshape(starts.shape), hit.reshape(shape)
