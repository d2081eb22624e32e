"""Pre-alignment of a scan before rendering, and the inverse mapping of the landmarks.

Honours the ``pre-align`` block of the Deep-MVLM JSON configs.  In the reference this
exists only in the never-instantiated legacy class (src/mvlm/utils/utils3d.py:465-527:
``apply_pre_transformation`` / ``transform_landmarks_to_original_space``); the live pipeline
renders the mesh as-is, so meshes must already sit centred inside the +-150 view box.  The
transform restated here is VTK's pre-multiplied sequence Scale, RotateY, RotateX, RotateZ,
Translate(-centre of mass): p' = S * Ry * Rx * Rz * (p + t).
"""
from __future__ import annotations

import numpy as np

from .mesh_io import Mesh

__all__ = ["prealign_matrix", "apply_prealign", "landmarks_to_original_space", "is_active", "aligned", "write_pre_aligned"]


def prealign_matrix(verts: np.ndarray, cfg: dict) -> np.ndarray:
    """4x4 float64 matrix of the ``pre-align`` block (utils3d.py:466-487)."""
    t = np.zeros(3)
    if cfg.get("align_center_of_mass", False):
        # vtkCenterOfMass with UseScalarsAsWeights(False): the mean of the points
        t = -np.asarray(verts, dtype=np.float64).mean(axis=0)
    rx, ry, rz = (np.deg2rad(float(cfg.get(k, 0))) for k in ("rot_x", "rot_y", "rot_z"))
    s = float(cfg.get("scale", 1))
    mx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    my = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    mz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = s * (my @ mx @ mz)
    m[:3, 3] = m[:3, :3] @ t
    return m


def is_active(cfg: dict | None) -> bool:
    """Does the block change the mesh at all?  (every reference config carries one; most are neutral)"""
    if not cfg:
        return False
    return bool(cfg.get("align_center_of_mass", False)) or any(float(cfg.get(k, 0)) != 0.0 for k in ("rot_x", "rot_y", "rot_z")) \
        or float(cfg.get("scale", 1)) != 1.0


def apply_prealign(mesh: Mesh, cfg: dict) -> tuple[Mesh, np.ndarray]:
    """Transformed copy of the mesh (float32 points like vtkTransformPolyDataFilter) + the matrix.
    The copy remembers the matrix (``Mesh.to_original``) so whoever ends up with the landmarks can map them back."""
    m = prealign_matrix(mesh.verts, cfg)
    v = mesh.verts.astype(np.float64) @ m[:3, :3].T + m[:3, 3]
    # (a texture that is still the JPEG file's bytes stays that way: the upload decodes it on the device)
    out = Mesh(v.astype(np.float32), mesh.tris, mesh.uvs, getattr(mesh, "_texture", None), mesh.path, to_original=m,
               texture_jpeg=mesh.texture_jpeg)
    out._texture_ahead = getattr(mesh, "_texture_ahead", None)  # (a texture decoded ahead moves to the copy that is uploaded)
    if out._texture_ahead is not None:
        mesh._texture_ahead = None
    return out, m


def aligned(mesh: Mesh, cfg: dict | None) -> Mesh:
    """``mesh`` itself when the block is neutral or the mesh has been through it already, else the transformed copy."""
    if not is_active(cfg) or mesh.to_original is not None:
        return mesh
    return apply_prealign(mesh, cfg)[0]


def write_pre_aligned(mesh: Mesh, path) -> None:
    """``write_pre_aligned`` (utils3d.py:489-494 writes the transformed surface as a legacy .vtk): the same file kind,
    ASCII POLYDATA with the points and triangles of the aligned mesh."""
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 3.0\npre-aligned mesh\nASCII\nDATASET POLYDATA\n")
        f.write(f"POINTS {mesh.n_verts} float\n")
        np.savetxt(f, mesh.verts, fmt="%.9g")
        f.write(f"POLYGONS {mesh.n_tris} {4 * mesh.n_tris}\n")
        np.savetxt(f, np.concatenate([np.full((mesh.n_tris, 1), 3, np.int64), mesh.tris.astype(np.int64)], axis=1), fmt="%d")


def landmarks_to_original_space(landmarks: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Inverse transform of [NL,3] landmarks (utils3d.py:505-527)."""
    inv = np.linalg.inv(m)
    return np.asarray(landmarks, dtype=np.float64) @ inv[:3, :3].T + inv[:3, 3]
