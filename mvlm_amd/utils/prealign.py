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

__all__ = ["prealign_matrix", "apply_prealign", "landmarks_to_original_space"]


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


def apply_prealign(mesh: Mesh, cfg: dict) -> tuple[Mesh, np.ndarray]:
    """Transformed copy of the mesh (float32 points like vtkTransformPolyDataFilter) + the matrix."""
    m = prealign_matrix(mesh.verts, cfg)
    v = mesh.verts.astype(np.float64) @ m[:3, :3].T + m[:3, 3]
    return Mesh(v.astype(np.float32), mesh.tris, mesh.uvs, mesh.texture, mesh.path), m


def landmarks_to_original_space(landmarks: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Inverse transform of [NL,3] landmarks (utils3d.py:505-527)."""
    inv = np.linalg.inv(m)
    return np.asarray(landmarks, dtype=np.float64) @ inv[:3, :3].T + inv[:3, 3]
