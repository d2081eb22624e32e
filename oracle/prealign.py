"""Oracle: the ``pre-align`` block of a Deep-MVLM config, restated on the CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Follows the reference's
``Utils3D.apply_pre_transformation`` (src/mvlm/utils/utils3d.py:465-503) and
``transform_landmarks_to_original_space`` (:505-527) statement by statement: a
vtkTransform starts as the identity in its default PreMultiply mode, so every call
multiplies the current matrix from the right - Scale, RotateY, RotateX, RotateZ,
Translate - and vtkTransformPolyDataFilter writes float points.  The arithmetic itself
lives in the absent third-party ``vtk`` (un-pinned): **parity unpinned** at
the bit level; the statement below is the textbook homogeneous-matrix form of those calls.
"""
from __future__ import annotations

import math

import numpy as np


def _premultiply(current: np.ndarray, op: np.ndarray) -> np.ndarray:
    return current @ op


def _rotation(axis: str, degrees: float) -> np.ndarray:
    a = math.radians(float(degrees))
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    m[i, i] = c
    m[j, j] = c
    m[i, j] = -s
    m[j, i] = s
    return m


def pre_transformation(points: np.ndarray, block: dict):
    """(float32 points after the transform, 4x4 matrix) - utils3d.py:466-503."""
    translation = [0.0, 0.0, 0.0]
    if block["align_center_of_mass"]:                       # :467-473, vtkCenterOfMass without scalar weights
        cm = np.asarray(points, dtype=np.float64).sum(axis=0) / len(points)
        translation = [-cm[0], -cm[1], -cm[2]]
    t = np.eye(4)                                           # :475-476
    s = float(block["scale"])
    t = _premultiply(t, np.diag([s, s, s, 1.0]))            # :483
    t = _premultiply(t, _rotation("y", block["rot_y"]))     # :484
    t = _premultiply(t, _rotation("x", block["rot_x"]))     # :485
    t = _premultiply(t, _rotation("z", block["rot_z"]))     # :486
    move = np.eye(4)
    move[:3, 3] = translation
    t = _premultiply(t, move)                               # :487
    hom = np.concatenate([np.asarray(points, dtype=np.float64), np.ones((len(points), 1))], axis=1)
    out = (hom @ t.T)[:, :3]
    return out.astype(np.float32), t                        # :490-493: float points out of the filter


def landmarks_to_original_space(landmarks: np.ndarray, t: np.ndarray) -> np.ndarray:
    """utils3d.py:505-527: the landmarks through ``t.GetInverse()``."""
    inv = np.linalg.inv(t)
    hom = np.concatenate([np.asarray(landmarks, dtype=np.float64), np.ones((len(landmarks), 1))], axis=1)
    return (hom @ inv.T)[:, :3]
