"""Oracle: heatmap maxima -> 3-D rays -> per-landmark consensus.

Restates src/mvlm/utils/estimator3d.py (:8-15 rotations, :31-90 rays,
:92-137 one-shot RANSAC, :140-155 filters, :158-183 per-landmark loop) and
src/mvlm/utils/utils3d.py:99-124 (least-squares line intersection, pinv 3x3).
TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
from __future__ import annotations

import numpy as np

X_MIN, X_LEN = -150, 300
Y_MIN, Y_LEN = -150, 300


def rotation_matrix_x(a):
    return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])


def rotation_matrix_y(a):
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def rotation_matrix_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])


def view_rotation(rx, ry, rz) -> np.ndarray:
    """M = Ry @ Rx @ Rz in degrees (estimator3d.py:57; mirrors the VTK
    RotateY/RotateX/RotateZ order of render3d.py:141-143)."""
    return (rotation_matrix_y(np.deg2rad(ry)) @ rotation_matrix_x(np.deg2rad(rx))) @ rotation_matrix_z(np.deg2rad(rz))


def estimate_landmark_lines(image_size: int, landmarks_stack: np.ndarray, transform_stack: np.ndarray):
    """[NL,N,3] (row, col, value) f32 + [N,>=3] poses -> starts, ends [NL,N,3] f64.

    estimator3d.py:31-90.  The pixel->model arithmetic runs in the dtype of the
    landmark scalars (float32 for the CNN predictors; python ints are weak), the
    rotation in float64 via ``t.T @ points``.
    """
    n_landmarks, n_views = landmarks_stack.shape[:2]
    starts = np.empty((n_landmarks, n_views, 3))
    ends = np.empty((n_landmarks, n_views, 3))
    img_size = hm_size = image_size
    for idx in range(n_views):
        rx, ry, rz = transform_stack[idx, :3]
        t = np.diag(np.ones(4))
        t[0:3, 0:3] = view_rotation(rx, ry, rz)
        for lm in range(n_landmarks):
            y = landmarks_stack[lm, idx, 0]
            x = landmarks_stack[lm, idx, 1]
            y = y / hm_size * img_size
            x = x / hm_size * img_size
            px = (x / img_size) * X_LEN + X_MIN
            py = ((img_size - 1 - y) / img_size) * Y_LEN + Y_MIN
            pts = np.array([[px, px], [py, py], [500.0, -500.0], [1.0, 1.0]], dtype=np.float64)
            pts = np.matmul(t.T, pts)
            starts[lm, idx, :] = pts[:3, 0]
            ends[lm, idx, :] = pts[:3, 1]
    return starts, ends


def compute_intersection_between_lines(pa: np.ndarray, pb: np.ndarray) -> np.ndarray:
    """utils3d.py:99-124 - least-squares point closest to k lines, via pinv(S)."""
    n_lines = pa.shape[0]
    si = pb - pa
    ni = np.divide(si, np.transpose(np.sqrt(np.sum(si ** 2, 1)) * np.ones((3, n_lines))))
    nx, ny, nz = ni[:, 0], ni[:, 1], ni[:, 2]
    sxx = np.sum(nx ** 2 - 1)
    syy = np.sum(ny ** 2 - 1)
    szz = np.sum(nz ** 2 - 1)
    sxy = np.sum(nx * ny)
    sxz = np.sum(nx * nz)
    syz = np.sum(ny * nz)
    s = np.array([[sxx, sxy, sxz], [sxy, syy, syz], [sxz, syz, szz]])
    cx = np.sum(pa[:, 0] * (nx ** 2 - 1) + pa[:, 1] * (nx * ny) + pa[:, 2] * (nx * nz))
    cy = np.sum(pa[:, 0] * (nx * ny) + pa[:, 1] * (ny ** 2 - 1) + pa[:, 2] * (ny * nz))
    cz = np.sum(pa[:, 0] * (nx * nz) + pa[:, 1] * (ny * nz) + pa[:, 2] * (nz ** 2 - 1))
    c = np.array([[cx], [cy], [cz]])
    return np.matmul(np.linalg.pinv(s), c)[:, 0]


def _sq_dist_to_lines(p, pa, pb):
    top = np.cross(p - pa, p - pb)
    return (np.linalg.norm(top, axis=1) / np.linalg.norm(pb - pa, axis=1)) ** 2


def ransac_with_draw(pa, pb, ran_lines):
    """estimator3d.py:92-137 with the single index draw (:105) supplied."""
    best_error = 100000000
    dist_thres = 10 * 10
    n_lines = len(pa)
    d = n_lines / 3
    used = -1
    best_p = (0, 0, 0)
    p_est = compute_intersection_between_lines(pa[ran_lines, :], pb[ran_lines, :])
    distances = _sq_dist_to_lines(p_est, pa, pb)
    n_in = np.sum(distances < dist_thres)
    if n_in > d:
        idx = distances < dist_thres
        p_est = compute_intersection_between_lines(pa[idx, :], pb[idx, :])
        distances = _sq_dist_to_lines(p_est, pa[idx, :], pb[idx, :])
        sum_squared = np.sum(distances) / n_in
        if sum_squared < best_error:
            best_error, best_p, used = sum_squared, p_est, n_in
    if used == -1:
        best_p = compute_intersection_between_lines(pa, pb)
    return best_p, best_error


def line_mask(values: np.ndarray, mode: str, q: float, thr: float) -> np.ndarray:
    """estimator3d.py:140-155 - which views survive for one landmark."""
    if mode == "absolute":
        return values > thr
    if mode == "quantile":
        return values > np.quantile(values, q)
    raise ValueError(f"Unknown mode for line matching in Estimator: {mode}")


def estimate_landmarks_from_lines(landmark_stack, lines_s, lines_e, mode="quantile", q=0.5, thr=0.5,
                                  draws: list | None = None):
    """estimator3d.py:158-183.

    The one-shot RANSAC draw comes from the global numpy RNG exactly as in the
    reference (``np.random.choice(range(k), 8, replace=True)``, :105), once per
    landmark with >= 3 surviving lines, in landmark order.  ``draws`` (optional
    list) records (landmark, k, indices) for replay by the HIP path.
    """
    n_landmarks = lines_s.shape[0]
    landmarks = np.empty((n_landmarks, 3))
    sum_error = 0
    for lm in range(n_landmarks):
        idx = line_mask(landmark_stack[lm, :, 2], mode, q, thr)
        pa, pb = lines_s[lm][idx], lines_e[lm][idx]
        if len(pa) < 3:
            p = compute_intersection_between_lines(pa, pb)
        else:
            ran = np.random.choice(range(len(pa)), 8, replace=True)
            if draws is not None:
                draws.append((lm, len(pa), ran.copy()))
            p, err = ransac_with_draw(pa, pb, ran)
            sum_error = sum_error + err
        landmarks[lm, :] = p
    return landmarks, sum_error / n_landmarks
