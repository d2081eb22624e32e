"""Shared pieces of the GPU parity tests (test infrastructure: imports oracle/)."""
import numpy as np


def survivors(maxima: np.ndarray, q: float = 0.5) -> np.ndarray:
    """estimator3d.py:140-147 for every landmark at once: views whose score exceeds the landmark's q-quantile."""
    v = maxima[:, :, 2]
    return v > np.quantile(v, q, axis=1, keepdims=True)


def oracle_landmarks_with_draws(starts, ends, keep, draws, verts, tris, landmarks):
    """The oracle's consensus + snap for the given landmarks, each with ITS OWN draw of eight line indices handed in
    (estimator3d.py:92-137, :174-179) instead of the global RNG's next one: a landmark's result then depends on its rays,
    its surviving views and its draw only - not on how many draws the landmarks before it consumed."""
    from oracle import estimator as oest
    from oracle import surface

    raw = np.empty((len(landmarks), 3))
    for i, lm in enumerate(landmarks):
        pa, pb = starts[lm][keep[lm]], ends[lm][keep[lm]]
        if len(pa) < 3:
            raw[i] = oest.compute_intersection_between_lines(pa, pb)
        else:
            raw[i], _ = oest.ransac_with_draw(pa, pb, np.asarray(draws[lm]))
    return surface.project_landmarks_to_surface(verts, tris, raw)


def compare_with_the_oracle_landmark_by_landmark(got, gmax, inter, mesh, estimator, seed: int):
    """Every landmark whose views ALL picked the oracle's pixel and whose surviving views are the oracle's, compared with the
    oracle's result for the draw the product really made for it (the product's draws = the estimator's own rule replayed
    from ``seed`` for the product's survivor counts).  Returns (mask of the landmarks compared, max deviation among them)."""
    diff = ~np.all(gmax[:, :, :2] == inter["maxima"][:, :, :2], axis=2)
    sg, so = survivors(gmax), survivors(inter["maxima"])
    same = ~diff.any(axis=1) & np.all(sg == so, axis=1)
    np.random.seed(seed)
    verbose, estimator.verbose = estimator.verbose, False
    try:
        draws = estimator.draw_ransac_indices(sg.sum(axis=1))
    finally:
        estimator.verbose = verbose
    idx = np.nonzero(same)[0]
    want = oracle_landmarks_with_draws(inter["starts"], inter["ends"], so, draws, mesh.verts, mesh.tris, idx)
    return same, diff, (np.abs(got[idx] - want).max() if len(idx) else 0.0)
