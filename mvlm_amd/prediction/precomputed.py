"""A Predictor2D that returns landmarks computed elsewhere.

The reference's MediaPipe / dlib / face_alignment predictors wrap third-party
detectors that are out of scope here.  Their *output format*
- ``[n_landmarks, n_views, 3]`` (row, col, score) plus a per-view validity mask
(mediapipepredictor.py:29-49) - is all the rest of the path sees, so this class
lets any such detector (or a synthetic stand-in for the 478-landmark fusion
stress configuration) feed the GPU fusion path.
"""
from __future__ import annotations

import numpy as np

from .predictor2d import Predictor2D

__all__ = ["PrecomputedPredictor"]


class PrecomputedPredictor(Predictor2D):
    def __init__(self, n_landmarks: int, fn):
        """fn(image_stack) -> (landmarks [NL,N,3] float32, valid [N] bool)"""
        super().__init__()
        self._nl = int(n_landmarks)
        self._fn = fn

    def get_lm_count(self) -> int:
        return self._nl

    def predict_landmarks_from_images(self, image_stack: np.ndarray):
        lms, valid = self._fn(image_stack)
        lms = np.asarray(lms, dtype=np.float32)
        if lms.shape[0] != self._nl or lms.shape[1] != image_stack.shape[0] or lms.shape[2] != 3:
            raise RuntimeError(f"Unexpected landmark stack shape: {lms.shape}")
        return lms, np.asarray(valid, dtype=bool)
