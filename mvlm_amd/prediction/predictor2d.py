"""The 2-D predictor plug-in contract (reference src/mvlm/prediction/predictor2d.py:8-27)."""
from __future__ import annotations

import abc

import numpy as np

__all__ = ["Predictor2D"]


class Predictor2D(abc.ABC):
    def __init__(self):
        pass

    @abc.abstractmethod
    def predict_landmarks_from_images(self, image_stack: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """image_stack [N,256,256,4] -> (landmarks [n_landmarks, n_views, 3] = (row, col, score),
        valid [n_views] bool)."""

    @abc.abstractmethod
    def get_lm_count(self) -> int:
        pass
