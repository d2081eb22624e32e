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
    def __init__(self, n_landmarks: int, fn=None, device_fn=None):
        """fn(image_stack) -> (landmarks [NL,N,3] float32, valid [N] bool) on the host (the reference's
        slot contract); device_fn(image_stack_dev [N,256,256,4] torch) -> landmarks torch f32 [NL,N,3] on the
        same GPU: with it the pipeline keeps the whole call in HBM (fused path).  A detector that finds nothing in
        some views (mediapipepredictor.py:38-41: ``valid[idx] = False``) returns ``(landmarks, valid)`` with ``valid`` a
        host bool array [N]; the pipeline drops those views before rays and consensus (general_pipeline.py:93-95)."""
        super().__init__()
        if fn is None and device_fn is None:
            raise ValueError("PrecomputedPredictor needs fn or device_fn")
        self._nl = int(n_landmarks)
        self._fn = fn
        if device_fn is not None:
            self.predict_device = lambda images: self._checked_device(device_fn(images), int(images.shape[0]))

    def _checked_device(self, result, n_views: int):
        lms, valid = result if isinstance(result, tuple) else (result, None)
        if tuple(lms.shape) != (self._nl, n_views, 3) or str(lms.dtype) != "torch.float32":
            raise RuntimeError(f"Unexpected landmark stack: {tuple(lms.shape)} {lms.dtype}")
        if valid is None:
            return lms.contiguous()
        valid = np.asarray(valid, dtype=bool)
        if valid.shape != (n_views,):
            raise RuntimeError(f"Unexpected validity mask: {valid.shape} for {n_views} views")
        return lms.contiguous(), valid

    def get_lm_count(self) -> int:
        return self._nl

    def predict_landmarks_from_images(self, image_stack: np.ndarray):
        if self._fn is None:  # device-only predictor driven through the numpy slot protocol
            import torch

            res = self.predict_device(torch.from_numpy(np.ascontiguousarray(image_stack, np.float32)).cuda())
            lms, valid = res if isinstance(res, tuple) else (res, np.ones(image_stack.shape[0], bool))
            return lms.cpu().numpy(), valid
        lms, valid = self._fn(image_stack)
        lms = np.asarray(lms, dtype=np.float32)
        if lms.shape[0] != self._nl or lms.shape[1] != image_stack.shape[0] or lms.shape[2] != 3:
            raise RuntimeError(f"Unexpected landmark stack shape: {lms.shape}")
        return lms, np.asarray(valid, dtype=bool)
