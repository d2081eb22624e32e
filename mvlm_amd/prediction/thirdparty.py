"""Adapters that put the reference's third-party 2-D detectors behind the MI355X pipeline.

The reference wraps MediaPipe's face landmarker (478 landmarks), dlib's 68-point shape predictor and
``face_alignment`` (68 points) as ``Predictor2D`` objects that are fed with the rendered views and whose
``(row, col, score)`` stack goes into the same ray / consensus / snap stages as the heatmap network's
(src/mvlm/prediction/mediapipepredictor.py:12-49, dlibpredictor.py:13-74, face_alignmentpredictor.py:20-53).
The detectors themselves are un-vendored dependencies (setup.cfg:40-61) and out of this build's scope; what
IS on the hot path - the views they look at (HIP rasteriser) and the fusion of their output (HIP consensus,
478 x 128 is BASELINE configs[4]) - runs on the GPU.  These classes only translate: one detector call per
view, the per-library output convention, ``NaN`` + ``valid = False`` for views without a detection
(general_pipeline.py:93-95 slices them away).  A missing library raises ``ImportError`` when the predictor is
built - never a silent substitute.

``DetectorPredictor`` is the shared loop; any other detector plugs in by subclassing it or through
``PrecomputedPredictor``.
"""
from __future__ import annotations

import abc
from pathlib import Path

import numpy as np

from .predictor2d import Predictor2D

__all__ = ["DetectorPredictor", "MediaPipePredictor", "DlibPredictor", "FaceAlignmentPredictor"]


def _require(module: str, pipeline: str):
    import importlib

    try:
        return importlib.import_module(module)
    except ImportError as e:
        raise ImportError(f"the {pipeline!r} pipeline needs the third-party package {module!r}, which is not installed "
                          "(it is not part of this build; install it, or hand its landmarks to PrecomputedPredictor)") from e


def depth_at(image: np.ndarray, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
    """The depth plane under integer-truncated, image-clamped landmark positions - the per-landmark score of the
    dlib and face_alignment wrappers (dlibpredictor.py:68, face_alignmentpredictor.py:48)."""
    h, w = image.shape[:2]
    r = np.clip(np.asarray(rows), 0, h - 1).astype(int)
    c = np.clip(np.asarray(cols), 0, w - 1).astype(int)
    return image[r, c, 3]


class DetectorPredictor(Predictor2D):
    """One detector call per rendered view.  Subclasses implement ``detect(rgb_u8, view) -> [NL,3] | None`` returning
    (row, col, score) per landmark, or None when nothing was found in that view."""

    n_landmarks: int = 0

    def get_lm_count(self) -> int:
        return self.n_landmarks

    @abc.abstractmethod
    def detect(self, rgb_u8: np.ndarray, view: np.ndarray):
        """rgb_u8 uint8 [H,W,3], view float32 [H,W,4] (RGB + depth in [0,1]) -> float array [NL,3] or None."""

    def predict_landmarks_from_images(self, image_stack: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        n_views = int(image_stack.shape[0])
        nl = self.get_lm_count()
        landmarks = np.full((nl, n_views, 3), np.nan, dtype=np.float32)
        valid = np.zeros(n_views, dtype=bool)
        for v in range(n_views):
            view = image_stack[v]
            found = self.detect((view[..., :3] * 255).astype(np.uint8), view)
            if found is None:
                continue
            found = np.asarray(found, dtype=np.float32)
            if found.shape != (nl, 3):
                raise RuntimeError(f"detector returned {found.shape}, expected ({nl}, 3)")
            landmarks[:, v, :] = found
            valid[v] = True
        return landmarks, valid


class MediaPipePredictor(DetectorPredictor):
    """MediaPipe face landmarker, 478 landmarks: (y * h, x * w, -z * w) per normalised landmark
    (mediapipepredictor.py:26-48).  ``model_asset_path``: the ``face_landmarker.task`` bundle
    (the reference ships it as prediction/models/2023-07-09_face_landmarker.task)."""

    n_landmarks = 478

    def __init__(self, model_asset_path=None):
        super().__init__()
        self._mp = _require("mediapipe", "mediapipe")
        python = _require("mediapipe.tasks.python", "mediapipe")
        vision = _require("mediapipe.tasks.python.vision", "mediapipe")
        path = Path(model_asset_path) if model_asset_path else Path(__file__).parent / "models" / "2023-07-09_face_landmarker.task"
        if not path.is_file():
            raise FileNotFoundError(f"MediaPipe model bundle {path} not found (pass model_asset_path=...)")
        options = vision.FaceLandmarkerOptions(base_options=python.BaseOptions(model_asset_path=str(path)),
                                               running_mode=vision.RunningMode.IMAGE, output_face_blendshapes=False,
                                               output_facial_transformation_matrixes=False, num_faces=1)
        self.detector = vision.FaceLandmarker.create_from_options(options)

    def detect(self, rgb_u8, view):
        mp = self._mp
        result = self.detector.detect(mp.Image(image_format=mp.ImageFormat.SRGB, data=np.ascontiguousarray(rgb_u8)))
        if not result.face_landmarks:
            return None
        h, w = rgb_u8.shape[:2]
        face = result.face_landmarks[0]
        return np.array([[lm.y * h, lm.x * w, -lm.z * w] for lm in face], dtype=np.float32)


class DlibPredictor(DetectorPredictor):
    """dlib frontal face detector + 68-point shape predictor: (y, x, depth under the point)
    (dlibpredictor.py:33-73).  ``shape_predictor_path``: ``shape_predictor_68_face_landmarks.dat`` (the reference
    downloads it on first use; there is no network here, so it must be on disk)."""

    n_landmarks = 68

    def __init__(self, shape_predictor_path=None):
        super().__init__()
        dlib = _require("dlib", "dlib")
        self._cv2 = _require("cv2", "dlib")
        path = Path(shape_predictor_path) if shape_predictor_path else Path(__file__).parent / "models" / "shape_predictor_68_face_landmarks.dat"
        if not path.is_file():
            raise FileNotFoundError(f"dlib shape predictor {path} not found (pass shape_predictor_path=...)")
        self.detector = dlib.get_frontal_face_detector()
        self.predictor = dlib.shape_predictor(str(path))

    def detect(self, rgb_u8, view):
        gray = self._cv2.cvtColor(rgb_u8, self._cv2.COLOR_BGR2GRAY)  # the reference's channel order, kept (:55)
        rects = self.detector(gray, 1)
        if len(rects) == 0:
            return None
        shape = self.predictor(rgb_u8, rects[0])
        xy = np.array([[shape.part(j).x, shape.part(j).y] for j in range(shape.num_parts)], dtype=np.float32)
        out = np.zeros((self.n_landmarks, 3), np.float32)
        k = min(len(xy), self.n_landmarks)
        out[:k, 0], out[:k, 1] = xy[:k, 1], xy[:k, 0]
        out[:k, 2] = view[np.minimum(255, xy[:k, 1].astype(int)), np.minimum(255, xy[:k, 0].astype(int)), 3]
        return out


class FaceAlignmentPredictor(DetectorPredictor):
    """``face_alignment`` 2-D landmarks (blazeface detector), 68 points: (y, x, depth under the point)
    (face_alignmentpredictor.py:24-52).  The detector network runs on ``device`` through that library's own
    PyTorch code; only the views and the fusion are this build's kernels."""

    n_landmarks = 68

    def __init__(self, device: str | None = None):
        super().__init__()
        fa = _require("face_alignment", "face_alignment")
        import torch

        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        kw = {"dtype": torch.bfloat16} if device.startswith("cuda") else {}
        self.fa = fa.FaceAlignment(fa.LandmarksType.TWO_D, face_detector="blazeface", device=device, **kw)

    def detect(self, rgb_u8, view):
        pred = self.fa.get_landmarks_from_image(rgb_u8, return_bboxes=False, return_landmark_score=False)
        if pred is None:
            return None
        x, y = np.asarray(pred[0])[:, 0], np.asarray(pred[0])[:, 1]
        return np.stack([y, x, depth_at(view, y, x)], axis=1)
