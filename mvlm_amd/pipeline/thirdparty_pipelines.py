"""Pipelines around third-party 2-D detectors (reference src/mvlm/pipeline/mediapipe_pipeline.py:7-10,
dlib_pipeline.py:7-12, face_alignment_pipeline.py:9-12): the views come from the HIP rasteriser, the
detector's landmarks go through the HIP ray / consensus / snap stages.  The detector libraries are not part
of this build; constructing one of these without its library raises ImportError."""
from __future__ import annotations

from ..prediction.thirdparty import DlibPredictor, FaceAlignmentPredictor, MediaPipePredictor
from .general_pipeline import Pipeline

__all__ = ["MediaPipePipeline", "DlibPipeline", "FaceAlignmentPipeline"]


class MediaPipePipeline(Pipeline):
    def __init__(self, *args, model_asset_path=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.predictor_2d = MediaPipePredictor(model_asset_path=model_asset_path)


class DlibPipeline(Pipeline):
    def __init__(self, *args, shape_predictor_path=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.predictor_2d = DlibPredictor(shape_predictor_path=shape_predictor_path)
        # scores are depth values, not heatmap maxima: an absolute cut instead of the median (dlib_pipeline.py:11-12)
        self.estimator_3d.mode = "absolute"
        self.estimator_3d.threshold_absolute = 0.1


class FaceAlignmentPipeline(Pipeline):
    def __init__(self, *args, detector_device=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.predictor_2d = FaceAlignmentPredictor(device=detector_device)
