"""BU-3DFE / DTU-3D pipelines (reference src/mvlm/pipeline/paulsen_pipeline.py:7-15)."""
from __future__ import annotations

from ..prediction import BU3DFEPredictor, DTU3DPredictor
from .general_pipeline import Pipeline

__all__ = ["BU3DFEPipeline", "DTU3DPipeline"]

_PREDICTOR_KEYS = ("weights", "image_mode", "selection_method", "batch_size", "device_batch", "model_dir", "precision", "n_gpus")


def _split(kwargs):
    pk = {k: kwargs.pop(k) for k in _PREDICTOR_KEYS if k in kwargs}
    return pk, kwargs


class BU3DFEPipeline(Pipeline):
    def __init__(self, *args, **kwargs):
        pk, kwargs = _split(kwargs)
        super().__init__(*args, **kwargs)
        self.predictor_2d = BU3DFEPredictor(device=self.device, verbose=self.verbose, **pk)


class DTU3DPipeline(Pipeline):
    def __init__(self, *args, **kwargs):
        pk, kwargs = _split(kwargs)
        super().__init__(*args, **kwargs)
        self.predictor_2d = DTU3DPredictor(device=self.device, verbose=self.verbose, **pk)
