"""Pipeline factory (reference src/mvlm/pipeline/__init__.py:15-43)."""
__all__ = ["Pipeline", "BU3DFEPipeline", "DTU3DPipeline", "create_pipeline", "pipeline_from_config"]

from .general_pipeline import Pipeline
from .paulsen_pipeline import BU3DFEPipeline, DTU3DPipeline

_THIRD_PARTY = ("mediapipe", "dlib", "face_alignment")


def create_pipeline(name: str, **kwargs):
    """Create a pipeline by name ("bu3dfe", "dtu3d"; case-insensitive).

    The reference also knows "mediapipe", "dlib" and "face_alignment"; their 2-D
    detectors are third-party packages outside this build's scope - plug such a
    detector into ``Pipeline.predictor_2d`` (see prediction.PrecomputedPredictor).
    """
    name = name.lower()
    if name == "bu3dfe":
        return BU3DFEPipeline(**kwargs)
    if name == "dtu3d":
        return DTU3DPipeline(**kwargs)
    if name in _THIRD_PARTY:
        raise ValueError(f"Pipeline {name!r} wraps a third-party 2-D detector that this build does not ship; "
                         "assign your detector to Pipeline.predictor_2d instead")
    raise ValueError(f"Unknown pipeline: {name}")


def pipeline_from_config(config, **kwargs):
    """Build a pipeline from a Deep-MVLM JSON config (path or dict), see mvlm_amd.config."""
    from ..config import load_config

    cfg = load_config(config)
    return cfg.build_pipeline(**kwargs)
