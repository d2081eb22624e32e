"""Pipeline factory (reference src/mvlm/pipeline/__init__.py:15-43)."""
__all__ = ["Pipeline", "BU3DFEPipeline", "DTU3DPipeline", "MediaPipePipeline", "DlibPipeline", "FaceAlignmentPipeline",
           "create_pipeline", "pipeline_from_config"]

from .general_pipeline import Pipeline
from .paulsen_pipeline import BU3DFEPipeline, DTU3DPipeline
from .thirdparty_pipelines import DlibPipeline, FaceAlignmentPipeline, MediaPipePipeline


def create_pipeline(name: str, **kwargs):
    """Create a pipeline by name, case-insensitive: "bu3dfe", "dtu3d" (the landmark network of this build), and the
    reference's "mediapipe", "dlib", "face_alignment", whose 2-D detectors are third-party packages: those
    pipelines render and fuse on the GPU around the detector and raise ImportError when its package is missing."""
    name = name.lower()
    if name == "mediapipe":
        return MediaPipePipeline(**kwargs)
    if name == "bu3dfe":
        return BU3DFEPipeline(**kwargs)
    if name == "dlib":
        return DlibPipeline(**kwargs)
    if name == "dtu3d":
        return DTU3DPipeline(**kwargs)
    if name == "face_alignment":
        return FaceAlignmentPipeline(**kwargs)
    raise ValueError(f"Unknown pipeline: {name}")


def pipeline_from_config(config, **kwargs):
    """Build a pipeline from a Deep-MVLM JSON config (path or dict), see mvlm_amd.config."""
    from ..config import load_config

    cfg = load_config(config)
    return cfg.build_pipeline(**kwargs)
