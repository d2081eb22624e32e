"""Reader for the Deep-MVLM JSON configs (reference configs/*.json).

No live code in the reference reads these files (the only
consumer is the never-instantiated ``Utils3D``, src/mvlm/utils/utils3d.py:137-145),
so "configs stay drop-in" means this build parses them itself.  Only the keys that
matter at inference are honoured; training keys are ignored.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path

from . import arch

__all__ = ["MVLMConfig", "load_config", "default_config"]


@dataclass
class MVLMConfig:
    name: str = "MVLMModel_DTU3D"            # weight-file prefix (paulsenpredictor.py:95)
    n_gpu: int = 1
    n_landmarks: int = 73
    n_features: int = 256
    dropout_rate: float = 0.2
    image_channels: str = "RGB"
    heatmap_size: int = 256
    image_size: int = 256
    n_views: int = 96
    batch_size: int = 8
    filter_view_lines: str = "quantile"
    heatmap_max_quantile: float = 0.5
    heatmap_abs_threshold: float = 0.5
    off_screen_rendering: bool = True
    angles: dict = field(default_factory=lambda: dict(min_x_angle=-40, max_x_angle=40, min_y_angle=-80,
                                                      max_y_angle=80, min_z_angle=-20, max_z_angle=20))
    pre_align: dict = field(default_factory=lambda: dict(align_center_of_mass=False, rot_x=0, rot_y=0, rot_z=0,
                                                         scale=1, write_pre_aligned=False))

    @property
    def in_channels(self) -> int:
        return arch.IMAGE_CHANNELS[self.image_channels]

    def validate(self) -> None:
        if self.image_channels not in arch.IMAGE_CHANNELS:
            raise ValueError("Image channels should be: geometry, RGB, depth, RGB+depth or geometry+depth")
        if self.n_features != arch.N_FEATURES:
            raise ValueError("only n_features=256 networks exist (all reference configs)")
        if self.image_size != 256 or self.heatmap_size != 256:
            raise ValueError("the hot path is built for 256x256 views and heatmaps (all reference configs)")
        if self.filter_view_lines not in ("quantile", "absolute"):
            raise ValueError(f"Unknown mode for line matching in Estimator: {self.filter_view_lines}")
        if self.name not in ("MVLMModel_BU_3DFE", "MVLMModel_DTU3D"):
            raise ValueError(f"unknown model name {self.name!r}")

    def build_pipeline(self, n_views: int | None = None, **kwargs):
        """Pipeline configured like this file: landmark model, image mode, view count,
        pose ranges and line filter."""
        from .pipeline import BU3DFEPipeline, DTU3DPipeline

        self.validate()
        cls = BU3DFEPipeline if self.name == "MVLMModel_BU_3DFE" else DTU3DPipeline
        pipe = cls(n_views=n_views or self.n_views, image_mode=self.image_channels, **kwargs)
        if pipe.get_lm_count() != self.n_landmarks:
            raise ValueError(f"config asks for {self.n_landmarks} landmarks, {self.name} has {pipe.get_lm_count()}")
        for k, v in self.angles.items():
            setattr(pipe.renderer_3d, k, v)
        pipe.estimator_3d.mode = self.filter_view_lines
        pipe.estimator_3d.threshold_quantile = self.heatmap_max_quantile
        pipe.estimator_3d.threshold_absolute = self.heatmap_abs_threshold
        pipe.pre_align = dict(self.pre_align)
        if "geometry" in self.image_channels:
            pipe.renderer_3d.shading = "geometry"  # plane 0 carries the shaded geometry
        return pipe


def load_config(src) -> MVLMConfig:
    if isinstance(src, MVLMConfig):
        return src
    d = json.loads(Path(src).read_text()) if not isinstance(src, dict) else src
    cfg = MVLMConfig()
    cfg.name = d.get("name", cfg.name)
    cfg.n_gpu = int(d.get("n_gpu", cfg.n_gpu))
    a = d.get("arch", {}).get("args", {})
    cfg.n_landmarks = int(a.get("n_landmarks", cfg.n_landmarks))
    cfg.n_features = int(a.get("n_features", cfg.n_features))
    cfg.dropout_rate = float(a.get("dropout_rate", cfg.dropout_rate))
    cfg.image_channels = a.get("image_channels", cfg.image_channels)
    dl = d.get("data_loader", {}).get("args", {})
    cfg.heatmap_size = int(dl.get("heatmap_size", cfg.heatmap_size))
    cfg.image_size = int(dl.get("image_size", cfg.image_size))
    cfg.n_views = int(dl.get("n_views", cfg.n_views))
    cfg.batch_size = int(dl.get("batch_size", cfg.batch_size))
    p3 = d.get("process_3d", {})
    cfg.filter_view_lines = p3.get("filter_view_lines", cfg.filter_view_lines)
    cfg.heatmap_max_quantile = float(p3.get("heatmap_max_quantile", cfg.heatmap_max_quantile))
    cfg.heatmap_abs_threshold = float(p3.get("heatmap_abs_threshold", cfg.heatmap_abs_threshold))
    cfg.off_screen_rendering = bool(p3.get("off_screen_rendering", cfg.off_screen_rendering))
    for k in list(cfg.angles):
        if k in p3:
            cfg.angles[k] = p3[k]
    pa = d.get("pre-align", {})
    for k in list(cfg.pre_align):
        if k in pa:
            cfg.pre_align[k] = pa[k]
    cfg.validate()
    return cfg


def default_config(dataset: str, image_channels: str, n_views: int = 96) -> dict:
    """A config dict with the reference's schema, e.g. default_config("DTU3D", "RGB")
    has the inference-relevant content of configs/DTU3D-RGB.json."""
    name = {"DTU3D": "MVLMModel_DTU3D", "BU_3DFE": "MVLMModel_BU_3DFE"}[dataset]
    return {
        "name": name,
        "n_gpu": 1,
        "arch": {"type": "MVLMModel", "args": {"n_landmarks": 73 if dataset == "DTU3D" else 84, "n_features": 256,
                                                "dropout_rate": 0.2, "image_channels": image_channels}},
        "data_loader": {"type": "FaceDataLoader", "args": {"heatmap_size": 256, "image_size": 256,
                                                           "image_channels": image_channels, "n_views": n_views,
                                                           "batch_size": 8}},
        "process_3d": {"filter_view_lines": "quantile", "heatmap_max_quantile": 0.5, "heatmap_abs_threshold": 0.5,
                       "off_screen_rendering": True, "min_x_angle": -40, "max_x_angle": 40, "min_y_angle": -80,
                       "max_y_angle": 80, "min_z_angle": -20, "max_z_angle": 20},
        "pre-align": {"align_center_of_mass": False, "rot_x": 0, "rot_y": 0, "rot_z": 0, "scale": 1,
                      "write_pre_aligned": False},
    }
