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

__all__ = ["MVLMConfig", "load_config", "default_config", "REFERENCE_CONFIGS"]


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
    write_renderings: bool = False           # process_3d.write_renderings: PNG dump of the rendered views
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
        if self.write_renderings:
            kwargs.setdefault("render_image_stack", True)  # general_pipeline.py:133-146, next to the scan unless a folder is given
        if self.n_gpu > 1:
            kwargs.setdefault("n_gpus", self.n_gpu)
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
    cfg.write_renderings = bool(p3.get("write_renderings", cfg.write_renderings))
    for k in list(cfg.angles):
        if k in p3:
            cfg.angles[k] = p3[k]
    pa = d.get("pre-align", {})
    for k in list(cfg.pre_align):
        if k in pa:
            cfg.pre_align[k] = pa[k]
    cfg.validate()
    return cfg


# What each of the reference's configs/*.json says at inference, as differences from the common block that
# default_config() writes out (configuration values, checked key by key against the files in
# tests/test_host_logic.py when /root/reference is present).  n_views 96 and batch_size 8 unless listed.
REFERENCE_CONFIGS = {
    "BU_3DFE-RGB+depth": dict(dataset="BU_3DFE", mode="RGB+depth", n_views=8),
    "BU_3DFE-RGB": dict(dataset="BU_3DFE", mode="RGB", n_views=64, batch_size=4,
                        pre_align=dict(align_center_of_mass=True, scale=10)),
    "BU_3DFE-RGB_train_test": dict(dataset="BU_3DFE", mode="RGB",
                                   process_3d=dict(min_x_angle=-90, max_x_angle=20, min_y_angle=-60, max_y_angle=60,
                                                   min_z_angle=-40, max_z_angle=40)),
    "BU_3DFE-depth": dict(dataset="BU_3DFE", mode="depth", pre_align=dict(align_center_of_mass=True, scale=20)),
    "BU_3DFE-geometry+depth": dict(dataset="BU_3DFE", mode="geometry+depth"),
    "BU_3DFE-geometry": dict(dataset="BU_3DFE", mode="geometry"),
    "DTU3D-RGB+depth": dict(dataset="DTU3D", mode="RGB+depth"),
    "DTU3D-RGB": dict(dataset="DTU3D", mode="RGB"),
    "DTU3D-RGB_Artec3D": dict(dataset="DTU3D", mode="RGB", process_3d=dict(write_renderings=True),
                              pre_align=dict(rot_x=-90, write_pre_aligned=True)),
    "DTU3D-RGB_BU3DFE_RAW": dict(dataset="DTU3D", mode="RGB", pre_align=dict(rot_x=-35)),
    "DTU3D-RGB_infinite": dict(dataset="DTU3D", mode="RGB", process_3d=dict(write_renderings=True),
                               pre_align=dict(scale=800, write_pre_aligned=True)),
    "DTU3D-depth-MRI": dict(dataset="DTU3D", mode="depth", process_3d=dict(write_renderings=True),
                            pre_align=dict(align_center_of_mass=True, rot_z=180, write_pre_aligned=True)),
    "DTU3D-depth": dict(dataset="DTU3D", mode="depth"),
    "DTU3D-depth_infinite": dict(dataset="DTU3D", mode="depth", process_3d=dict(write_renderings=True),
                                 pre_align=dict(scale=800, write_pre_aligned=True)),
    "DTU3D-geometry+depth": dict(dataset="DTU3D", mode="geometry+depth"),
    "DTU3D-geometry+depth_BU3DFE_RAW": dict(dataset="DTU3D", mode="geometry+depth", pre_align=dict(rot_x=-35)),
    "DTU3D-geometry": dict(dataset="DTU3D", mode="geometry"),
}


def default_config(dataset: str, image_channels: str | None = None, n_views: int | None = None) -> dict:
    """A config dict with the reference's schema and the inference-relevant content of the file of that name:
    ``default_config("BU_3DFE", "depth")`` == ``default_config("BU_3DFE-depth")`` has what configs/BU_3DFE-depth.json
    has (pre-align: centre of mass + scale 20, 96 views); any of the 17 file stems is accepted as the first argument.
    ``n_views`` overrides the file's view count (BASELINE.json quotes its configurations at view counts of its own)."""
    stem = dataset if image_channels is None else f"{dataset}-{image_channels}"
    if stem not in REFERENCE_CONFIGS:
        raise ValueError(f"no reference config named {stem}.json; known: {sorted(REFERENCE_CONFIGS)}")
    spec = REFERENCE_CONFIGS[stem]
    dataset, image_channels = spec["dataset"], spec["mode"]
    name = {"DTU3D": "MVLMModel_DTU3D", "BU_3DFE": "MVLMModel_BU_3DFE"}[dataset]
    process_3d = {"filter_view_lines": "quantile", "heatmap_max_quantile": 0.5, "heatmap_abs_threshold": 0.5,
                  "write_renderings": False, "off_screen_rendering": True, "min_x_angle": -40, "max_x_angle": 40,
                  "min_y_angle": -80, "max_y_angle": 80, "min_z_angle": -20, "max_z_angle": 20}
    process_3d.update(spec.get("process_3d", {}))
    pre_align = {"align_center_of_mass": False, "rot_x": 0, "rot_y": 0, "rot_z": 0, "scale": 1,
                 "write_pre_aligned": False}
    pre_align.update(spec.get("pre_align", {}))
    return {
        "name": name,
        "n_gpu": 1,
        "arch": {"type": "MVLMModel", "args": {"n_landmarks": 73 if dataset == "DTU3D" else 84, "n_features": 256,
                                                "dropout_rate": 0.2, "image_channels": image_channels}},
        "data_loader": {"type": "FaceDataLoader", "args": {"heatmap_size": 256, "image_size": 256,
                                                           "image_channels": image_channels,
                                                           "n_views": int(n_views or spec.get("n_views", 96)),
                                                           "batch_size": spec.get("batch_size", 8)}},
        "process_3d": process_3d,
        "pre-align": pre_align,
    }
