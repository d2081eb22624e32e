__all__ = ["Predictor2D", "HipPaulsenModel", "BU3DFEPredictor", "DTU3DPredictor", "PrecomputedPredictor", "DetectorPredictor",
           "MediaPipePredictor", "DlibPredictor", "FaceAlignmentPredictor"]

from .predictor2d import Predictor2D
from .paulsenpredictor import HipPaulsenModel, BU3DFEPredictor, DTU3DPredictor
from .precomputed import PrecomputedPredictor
from .thirdparty import DetectorPredictor, DlibPredictor, FaceAlignmentPredictor, MediaPipePredictor
