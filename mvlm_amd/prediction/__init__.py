__all__ = ["Predictor2D", "HipPaulsenModel", "BU3DFEPredictor", "DTU3DPredictor", "PrecomputedPredictor"]

from .predictor2d import Predictor2D
from .paulsenpredictor import HipPaulsenModel, BU3DFEPredictor, DTU3DPredictor
from .precomputed import PrecomputedPredictor
