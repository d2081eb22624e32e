__all__ = ["HipRenderer3D", "HipEstimator3D", "Mesh", "load_obj", "view_rotations"]

from .mesh_io import Mesh, load_obj
from .render3d import HipRenderer3D, view_rotations
from .estimator3d import HipEstimator3D
