__all__ = ["HipRenderer3D", "HipEstimator3D", "Mesh", "load_obj", "load_mesh", "find_texture", "view_rotations"]

from .mesh_io import Mesh, find_texture, load_mesh, load_obj
from .render3d import HipRenderer3D, view_rotations
from .estimator3d import HipEstimator3D
