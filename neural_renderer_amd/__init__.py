"""neural_renderer_amd -- MI355X-native differentiable mesh rasterizer behind the API of
hiroharu-kato/neural_renderer (reference neural_renderer/__init__.py:1-16)."""
from .cross import cross
from .get_points_from_angles import get_points_from_angles
from .lighting import lighting
from .load_obj import load_obj
from .look import look
from .look_at import look_at
from .mesh import Mesh
from .optimizers import Adam
from .perspective import perspective
from .rasterize import (
    rasterize_rgbad, rasterize, rasterize_silhouettes, rasterize_depth, use_unsafe_rasterizer, Rasterize)
from .renderer import Renderer
from .save_obj import save_obj
from .vertices_to_faces import vertices_to_faces

__version__ = '0.1.0'
