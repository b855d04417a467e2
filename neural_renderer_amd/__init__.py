"""neural_renderer_amd -- MI355X-native differentiable mesh rasterizer behind the API of hiroharu-kato/neural_renderer.

The public names are the reference's (neural_renderer/__init__.py:1-16); `import neural_renderer` is an alias package."""
# the operator and its wrappers (HIP kernels behind include/nr_hip.h)
from .rasterize import (Rasterize, rasterize, rasterize_depth, rasterize_rgbad, rasterize_silhouettes,
                        use_unsafe_rasterizer, use_graph_replay, clear_workspace_cache)
from .renderer import Renderer
# geometry / lighting glue in front of the rasterizer
from .cross import cross
from .get_points_from_angles import get_points_from_angles
from .lighting import lighting
from .look import look
from .look_at import look_at
from .perspective import perspective
from .vertices_to_faces import vertices_to_faces
# meshes, files, optimiser
from .load_obj import load_obj
from .mesh import Mesh
from .optimizers import Adam
from .save_obj import save_obj
# not in the reference: multi-GPU helpers and the captured-graph helper for fixed-shape loops
from . import distributed, graph

# NR_BACKWARD_ON_CALLER_THREAD=1 (opt-in, read once here): graph.backward_on_caller_thread() for the importing thread -- torch's
# autograd then runs CUDA nodes on the thread that calls backward() instead of handing them to its device thread, which costs a
# host-bound loop ~80 us per step (bench.py shard_rows).  One process per GPU gains nothing from the device thread.
import os as _os
if _os.environ.get('NR_BACKWARD_ON_CALLER_THREAD', '0') not in ('', '0'):
    graph.backward_on_caller_thread()

# the C ABI's version (include/nr_hip.h NR_VERSION = major * 100 + minor), checked against the loaded library by _lib.load()
__version__ = '0.6.0'
__all__ = ['Rasterize', 'rasterize', 'rasterize_depth', 'rasterize_rgbad', 'rasterize_silhouettes', 'use_unsafe_rasterizer', 'use_graph_replay',
           'clear_workspace_cache',
           'Renderer', 'cross', 'get_points_from_angles', 'lighting', 'look', 'look_at', 'perspective', 'vertices_to_faces',
           'load_obj', 'Mesh', 'Adam', 'save_obj']
