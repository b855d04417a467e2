"""Drop-in alias: `import neural_renderer` resolves to the MI355X-native implementation, so scripts written
against hiroharu-kato/neural_renderer (examples 1-4) keep their import line."""
from neural_renderer_amd import *  # noqa: F401,F403
from neural_renderer_amd import __version__, rasterize as _rasterize_module  # noqa: F401
from neural_renderer_amd import distributed, graph  # noqa: F401  (not in the reference: multi-GPU and graph helpers)
