"""Quick per-stage timing of the C-ABI on the headline scene (development helper)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import neural_renderer_amd as nr
B = int(os.environ.get('B', 64)); S = int(os.environ.get('S', 256)); ts = 2
dev = torch.device('cuda', 0)
faces, textures = bench.build_scene(dev, B, 0, B, S, ts)
with torch.no_grad():
    rgb0, alpha0, depth0 = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
    gen = torch.Generator(device='cpu').manual_seed(1234)
    g_rgb = (2 * (rgb0 - torch.rand(rgb0.shape, generator=gen).to(dev))).contiguous()
    g_alpha = (2 * (alpha0 - torch.rand(alpha0.shape, generator=gen).to(dev))).contiguous()
    g_depth = (2 * (depth0 / 100.0 - torch.rand(depth0.shape, generator=gen).to(dev)) / 100.0).contiguous()
st = bench.time_stages(faces, textures, S, 1e-3, g_rgb, g_alpha, g_depth, int(os.environ.get('ITERS', 10)),
                       int(os.environ.get('NR_STAGE_FLAGS', 0)))
print(os.environ.get('TAG', ''), json.dumps({k: round(v, 1) for k, v in st.items()}))
