"""End-to-end Renderer.render / render_silhouettes timing (development helper): how much time the torch glue
(fill_back, lighting, look_at, perspective, vertices_to_faces, flip / permute / avg_pool and their backward)
adds around the rasterizer."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import neural_renderer_amd as nr
dev = torch.device('cuda', 0)
v, f = bench.load_teapot()
B = 64
vertices = torch.from_numpy(v).to(dev)[None].repeat(B, 1, 1).requires_grad_(True)
faces = torch.from_numpy(f).to(dev)[None].repeat(B, 1, 1)
textures = torch.ones((B, f.shape[0], 2, 2, 2, 3), device=dev, requires_grad=True)
r = nr.Renderer()
r.eye = torch.tensor([nr.get_points_from_angles(2.732, 30., 360.0 * i / B) for i in range(B)], dtype=torch.float32, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
from neural_renderer_amd import frontend
_fusable = frontend.fusable
for fused, aa in ((True, False), (False, False), (True, True), (False, True)):
    frontend.fusable = _fusable if fused else (lambda *a: False)   # False: module-by-module torch front-end
    r.anti_aliasing = aa
    def rgb():
        vertices.grad = None; textures.grad = None
        img = r.render(vertices, faces, textures); img.square().sum().backward()
    def sil():
        vertices.grad = None
        img = r.render_silhouettes(vertices, faces); img.square().sum().backward()
    print(json.dumps({'fused_frontend': fused, 'anti_aliasing': aa, 'render_fwd_bwd_ms': round(t(rgb), 3), 'silhouettes_fwd_bwd_ms': round(t(sil), 3)}))
