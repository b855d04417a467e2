import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import neural_renderer_amd as nr
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
v, f = bench.load_teapot()
B = 64
vertices = torch.from_numpy(v).to(dev)[None].repeat(B, 1, 1).requires_grad_(True)
faces = torch.from_numpy(f).to(dev)[None].repeat(B, 1, 1)
textures = torch.ones((B, f.shape[0], 2, 2, 2, 3), device=dev, requires_grad=True)
r = nr.Renderer(); r.anti_aliasing = False
r.eye = torch.tensor([nr.get_points_from_angles(2.732, 30., 360.0 * i / B) for i in range(B)], dtype=torch.float32, device=dev)
def step(which):
    vertices.grad = None; textures.grad = None
    img = r.render(vertices, faces, textures) if which == 'rgb' else r.render_silhouettes(vertices, faces)
    img.square().sum().backward()
for which in ('sil', 'rgb'):
    for _ in range(3): step(which)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5): step(which)
        torch.cuda.synchronize()
    print('=====', which)
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=70))
