"""Timing probe (development helper): a few screen-filling faces on top of the teapot -- ground-plane-like geometry."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import bench, helpers as H
import neural_renderer_amd as nr
dev = torch.device('cuda', 0)
B, S = 64, 256
faces, textures = bench.build_scene(dev, B, 0, B, S, 2)
n_big = int(os.environ.get('NBIG', 2))
if n_big:
    quad = np.array([[[-0.95, -0.9, 4.0], [0.95, -0.9, 4.0], [0.9, 0.95, 5.0]], [[-0.95, -0.9, 4.0], [0.9, 0.95, 5.0], [-0.9, 0.9, 5.0]]], np.float32)[:n_big]
    big = torch.from_numpy(quad).to(dev)[None].repeat(B, 1, 1, 1)
    faces = torch.cat((faces, big), dim=1).contiguous()
    textures = torch.cat((textures, torch.rand((B, n_big, 2, 2, 2, 3), device=dev)), dim=1).contiguous()
faces.requires_grad_(True); textures.requires_grad_(True)
g = None
def step():
    global g
    faces.grad = None; textures.grad = None
    out = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
    if g is None: g = [torch.randn_like(o) for o in out]
    torch.autograd.backward(list(out), g)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print('NBIG', n_big, 'fwd+bwd ms', round((time.perf_counter() - t0) / 10 * 1e3, 3))
def fwd():
    with torch.no_grad(): nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
for _ in range(3): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fwd()
torch.cuda.synchronize(); print('   fwd ms', round((time.perf_counter() - t0) / 10 * 1e3, 3))
