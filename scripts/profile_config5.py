#!/usr/bin/env python3
"""BASELINE config 5 (one 655 360-face mesh, 1024x1024, texture_size 8, rgb + alpha + depth) alone, for
`rocprofv3 --kernel-trace --stats -- python scripts/profile_config5.py` (development helper)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import neural_renderer_amd as nr  # noqa: E402
from test_hip_parity import icosphere, project_mesh  # noqa: E402

dev = torch.device('cuda', 0)
rng = np.random.default_rng(1234)
v0, f0 = icosphere(7)
v = v0 * (0.6 + 0.02 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
faces = torch.from_numpy(project_mesh(v.astype(np.float32), f0, [0.0, 0.0, -2.4])[None]).to(dev).requires_grad_(True)
textures = torch.rand((1, faces.shape[1], 8, 8, 8, 3), device=dev, requires_grad=True)
g = None


def step():
    global g
    faces.grad = None
    textures.grad = None
    outs = nr.Rasterize(1024, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
    if g is None:
        g = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(list(outs), g)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(os.environ.get('TAG', ''), 'C5 fwd+bwd ms', round((time.perf_counter() - t0) / 10 * 1e3, 4))
