#!/usr/bin/env python3
"""BASELINE config 4 (per-GPU share: 64 random ~5k-face meshes, 10 240 faces with fill_back, texture_size 4, 256x256 RGB)
alone, for `rocprofv3 --kernel-trace --stats -- python scripts/profile_config4.py` (development helper)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import neural_renderer_amd as nr  # noqa: E402
from test_hip_parity import icosphere, project_mesh  # noqa: E402

dev = torch.device('cuda', 0)
rng = np.random.default_rng(1234)
v0, f0 = icosphere(4)
batch = []
for _ in range(64):
    v = v0 * (0.55 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
    q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    batch.append(project_mesh((v @ q).astype(np.float32), f0, [0.3, 0.4, -2.6]))
faces = torch.from_numpy(np.stack(batch)).to(dev).requires_grad_(True)
textures = torch.rand((64, faces.shape[1], 4, 4, 4, 3), device=dev, requires_grad=True)
g = None
for it in range(int(os.environ.get('ITERS', 12))):
    faces.grad = None
    textures.grad = None
    rgb, _, _ = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, False, False)(faces, textures)
    if g is None:
        g = torch.randn_like(rgb)
    rgb.backward(g)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for it in range(20):
    faces.grad = None
    textures.grad = None
    rgb, _, _ = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, False, False)(faces, textures)
    rgb.backward(g)
torch.cuda.synchronize()
print(os.environ.get('TAG', ''), 'C4 fwd+bwd ms', round((time.perf_counter() - t0) / 20 * 1e3, 4))
