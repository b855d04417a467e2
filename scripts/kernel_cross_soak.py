"""Development soak without the oracle: K6's two band kernels against each other on random scenes at sizes the O(pixels x faces)
oracle does not reach quickly -- rasters 17 ... 1024 incl. odd ones, 1 ... 6 images, soups / spheres / teapots, every output mode.
Exact mode: the kernels form the same float terms and add them in double -- grad_faces must agree in (nearly) every bit.  Default
mode: two roundings of the same terms -- agreement to ~1e-4 of the largest gradient; a dropped or doubled pixel shows as 1e-2.
    N=200 SEED=1 python scripts/kernel_cross_soak.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import abi
import helpers as H
from test_hip_parity import icosphere, project_mesh

rng = np.random.default_rng(int(os.environ.get('SEED', 1)))
worst = {'default': 0.0, 'exact_bits': 0, 'exact': 0.0}
bad = []
for it in range(int(os.environ.get('N', 100))):
    S = int(rng.choice([int(x) for x in os.environ["SIZES"].split()] if os.environ.get("SIZES") else [17, 31, 32, 33, 48, 64, 100, 128, 200, 255, 256, 257, 320, 500, 512, 513, 640, 777, 1000, 1024]))
    B = int(rng.integers(1, 7)) if S <= 512 else int(rng.integers(1, 3))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        faces = H.random_scene(rng, B, int(rng.choice([5, 60, 400, 2000])), spread=float(rng.choice([0.4, 0.9, 1.5])),
                               size=float(rng.choice([0.03, 0.2, 1.0])))
    elif kind == 1:
        v0, f0 = icosphere(int(rng.integers(1, 5)))
        faces = np.stack([project_mesh((v0 * (0.5 + 0.2 * rng.normal(size=(v0.shape[0], 1)))).astype(np.float32), f0,
                                       [float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.6, 0.6)), -float(rng.uniform(1.2, 3.0))])
                          for _ in range(B)])
    else:
        faces = H.teapot_views(64, S)[0][rng.integers(0, 64, B)]
    if rng.uniform() < 0.3:   # snap some vertices onto pixel centres
        q = (np.round((faces[..., :2] * S + S - 1) / 2) * 2 + 1 - S) / S
        m = rng.uniform(size=faces[..., :2].shape) < 0.3
        faces[..., :2] = np.where(m, q, faces[..., :2]).astype(np.float32)
    F = faces.shape[1]
    rgb, alpha = [(True, True), (True, False), (False, True)][int(rng.integers(0, 3))]
    eps = float(rng.choice([1e-4, 1e-3, 1e-2, 0.1]))
    bright = rng.uniform() < 0.3
    textures = ((0.9 + 0.1 * rng.uniform(size=(B, F, 2, 2, 2, 3))) if bright else rng.uniform(0, 1, (B, F, 2, 2, 2, 3))).astype(np.float32)
    bg = (0.95, 0.95, 0.95) if bright else (0.1, 0.2, 0.3)
    fw = abi.forward_fused(faces, textures if rgb else None, S, 0.1, 100.0, eps, bg, 0, rgb, alpha, False)
    g_rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32) if rgb else None
    g_alpha = rng.normal(size=(B, S, S)).astype(np.float32) if alpha else None
    out = {k: abi.host(abi.backward_fused(fw, g_rgb, g_alpha, None, k6_flags=k)[0]) for k in (0, 128, 2, 130)}
    scale = max(float(np.abs(out[2]).max()), 1e-30)
    e_def = float(np.abs(out[0] - out[128]).max()) / scale
    e_def_exact = float(np.abs(out[0] - out[2]).max()) / scale
    e_ex = float(np.abs(out[2] - out[130]).max()) / scale
    bits = int((out[2] != out[130]).sum())
    worst['default'] = max(worst['default'], e_def, e_def_exact)
    worst['exact'] = max(worst['exact'], e_ex)
    worst['exact_bits'] = max(worst['exact_bits'], bits)
    if e_def > 1e-4 or e_def_exact > 1e-4 or bits > 2 or not np.isfinite(out[0]).all():
        bad.append(dict(it=it, S=S, B=B, F=F, kind=kind, rgb=rgb, alpha=alpha, eps=eps, bright=bright, row_vs_fast=e_def,
                        row_vs_exact=e_def_exact, exact_row_vs_fast=e_ex, exact_bits=bits))
        print(json.dumps(bad[-1]), flush=True)
print(json.dumps({'scenes': it + 1, 'worst': worst, 'flagged': len(bad)}), flush=True)
