"""Renderer.render forward + backward with and without per-face light colours (Renderer.face_light):

    python scripts/face_light_timing.py            # one JSON line per scene

Scenes: the headline teapot (64 views, 256 x 256 after 2x anti-aliasing off, ts 2) and config 4's shape (64 distinct
5 120-face meshes, fill_back -> 10 240, ts 4, 256 x 256), vertices AND textures receiving gradients.  Reports ms per
render + backward, the largest relative difference of the images and of the two gradients between the two modes, and the
peak torch memory of a step.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import bench
import neural_renderer_amd as nr


# development switches: FL_MODES=0 | 1 (one mode only, for a kernel trace), FL_ONLY=teapot
MODES = [bool(int(c)) for c in os.environ.get('FL_MODES', '01')]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def scene(name, vertices, faces, textures, image_size, anti_aliasing):
    dev = vertices.device
    B = vertices.shape[0]
    out = {'scene': name, 'B': B, 'faces': int(faces.shape[1]), 'ts': int(textures.shape[2]), 'image_size': image_size,
           'anti_aliasing': anti_aliasing}
    keep = {}
    for flag in MODES:
        r = nr.Renderer()
        r.image_size = image_size
        r.anti_aliasing = anti_aliasing
        r.face_light = flag
        r.eye = torch.tensor([nr.get_points_from_angles(2.732, 30., 360.0 * i / B) for i in range(B)], dtype=torch.float32,
                             device=dev)
        v = vertices.clone().requires_grad_(True)
        t = textures.clone().requires_grad_(True)

        def step():
            v.grad = None
            t.grad = None
            img = r.render(v, faces, t)
            img.square().sum().backward()
            return img

        def fwd():
            with torch.no_grad():
                return r.render(v, faces, t)

        img = step()
        keep[flag] = (img.detach().cpu(), v.grad.cpu(), t.grad.cpu())  # (off the device: not part of the peak below)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        step()
        torch.cuda.synchronize()
        key = 'face_light' if flag else 'lit_textures'
        out[key] = {'fwd_bwd_ms': round(timeit(step), 3), 'fwd_ms': round(timeit(fwd), 3),
                    'peak_MB': round(torch.cuda.max_memory_allocated() / 1e6, 1)}
    if len(MODES) < 2:
        print(json.dumps(out), flush=True)
        return
    out['speedup'] = round(out['lit_textures']['fwd_bwd_ms'] / out['face_light']['fwd_bwd_ms'], 3)
    out['max_rel_diff'] = {'images': rel(keep[True][0], keep[False][0]), 'grad_vertices': rel(keep[True][1], keep[False][1]),
                           'grad_textures': rel(keep[True][2], keep[False][2])}
    print(json.dumps(out), flush=True)


def main():
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(3)
    v, f = bench.load_teapot()
    B = 64
    scene('teapot 64 views ts2', torch.from_numpy(v).to(dev)[None].repeat(B, 1, 1), torch.from_numpy(f).to(dev)[None].repeat(B, 1, 1),
          torch.rand((B, f.shape[0], 2, 2, 2, 3), device=dev), 256, False)
    if os.environ.get('FL_ONLY') == 'teapot':
        return
    from test_hip_parity import icosphere
    v0, f0 = icosphere(4)
    vs = []
    for _ in range(B):
        vv = v0 * (0.55 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        vs.append((vv @ q).astype(np.float32))
    vertices = torch.from_numpy(np.stack(vs)).to(dev)
    faces = torch.from_numpy(f0.astype(np.int32)).to(dev)[None].repeat(B, 1, 1)
    for ts in (4, 8):
        scene('C4 shape: 64 meshes x %d faces ts%d' % (f0.shape[0], ts), vertices, faces,
              torch.rand((B, f0.shape[0], ts, ts, ts, 3), device=dev), 256, False)
    scene('C4 shape, anti-aliasing on (512 raster)', vertices, faces, torch.rand((B, f0.shape[0], 4, 4, 4, 3), device=dev), 256, True)


if __name__ == '__main__':
    main()
