"""Development: time the K6 stage call (nr_backward_pixel_map) and the fused backward for several libraries (variant builds,
NR_K6_* knobs of csrc/nr_k6_tune.h) on teapot batches of several sizes, one process.
    VARIANTS="t256b40 t128b20" SHAPES="8x256 16x256 64x256 64x512" python scripts/k6_variants.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import neural_renderer_amd as nr
from neural_renderer_amd import _lib
from k6_numerics import use_library

dev = torch.device('cuda', 0)
iters = int(os.environ.get('ITERS', 30))
variants = [''] + os.environ.get('VARIANTS', '').split()
for shape in os.environ.get('SHAPES', '8x256 16x256 32x256 64x256').split():
    B, S = (int(x) for x in shape.split('x'))
    use_library('')
    if os.environ.get('K6V_MESH', '').startswith('ico'):  # K6V_MESH=ico3: spheres of 20 * 4^3 faces (fill_back: twice that), random rotations
        import numpy as np
        v0, f0 = bench.icosphere(int(os.environ['K6V_MESH'][3:]))
        rng = np.random.default_rng(5)
        verts = np.stack([(0.75 * v0) @ np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32) for _ in range(B)]).astype(np.float32)
        fi_ = torch.from_numpy(f0).to(dev)[None].repeat(B, 1, 1)
        fi_ = torch.cat((fi_, torch.flip(fi_, dims=[2])), dim=1)
        eye = torch.tensor([[0.3, 0.4, -2.6]], dtype=torch.float32, device=dev).repeat(B, 1)
        faces = nr.vertices_to_faces(nr.perspective(nr.look_at(torch.from_numpy(verts).to(dev), eye), 30.), fi_).contiguous()
        textures = torch.from_numpy(rng.uniform(0, 1, (B, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)).to(dev)
    else:
        faces, textures = bench.build_scene(dev, B, 0, 64 if B <= 64 else B, S, 2)
    F, ts = faces.shape[1], 2
    g_rgb, g_alpha, g_depth = bench.upstream_gradients(faces, textures, S, 1e-3, 1234)
    fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    fn.forward_gpu((faces, textures))
    r = fn._res
    row = {'B': B, 'S': S}
    for tag in [variants[0]] + variants:  # (the first pass is thrown away: whatever runs first is ~3 % slow)
        lib = use_library(tag)
        st = torch.cuda.current_stream(dev).cuda_stream
        gf = torch.empty_like(faces)
        gt = torch.empty_like(textures)
        wsb = lib.nr_backward_workspace_bytes(B, F, S, 1, 1)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        m_rgb, m_alpha = (int(c) for c in os.environ.get('K6V_MODE', '11'))  # which gradients the K6 stage call gets: rgb, alpha
        for kfl in [int(x) for x in os.environ.get('K6V_FLAGS', '0').split()]:  # e.g. "128 65536": NR_FLAG_K6_LEGACY, NR_FLAG_K6_PX
            calls = {
                'k6': lambda: lib.nr_backward_pixel_map(faces.data_ptr(), r.face_index_map.data_ptr(), r.rgb_map.data_ptr(),
                                                        r.alpha_map.data_ptr(), g_rgb.data_ptr(), g_alpha.data_ptr(), gf.data_ptr(), B, F,
                                                        S, 1e-3, m_rgb, m_alpha, kfl, r.visible.data_ptr(), ws.data_ptr(), wsb, st),
                'bwd': lambda: lib.nr_backward_rasterize(faces.data_ptr(), None, r.face_index_map.data_ptr(), r.weight_map.data_ptr(),
                                                         r.depth_map.data_ptr(), r.rgb_map.data_ptr(), r.alpha_map.data_ptr(),
                                                         g_rgb.data_ptr(), g_alpha.data_ptr(), g_depth.data_ptr(), gf.data_ptr(),
                                                         gt.data_ptr(), B, F, S, ts, 1e-3, kfl, r.visible.data_ptr(), ws.data_ptr(), wsb, st)}
            for name, call in calls.items():
                for _ in range(3):
                    _lib.check(call(), name)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    call()
                e1.record()
                torch.cuda.synchronize()
                row['%s_%s%s' % (tag or 'product', name, '_f%d' % kfl if kfl else '')] = round(e0.elapsed_time(e1) * 1e3 / iters, 1)
    print(json.dumps(row), flush=True)
use_library('')
