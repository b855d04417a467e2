"""Development: time the fused forward (nr_forward_rasterize, kept workspace with epochs) for several libraries on teapot
batches of several sizes, one process.   VARIANTS="tiny" SHAPES="1x256 8x256" python scripts/fwd_variants.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from neural_renderer_amd import _lib
from k6_numerics import use_library

dev = torch.device('cuda', 0)
iters = int(os.environ.get('ITERS', 50))
fwd_flags = int(os.environ.get('FWD_FLAGS', 16 | 32))  # epochs + sparse weight map: what the operator passes
variants = [''] + os.environ.get('VARIANTS', '').split()
for shape in os.environ.get('SHAPES', '1x256 4x256 8x256 16x256').split():
    B, S = (int(x) for x in shape.split('x'))
    use_library('')
    if os.environ.get('SCENE') == 'C4':  # config 4: B distinct random meshes x 10 240 faces, texture_size 4
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from test_full_size_gpu import config4_meshes
        faces = torch.from_numpy(config4_meshes(B)).to(dev)
        textures = torch.rand((B, faces.shape[1], 4, 4, 4, 3), device=dev)
    else:
        faces, textures = bench.build_scene(dev, B, 0, 64 if B <= 64 else B, S, 2)
    F, ts = faces.shape[1], int(textures.shape[2])
    fi = torch.empty((B, S, S), dtype=torch.int32, device=dev)
    wm = torch.empty((B, S, S, 3), device=dev); dm = torch.empty((B, S, S), device=dev)
    rgb = torch.empty((B, S, S, 3), device=dev); am = torch.empty((B, S, S), device=dev)
    vis = torch.empty((B, F), dtype=torch.uint8, device=dev); bg = torch.zeros(3, device=dev)
    row = {'B': B, 'S': S}
    for tag in [variants[0]] + variants:  # (first pass thrown away)
        lib = use_library(tag)
        st = torch.cuda.current_stream(dev).cuda_stream
        wsb = lib.nr_forward_workspace_bytes(B, F, S)
        ws = torch.full((wsb,), 255, dtype=torch.uint8, device=dev)
        ep = [254]

        def call():
            e = ep[0]
            ep[0] -= 1
            return lib.nr_forward_rasterize(faces.data_ptr(), None, textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(),
                                            rgb.data_ptr(), am.data_ptr(), vis.data_ptr(), bg.data_ptr(), 0, B, F, S, ts, 0.1, 100.0,
                                            1e-3, fwd_flags | (e << 8), ws.data_ptr(), wsb, st)
        for _ in range(3):
            _lib.check(call(), 'fwd')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        row[tag or 'product'] = round(e0.elapsed_time(e1) * 1e3 / iters, 1)
        got = (fi, dm, rgb, am)
        ref = tuple(x.clone() for x in got) if tag == '' else ref
        assert all(torch.equal(x, y) for x, y in zip(got, ref)), 'variant %r differs from the product library' % tag
    print(json.dumps(row), flush=True)
use_library('')
