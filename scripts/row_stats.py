"""Development: work counters of k_bpm_row (a -DNR_ROW_STATS variant build, libnr_hip_stats.so) on teapot batches.
    SHAPES="64x256 64x512" python scripts/row_stats.py"""
import ctypes
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
for shape in os.environ.get('SHAPES', '64x256').split():
    B, S = (int(x) for x in shape.split('x'))
    use_library('')
    faces, textures = bench.build_scene(dev, B, 0, 64 if B <= 64 else B, S, 2)
    F = faces.shape[1]
    g_rgb, g_alpha, g_depth = bench.upstream_gradients(faces, textures, S, 1e-3, 1234)
    fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    fn.forward_gpu((faces, textures))
    r = fn._res
    lib = use_library('stats')
    st = torch.cuda.current_stream(dev).cuda_stream
    gf = torch.empty_like(faces)
    wsb = lib.nr_backward_workspace_bytes(B, F, S, 1, 1)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    out = (ctypes.c_ulonglong * 8)()
    raw = ctypes.CDLL(os.environ['NR_HIP_LIB'])
    raw.nr_dev_row_stats(out)
    _lib.check(lib.nr_backward_pixel_map(faces.data_ptr(), r.face_index_map.data_ptr(), r.rgb_map.data_ptr(), r.alpha_map.data_ptr(),
                                         g_rgb.data_ptr(), g_alpha.data_ptr(), gf.data_ptr(), B, F, S, 1e-3, 1, 1, 65536,
                                         r.visible.data_ptr(), ws.data_ptr(), wsb, st), 'k6')
    torch.cuda.synchronize()
    raw.nr_dev_row_stats(out)
    w, rec, n_out, segs, px, groups, steps = [int(out[i]) for i in range(7)]
    print(json.dumps({'B': B, 'S': S, 'windows': w, 'records': rec, 'out_records': n_out, 'segments': segs, 'out_pixels': px,
                      'groups': groups, 'steps': steps, 'records_per_window': round(rec / max(w, 1), 1),
                      'out_per_group': round(n_out / max(groups, 1), 2), 'steps_per_group': round(steps / max(groups, 1), 2),
                      'pixels_per_record': round(px / max(n_out, 1), 1),
                      'lane_efficiency_segments': round(px / max(16.0 * segs, 1), 3),
                      'lane_efficiency_steps': round(px / max(64.0 * steps, 1), 3)}), flush=True)
use_library('')
