"""Development: what other ways to group a window's records in k_bpm_row's phase B would cost, from the records of real windows
(a -DNR_ROW_STATS build dumps segments / direction / has-an-out-sweep of every record of the first 65 536 windows).
Policies, all in lane-steps (a step = 64 lanes for one 16-pixel segment each) per useful lane-step:
  now       groups of 4 records in falling order of segments, a group walks its longest sweep rounded up to 2 segments
  b16       batches of 16 records = 4 quads, quads pure in direction (sorted by direction, then segments), batch walks its longest
  b8        batches of 8 = 2 quads, the same
  q64       one quad x 64 pixels per step (4 aligned segments), quads pure in direction
    SHAPES="64x256 64x512" python scripts/row_groupings.py"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import neural_renderer_amd as nr
from neural_renderer_amd import _lib
from k6_numerics import use_library

dev = torch.device('cuda', 0)


def up(x, m):
    return (x + m - 1) // m * m


def simulate(win):
    """win: [n_windows, 64] uint32"""
    tot = dict(useful=0, now=0, b16=0, b8=0, q64=0, b16_mixed=0)
    for w in win:
        v = w[(w & 1024) != 0]
        v = v[(v & 512) != 0]
        if v.size == 0:
            continue
        nseg = (v & 255).astype(np.int64)
        dpos = (v & 256) != 0
        tot['useful'] += int(nseg.sum())
        # now: sorted by nseg desc, groups of 4
        s = np.sort(nseg)[::-1]
        for g in range(0, s.size, 4):
            tot['now'] += up(int(s[g]), 2) * 4
        # batches of 16 with any direction per block row (what the present layout would give with 16 records per step)
        for g in range(0, s.size, 16):
            tot['b16_mixed'] += up(int(s[g]), 2) * 16
        # direction-pure quads
        quads = []
        for d in (False, True):
            sd = np.sort(nseg[dpos == d])[::-1]
            for g in range(0, sd.size, 4):
                quads.append(int(sd[g]))
        quads.sort(reverse=True)
        for g in range(0, len(quads), 4):
            tot['b16'] += up(quads[g], 2) * 16
        for g in range(0, len(quads), 2):
            tot['b8'] += up(quads[g], 2) * 8
        for q in quads:
            tot['q64'] += up(q, 4) * 4
    return tot


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
    raw = ctypes.CDLL(os.environ['NR_HIP_LIB'])
    buf = np.zeros((1 << 16, 64), dtype=np.uint32)
    n = ctypes.c_uint(0)
    raw.nr_dev_row_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    _lib.check(lib.nr_backward_pixel_map(faces.data_ptr(), r.face_index_map.data_ptr(), r.rgb_map.data_ptr(), r.alpha_map.data_ptr(),
                                         g_rgb.data_ptr(), g_alpha.data_ptr(), gf.data_ptr(), B, F, S, 1e-3, 1, 1, 0,
                                         r.visible.data_ptr(), ws.data_ptr(), wsb, st), 'k6')
    torch.cuda.synchronize()
    raw.nr_dev_row_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    nwin = min(int(n.value), 1 << 16)
    tot = simulate(buf[:nwin])
    print(json.dumps({'B': B, 'S': S, 'windows': nwin, 'useful_segment_steps': tot['useful'],
                      **{k: round(tot['useful'] / max(tot[k], 1), 3) for k in ('now', 'b16_mixed', 'b16', 'b8', 'q64')}}), flush=True)
use_library('')
