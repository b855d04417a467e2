"""Development: the exact mode (NR_FLAG_EXACT_GRADIENT) on K6's two band kernels, k_bpm_row (flags 2) and k_bpm_fast (flags 130),
on teapot views at power-of-two and other rasters: the share of grad_faces entries whose BITS differ and the largest
difference in the parity metric.  Both kernels form the reference's float terms and add them in double, so what may differ is
the order of the double additions (a float result moves when a sum sits within 1e-16 of a rounding point: practically never);
a term rounded differently by one kernel shows up as a share of differing entries orders of magnitude above that.
    SHAPES="8x250 8x256 4x640" python scripts/exact_kernels_probe.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import abi
import helpers as H

for shape in os.environ.get('SHAPES', '8x250 8x256 4x640 8x300').split():
    B, S = (int(x) for x in shape.split('x'))
    faces, _ = H.teapot_views(B, S)
    rng = np.random.default_rng(S)
    textures = rng.uniform(0, 1, (B, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    g_rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(B, S, S)).astype(np.float32)
    fw = abi.forward_fused(faces, textures, S, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, True, True, False)
    out = {}
    for flags in (2, 130):
        out[flags] = abi.host(abi.backward_fused(fw, g_rgb, g_alpha, None, k6_flags=flags)[0])
    a, b = out[2], out[130]
    print(json.dumps(dict(B=B, S=S, entries=int(a.size), nonzero=int((a != 0).sum()), bits_differ=int((a != b).sum()),
                          rel=H.rel_err(a, b))), flush=True)
