#!/usr/bin/env python3
"""The error bar on "bit-exact vs the reference" (DESIGN.md 3).

The oracle and the HIP kernels share the UN-fused reading of the reference's CUDA text; nvcc's default (-fmad=true) may fuse
the `a*b+c` of rasterize.py:258 (pixel coordinates), :261-269 (inverse barycentric matrix) and :317-319 (weights) into
fused multiply-adds.  The reference cannot be executed here, so the oracle is run twice on the same inputs -- un-fused (the
parity convention) and with those expressions contracted (oracle.set_contraction(True)) -- and the forward maps are
compared: pixels whose face_index_map differs, max |d depth|, max |d weight| (and the same relative to ulps).

    python scripts/contraction_study.py [H] [C4] [C5]      # CPU only; C5 is 6.9e11 face tests per run (minutes)

One JSON line per configuration on stdout (kept as profiles/r03_contraction_study.jsonl).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import oracle as O  # noqa: E402
import helpers as H  # noqa: E402


def forward_maps(faces, S, contract):
    O.set_contraction(contract)
    try:
        fn = O.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), False, True, True)
        fn.blocked = True
        fn(faces)
    finally:
        O.set_contraction(False)
    return fn.face_index_map, fn.depth_map, fn.weight_map


def compare(name, faces, S):
    t0 = time.time()
    fi0, d0, w0 = forward_maps(faces, S, False)
    fi1, d1, w1 = forward_maps(faces, S, True)
    same = fi0 == fi1
    cov = same & (fi0 >= 0)
    dd = np.abs(d1[cov].astype(np.float64) - d0[cov])
    ulp_d = np.spacing(np.abs(d0[cov]))
    dw = np.abs(w1[cov].astype(np.float64) - w0[cov])
    changed = ~same
    rec = {
        'config': name, 'B': int(faces.shape[0]), 'F': int(faces.shape[1]), 'S': S, 'pixels': int(fi0.size),
        'covered_pixels': int((fi0 >= 0).sum()),
        'face_index_differs': int(changed.sum()),
        'of_which_coverage_changes': int((changed & ((fi0 < 0) | (fi1 < 0))).sum()),
        'max_abs_d_depth': float(dd.max()) if dd.size else 0.0,
        'max_d_depth_ulp': float((dd / ulp_d).max()) if dd.size else 0.0,
        'pixels_with_depth_changed': int((dd > 0).sum()),
        'max_abs_d_weight': float(dw.max()) if dw.size else 0.0,
        'pixels_with_weight_changed': int((dw.max(axis=-1) > 0).sum()) if dw.size else 0,
        'd_weight_percentiles_50_99_99.9': [float(x) for x in np.percentile(dw.max(axis=-1), [50, 99, 99.9])] if dw.size else None,
        'frac_pixels_d_weight_above_1e-4': float((dw.max(axis=-1) > 1e-4).mean()) if dw.size else 0.0,
        'd_depth_rel_percentiles_50_99_99.9': [float(x) for x in np.percentile(dd / np.abs(d0[cov]), [50, 99, 99.9])] if dd.size else None,
        'seconds': round(time.time() - t0, 1), 'threads': O.get_threads(),
    }
    print(json.dumps(rec), flush=True)
    return rec


def main():
    which = sys.argv[1:] or ['H']
    if 'H' in which:
        faces, _ = H.teapot_views(64, 256)
        compare('H: teapot, 64 views, 256x256', faces, 256)
    if 'C4' in which:
        from test_full_size_gpu import config4_meshes
        compare('C4: 64 random meshes x 10240 faces, 256x256', config4_meshes(64), 256)
    if 'C5' in which:
        from test_hip_parity import icosphere, project_mesh
        rng = np.random.default_rng(55)
        v0, f0 = icosphere(7)
        v = v0 * (0.6 + 0.02 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        faces = project_mesh(v.astype(np.float32), f0, [0.0, 0.0, -2.4])[None]
        compare('C5: one 655360-face mesh, 1024x1024', faces, 1024)


if __name__ == '__main__':
    main()
