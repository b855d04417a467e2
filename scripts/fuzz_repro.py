"""Development: one scene of tests/test_fuzz_gpu.py::test_fuzz_unusual_parameters again -- `SEED=118 IT=50 python scripts/fuzz_repro.py` --
with the errors of every K6 kernel / mode, with and without the depth gradient (K8's float sums), and the worst entries."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O
import abi
import helpers as H

seed, target = int(os.environ.get('SEED', 118)), int(os.environ.get('IT', 50))
rng = np.random.default_rng(seed)
for it in range(target + 1):
    B = int(rng.integers(1, 4))
    F = int(rng.integers(1, 50))
    S = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 47, 64]))
    ts = int(rng.choice([2, 2, 3, 4]))
    eps = float(rng.choice([0.0, 1e-10, 1e-4, 1e-3, 0.1, 1.0]))
    near = float(rng.choice([1e-6, 0.1, 0.5, 1.7]))
    far = float(rng.choice([2.0, 10.1, 100, 1e10]))
    faces = H.random_scene(rng, B, F, spread=float(rng.choice([0.3, 0.8, 1.5])), size=float(rng.choice([0.05, 0.3, 1.2])),
                           zmin=float(rng.choice([-1.0, 0.05, 1.0])), zmax=3.0)
    if rng.uniform() < 0.5:
        q = (np.round((faces[..., :2] * S + S - 1) / 2) * 2 + 1 - S) / S
        m = rng.uniform(size=faces[..., :2].shape) < 0.5
        faces[..., :2] = np.where(m, q, faces[..., :2]).astype(np.float32)
    if rng.uniform() < 0.3 and F > 1:
        faces[:, F // 2:] = faces[:, : F - F // 2][:, :, ::-1]
    textures = rng.uniform(0, 1, (B, F, ts, ts, ts, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, (B, 3)).astype(np.float32) if rng.uniform() < 0.5 else (0.2, 0.4, 0.6)
    flags = int(rng.integers(0, 2))
    if it < target:
        # (the generator's later draws must be consumed as the test consumes them)
        fn = O.Rasterize(S, near, far, eps, bg, True, True, True, bool(flags))
        fn(faces, textures)
        g = [rng.normal(size=x.shape).astype(np.float32) for x in (fn.rgb_map, fn.alpha_map, fn.depth_map)]
        if rng.uniform() < 0.3:
            rng.uniform(size=g[0].shape)
        continue
    print(dict(B=B, F=F, S=S, ts=ts, eps=eps, near=near, far=far, flags=flags))
    fn = O.Rasterize(S, near, far, eps, bg, True, True, True, bool(flags))
    fn(faces, textures)
    fw = abi.forward(faces, textures, S, near, far, eps, bg, flags, True, True, True)
    g = [rng.normal(size=x.shape).astype(np.float32) for x in (fn.rgb_map, fn.alpha_map, fn.depth_map)]
    if rng.uniform() < 0.3:
        g[0][rng.uniform(size=g[0].shape) < 0.5] = 0
    for with_depth in (True, False):
        gd = g[2] if with_depth else np.zeros_like(g[2])
        ref_gf, _ = fn.backward(g[0], g[1], gd, accumulate_double=True)
        print('depth gradient', 'on' if with_depth else 'zero', ' max |ref|', float(np.nanmax(np.abs(ref_gf))))
        for k6 in (0, 128, 2, 130, 4, 8):
            gf = abi.host(abi.backward(fw, g[0], g[1], gd, k6_flags=k6)[0])
            ok = np.isfinite(ref_gf) & np.isfinite(gf)
            e = H.rel_err(gf[ok], ref_gf[ok])
            d = np.abs(np.where(ok, gf - ref_gf, 0))
            w = np.unravel_index(np.argmax(d), d.shape)
            print('  k6_flags %3d  err %.3e   worst entry %s: got %.9g ref %.9g' % (k6, e, w, gf[w], ref_gf[w]))
    np.save('/tmp/fuzz_faces.npy', faces)
    # the entries on which the default mode's two band kernels disagree, and what moves them
    if os.environ.get('DIG'):
        gd = np.zeros_like(g[2])
        ref_gf, _ = fn.backward(g[0], g[1], gd, accumulate_double=True)
        a = abi.host(abi.backward(fw, g[0], g[1], gd, k6_flags=0)[0])
        b = abi.host(abi.backward(fw, g[0], g[1], gd, k6_flags=128)[0])
        bad = np.argwhere(np.abs(a - b) > 1e-4 * np.abs(ref_gf).max())
        print('entries where k_bpm_row and k_bpm_fast differ:', len(bad))
        for w in bad[:12]:
            w = tuple(w)
            print('   ', w, 'row %.7g fast %.7g ref %.7g' % (a[w], b[w], ref_gf[w]), ' face (pixels):', ((faces[w[0], w[1], :, :2] * S + S - 1) / 2).round(3).tolist())
        for name, kw in (('rgb gradient only', dict(ga=None)), ('alpha gradient only', dict(gr=None))):
            gr = g[0] if kw.get('gr', 1) is not None else None
            ga = g[1] if kw.get('ga', 1) is not None else None
            fn2 = O.Rasterize(S, near, far, eps, bg, gr is not None, ga is not None, False, bool(flags))
            fn2(faces, textures) if gr is not None else fn2(faces)
            fw2 = abi.forward(faces, textures if gr is not None else None, S, near, far, eps, bg, flags, gr is not None, ga is not None, False)
            r2 = fn2.backward(gr, ga, None, accumulate_double=True)[0]
            for k6 in (0, 128):
                x = abi.host(abi.backward(fw2, gr, ga, None, k6_flags=k6)[0])
                print('   ', name, 'k6_flags', k6, 'err %.3e' % H.rel_err(x, r2))
    if os.environ.get('DIG'):
        # which pixel's term is it?  the alpha gradient of one pixel at a time
        fn2 = O.Rasterize(S, near, far, eps, bg, False, True, False, bool(flags))
        fn2(faces)
        fw2 = abi.forward(faces, None, S, near, far, eps, bg, flags, False, True, False)
        print('alpha map of image 0:'); print(fn2.alpha_map[0].astype(int)); print('face index map of image 0:'); print(fn2.face_index_map[0])
        print('upstream alpha gradient of image 0:'); print(np.round(g[1][0], 3))
        for y in range(S):
            for x in range(S):
                ga = np.zeros_like(g[1]); ga[0, y, x] = g[1][0, y, x]
                r2 = fn2.backward(None, ga, None, accumulate_double=True)[0]
                a = abi.host(abi.backward(fw2, None, ga, None, k6_flags=0)[0])
                b = abi.host(abi.backward(fw2, None, ga, None, k6_flags=128)[0])
                if np.abs(a - b).max() > 1e-5 * max(np.abs(r2).max(), 1e-30):
                    w = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
                    print('pixel (y %d, x %d): entry %s row %.7g fast %.7g ref %.7g' % (y, x, tuple(int(v) for v in w), a[w], b[w], r2[w]))
