"""Development: re-run one scene of tests/test_fuzz_gpu.py::test_fuzz_unusual_parameters and print where the default K6
kernel deviates from the oracle.   python scripts/debug_fuzz_case.py <seed> <iteration>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import abi, helpers as H
from oracle import oracle as O
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(60):
    B = int(rng.integers(1, 4)); F = int(rng.integers(1, 50)); S = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 47, 64]))
    ts = int(rng.choice([2, 2, 3, 4])); eps = float(rng.choice([0.0, 1e-10, 1e-4, 1e-3, 0.1, 1.0]))
    near = float(rng.choice([1e-6, 0.1, 0.5, 1.7])); far = float(rng.choice([2.0, 10.1, 100, 1e10]))
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
    if it == target:
        fn = O.Rasterize(S, near, far, eps, bg, True, True, True, bool(flags)); fn(faces, textures)
        fw = abi.forward(faces, textures, S, near, far, eps, bg, flags, True, True, True)
    g = [rng.normal(size=x).astype(np.float32) for x in ((B, S, S, 3), (B, S, S), (B, S, S))]
    if rng.uniform() < 0.3:
        g[0][rng.uniform(size=g[0].shape) < 0.5] = 0
    if it == target:
        print(dict(B=B, F=F, S=S, ts=ts, eps=eps, near=near, far=far, flags=flags))
        ref_gf, _ = fn.backward(*g, accumulate_double=True)
        print('visits', fn.visits, 'max|ref|', np.nanmax(np.abs(ref_gf)), 'rgb range', np.nanmin(fn.rgb_map), np.nanmax(fn.rgb_map))
        ref_f, _ = fn.backward(*g)  # the reference's own float sums
        okf = np.isfinite(ref_gf) & np.isfinite(ref_f)
        print('reference float sums vs the same terms summed in double:', H.rel_err(ref_f[okf], ref_gf[okf]))
        for k6 in (0, 128, 65536, 2):
            gf, _ = abi.backward(fw, *g, k6_flags=k6)
            gf = abi.host(gf)
            ok = np.isfinite(ref_gf) & np.isfinite(gf)
            err = np.abs(gf - ref_gf); err[~ok] = 0
            idx = np.unravel_index(np.argmax(err), err.shape)
            print('flags', k6, 'rel_err', H.rel_err(gf[ok], ref_gf[ok]), 'worst at', idx, 'got', gf[idx], 'ref', ref_gf[idx], 'float ref',
                  ref_f[idx], 'face', faces[idx[0], idx[1]].tolist())
        break
