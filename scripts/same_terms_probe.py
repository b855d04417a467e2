"""Development: how far two calls of the default K6 mode differ (float run sums regrouped by the order of atomics), staged / fused /
scan / serial-order, BASELINE config 2 at full size; the bound SAME_TERMS of tests/test_hip_parity.py rests on this.
    python scripts/same_terms_probe.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import abi, helpers as H
faces, _ = H.teapot_views(16, 256)
rng = np.random.default_rng(22)
textures = rng.uniform(0, 1, (16, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
fw = abi.forward(faces, textures, 256, 0.1, 100.0, 1e-3, (0.2, 0.4, 0.6), 0, True, True, True)
rng = np.random.default_rng(23)
g_rgb = rng.normal(size=(16, 256, 256, 3)).astype(np.float32); g_alpha = rng.normal(size=(16, 256, 256)).astype(np.float32); g_depth = rng.normal(size=(16, 256, 256)).astype(np.float32)
for flags in (0, 128, 2):  # the default mode (k_bpm_row), k_bpm_fast (NR_FLAG_K6_LEGACY), the exact mode
    base = abi.host(abi.backward(fw, g_rgb, g_alpha, g_depth, k6_flags=flags)[0])
    worst = {}
    for it in range(40):
        for name, kw in (('same', {}), ('novis', {'use_visible': False}), ('scan', {'k6_flags': flags | 8})):
            kw = dict({'k6_flags': flags}, **kw)
            g = abi.host(abi.backward(fw, g_rgb, g_alpha, g_depth, **kw)[0])
            assert not np.isnan(g).any()
            worst[name] = max(worst.get(name, 0.0), H.rel_err(g, base))
        gf = abi.host(abi.backward_fused(fw, g_rgb, g_alpha, g_depth, k6_flags=flags)[0])
        worst['fused'] = max(worst.get('fused', 0.0), H.rel_err(gf, base))
        gs = abi.host(abi.backward_fused(fw, g_rgb, g_alpha, g_depth, k6_flags=flags | 64)[0])
        worst['fused_serial'] = max(worst.get('fused_serial', 0.0), H.rel_err(gs, base))
    print('flags', flags, {k: float('%.3g' % v) for k, v in worst.items()}, flush=True)
