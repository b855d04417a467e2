"""BASELINE.json configs[0] (teapot, 1 view, 64x64 silhouette, CPU only) and the cross-check of the two oracles.

oracle/numpy_naive.py is a second restatement of rasterize.py (NumPy per-pixel loop, SURVEY 8d "CPU baseline (i)") that
shares no code or loop structure with oracle/nr_oracle.c.  Integer and float maps must agree bit for bit; gradients, whose
terms are bit-identical but summed in a different order, to double round-off.
"""
import time

import numpy as np
import pytest

import helpers as H
from oracle import numpy_naive as N
from oracle import oracle as O

NEAR, FAR = 0.1, 100


def _oracle_forward(faces, s, eps, textures=None, **kw):
    fn = O.Rasterize(s, NEAR, FAR, eps, (0, 0, 0), return_rgb=textures is not None, return_alpha=True,
                     return_depth=True, **kw)
    fn(faces, textures)
    return fn


@pytest.mark.parametrize('s', [64, 256])
def test_config1_teapot_silhouette(s):
    """configs[0]: forward + backward of the silhouette at 64x64, one view, NumPy per-pixel loop vs the C oracle
    (and once more at the 256x256 of the reference's own fixtures)."""
    eps = 1e-4
    faces, _ = H.teapot_views(1, image_size=s)
    t0 = time.time()
    fi, weight, depth, inv_map = N.forward_face_index_map(faces, s, NEAR, FAR, return_face_inv=True)
    alpha = N.forward_alpha_map(fi)
    t_fwd = time.time() - t0
    fn = _oracle_forward(faces, s, eps)
    assert np.array_equal(fi, fn.face_index_map)
    assert 0.08 < (fi >= 0).mean() < 0.2                       # the teapot covers ~12 % of the image
    assert np.array_equal(weight.view(np.int32), fn.weight_map.view(np.int32))
    assert np.array_equal(depth.view(np.int32), fn.depth_map.view(np.int32))
    assert np.array_equal(inv_map.view(np.int32), fn.face_inv_map.view(np.int32))
    assert np.array_equal(alpha, fn.alpha_map)

    g_alpha = np.random.default_rng(0).standard_normal(alpha.shape).astype(np.float32)
    t0 = time.time()
    grad = N.backward_pixel_map(faces, fi, None, alpha, None, g_alpha, eps)
    t_bwd = time.time() - t0
    fn2 = O.Rasterize(s, NEAR, FAR, eps, (0, 0, 0), return_alpha=True)
    fn2(faces)
    ref = fn2.backward(None, g_alpha, None, accumulate_double=True)[0]
    assert np.abs(ref).max() > 0
    assert H.rel_err(grad, ref) < 1e-6
    print('naive NumPy, teapot 1 view %dx%d silhouette: forward %.2f s, backward %.2f s' % (s, s, t_fwd, t_bwd))


def test_naive_rgb_alpha_small_scene():
    """Random soup with textures, two images: K4 literal batch-0 z (Q1), per-batch background, K6 with rgb + alpha."""
    rng = np.random.default_rng(5)
    s, eps, ts = 24, 1e-3, 3
    faces = H.random_scene(rng, 2, 40)
    faces[0, 3, 1] = faces[0, 3, 0]                            # a degenerate face (two coincident vertices)
    faces[1, 7] = faces[1, 6]                                  # an exact duplicate: tie -> lowest index
    textures = rng.uniform(0, 1, (2, 40, ts, ts, ts, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, (2, 3)).astype(np.float32)
    fn = O.Rasterize(s, NEAR, FAR, eps, bg, return_rgb=True, return_alpha=True)
    fn(faces, textures)
    fi, weight, depth = N.forward_face_index_map(faces, s, NEAR, FAR)
    assert np.array_equal(fi, fn.face_index_map)
    assert not (fi[1] == 7).any()
    assert np.array_equal(weight.view(np.int32), fn.weight_map.view(np.int32))
    assert np.array_equal(depth.view(np.int32), fn.depth_map.view(np.int32))
    rgb = N.forward_texture_sampling(faces, textures, fi, weight, depth, eps, bg)
    assert np.array_equal(rgb.view(np.int32), fn.rgb_map.view(np.int32))
    alpha = N.forward_alpha_map(fi)

    g_rgb = rng.standard_normal(rgb.shape).astype(np.float32)
    g_alpha = rng.standard_normal(alpha.shape).astype(np.float32)
    grad = N.backward_pixel_map(faces, fi, rgb, alpha, g_rgb, g_alpha, eps)
    ref = fn.backward(g_rgb, g_alpha, None, accumulate_double=True)[0]
    assert np.abs(ref).max() > 0
    assert H.rel_err(grad, ref) < 1e-6
    # rgb only
    fn3 = O.Rasterize(s, NEAR, FAR, eps, bg, return_rgb=True)
    fn3(faces, textures)
    ref3 = fn3.backward(g_rgb, None, None, accumulate_double=True)[0]
    grad3 = N.backward_pixel_map(faces, fi, rgb, None, g_rgb, None, eps)
    assert H.rel_err(grad3, ref3) < 1e-6
