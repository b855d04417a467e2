"""The C oracle's OpenMP build and its cache-blocked K2 loop order change WHICH independent unit is evaluated when, never
a result: every map and gradient must be bit-identical for 1 thread, all threads, and the blocked / literal K2 order.
CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
import helpers as H


def _run(faces, textures, S, threads, blocked, double=False):
    O.set_threads(threads)
    try:
        fn = O.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, True)
        fn.blocked = blocked
        fn(faces, textures)
        rng = np.random.default_rng(5)
        g_rgb = rng.normal(size=fn.rgb_map.shape).astype(np.float32)
        g_alpha = rng.normal(size=fn.alpha_map.shape).astype(np.float32)
        g_depth = rng.normal(size=fn.depth_map.shape).astype(np.float32)
        gf, gt = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=double)
        return dict(fi=fn.face_index_map, w=fn.weight_map, d=fn.depth_map, inv=fn.face_inv_map, rgb=fn.rgb_map,
                    a=fn.alpha_map, si=fn.sampling_index_map, sw=fn.sampling_weight_map, gf=gf.copy(), gt=gt.copy(),
                    visits=fn.visits)
    finally:
        O.set_threads(0)


def _scenes():
    rng = np.random.default_rng(3)
    faces, _ = H.teapot_views(2, 70)  # 70: a row is one full block of 64 pixels plus a ragged one
    yield 'teapot', faces, rng.uniform(0, 1, (2, faces.shape[1], 3, 3, 3, 3)).astype(np.float32), 70
    soup = H.random_scene(rng, 3, 200, spread=0.7, size=0.3)
    soup[:, 0] = 0.0
    soup[:, 1] = soup[:, 1, :1]
    soup[:, 7] = soup[:, 8]          # duplicate faces: the tie rule (lower index wins) must survive the re-ordering
    soup[0, 3, 1, 0] = np.nan
    soup[1, 5, 0, 2] = 0.0
    soup[2, 9, :, :2] *= 1e20
    yield 'soup', soup, rng.uniform(0, 1, (3, 200, 2, 2, 2, 3)).astype(np.float32), 33


@pytest.mark.parametrize('double', [False, True], ids=['float_sums', 'double_sums'])
def test_threads_and_blocked_order_are_bit_identical(double):
    assert O.get_threads() >= 1
    for name, faces, textures, S in _scenes():
        ref = _run(faces, textures, S, 1, False, double)
        for threads, blocked in ((0, False), (1, True), (0, True), (3, True)):
            got = _run(faces, textures, S, threads, blocked, double)
            for k, v in ref.items():
                if k == 'visits':
                    assert got[k] == v, (name, k)
                else:
                    assert np.array_equal(got[k], v, equal_nan=True), (name, k, threads, blocked)


def test_blocked_order_is_selected_by_size():
    assert 64 * 256 * 256 * 4928 >= O.BLOCKED_K2_THRESHOLD      # headline batch
    assert 16 * 256 * 256 * 4928 < O.BLOCKED_K2_THRESHOLD       # config 2 keeps the literal order
