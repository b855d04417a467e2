"""Parity tests proper (`-m gpu`): the HIP kernels, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star): face_index_map bit-exact; rgb / depth / gradients
within 1e-4 relative.  The forward float maps use the oracle's operation order and are expected to be
bit-identical as well; the tests assert that and fall back to the stated tolerance only for the sums whose
order legitimately differs (K6 tree reduction, K7/K8 atomics)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
import abi
import helpers as H

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star tolerance for floating-point outputs
# K6 against the reference's per-pixel terms summed exactly (oracle, accumulate_double): the default kernel evaluates the
# terms in float through the hardware reciprocal, NR_FLAG_EXACT_GRADIENT with the reference's own arithmetic
K6_BOUND_DEFAULT = 1e-4  # the north star's tolerance; measured levels per test: profiles/r06_parity_summary.md (worst 4.8e-5)
K6_BOUND_EXACT = 2e-6
SAME_TERMS = 3e-5  # two evaluations of the same per-pixel terms in different summation orders (float run sums of the default K6
                   # kernel, regrouped by the order of its atomics: up to 1.2e-5 between two calls on config 2 where the line
                   # sums of a face cancel -- 200 pairs, scripts/same_terms_probe.py; the bound was 1e-5 until one run in ~10 of
                   # this file crossed it.  The exact mode's double sums came out bit-identical in all of them.)
EXACT = 2  # _lib.NR_FLAG_EXACT_GRADIENT
K6_GLOBAL = 4  # _lib.NR_FLAG_K6_GLOBAL
K6_SCAN = 8    # _lib.NR_FLAG_K6_SCAN
SERIAL = 64    # _lib.NR_FLAG_SERIAL_BACKWARD
K6_LEGACY = 128  # _lib.NR_FLAG_K6_LEGACY: the default mode on the piece-per-lane band kernel (k_bpm_fast) instead of k_bpm_row
K6_PX = 65536    # _lib.NR_FLAG_K6_PX: accepted and ignored since 0.6.0 (the default mode has one band kernel, k_bpm_row)


def report(test, **values):
    """Measured error levels go to gpurun_out/parity_errors.jsonl (when that directory exists) so that a GPU session
    leaves the numbers behind, not only pass / fail."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(d):
        import json
        with open(os.path.join(d, 'parity_errors.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test=os.environ.get('PYTEST_CURRENT_TEST', test), **values)) + '\n')


def oracle_forward(faces, textures, S, near, far, eps, background, return_rgb, return_alpha, return_depth,
                   fix_batch_z=False):
    fn = O.Rasterize(S, near, far, eps, background, return_rgb, return_alpha, return_depth, fix_batch_z)
    fn(faces, textures) if return_rgb else fn(faces)
    return fn


def check_forward(fw, fn, exact=True):
    fi = abi.host(fw['face_index_map'])
    assert int((fi != fn.face_index_map).sum()) == 0, 'face_index_map must be bit-exact'
    for name in ('weight_map', 'depth_map', 'face_inv_map', 'rgb_map', 'alpha_map', 'sampling_weight_map'):
        ref = getattr(fn, name, None)
        got = fw.get(name)
        if ref is None or got is None:
            continue
        got = abi.host(got)
        assert got.shape == ref.shape, name
        assert not np.isnan(got).any(), name + ' has unwritten / NaN elements'
        if exact:
            np.testing.assert_array_equal(got, ref, err_msg=name)
        else:
            assert H.rel_err(got, ref) <= RTOL, name
    if fw.get('sampling_index_map') is not None and fn.sampling_index_map is not None:
        np.testing.assert_array_equal(abi.host(fw['sampling_index_map']), fn.sampling_index_map)


# ---------------------------------------------------------------------------------------------------
def test_teapot_silhouette_depth_rgb_default_camera():
    """Reference fixture scene (tests/test_rasterize*.py): teapot, default eye, 256x256, no AA."""
    v, f = H.teapot()
    r = O.Renderer()
    faces = r.project(v[None], f[None])
    rng = np.random.default_rng(0)
    textures = rng.uniform(0, 1, (1, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, 256, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True, True)
    fw = abi.forward(faces, textures, 256, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, True, True, True,
                     want_face_inv=True, want_sampling=True)
    check_forward(fw, fn)
    # and the reference's own golden: silhouette == Blender render
    alpha = abi.host(fw['alpha_map'])[0][::-1]
    assert int((alpha != H.golden()['teapot_blender'].astype(np.float32)).sum()) == 0


@pytest.mark.parametrize('S', [64, 100, 256])
def test_teapot_views_forward(S):
    faces, _ = H.teapot_views(4, S)
    rng = np.random.default_rng(1)
    textures = rng.uniform(0, 1, (4, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, (4, 3)).astype(np.float32)  # per-batch background (rasterize.py:464-465)
    fn = oracle_forward(faces, textures, S, 0.1, 100, 1e-3, bg, True, True, True)
    fw = abi.forward(faces, textures, S, 0.1, 100.0, 1e-3, bg, 0, True, True, True, want_face_inv=True,
                     want_sampling=True)
    check_forward(fw, fn)


def test_texture_batch_z_quirk_both_ways():
    """literal (batch 0's z) and fixed (own batch) sampling both match the oracle."""
    faces, _ = H.teapot_views(3, 64)
    rng = np.random.default_rng(2)
    textures = rng.uniform(0, 1, (3, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
    for fix in (False, True):
        fn = oracle_forward(faces, textures, 64, 0.1, 100, 1e-3, (0, 0, 0), True, False, False, fix_batch_z=fix)
        fw = abi.forward(faces, textures, 64, 0.1, 100.0, 1e-3, (0, 0, 0), int(fix), True, False, False,
                         want_sampling=True)
        check_forward(fw, fn)


def edge_case_scene(rng, B=3, F=150):
    faces = H.random_scene(rng, B, F)
    faces[:, 0] = 0.0                                   # all-zero face (reference tests/utils.py empty slots)
    faces[:, 1] = faces[:, 1, :1]                       # three coincident vertices
    faces[:, 2, 2] = 0.5 * (faces[:, 2, 0] + faces[:, 2, 1])   # collinear (zero-area) face
    faces[:, 3, :, 2] = 0.05                            # in front of the near plane
    faces[:, 4, :, 2] = 150.0                           # beyond the far plane
    faces[:, 5, :, :2] += 5.0                           # completely off screen
    faces[:, 6, :, :2] *= 8.0                           # huge face covering the whole image
    faces[:, 7] = faces[:, 8]                           # exact duplicate: tie -> lower face index wins
    faces[:, 9, :, :2] = np.array([[-1, -1], [1, -1], [-1, 1]], np.float32) * (1 - 1.0 / 32)  # on pixel centres
    faces[:, 10, :, 2] = -1.0                           # behind the camera (negative depth)
    return faces


@pytest.mark.parametrize('S', [8, 33, 64, 96])
def test_edge_cases_forward(S):
    rng = np.random.default_rng(3)
    faces = edge_case_scene(rng)
    textures = rng.uniform(0, 1, (faces.shape[0], faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, S, 0.1, 100, 1e-4, (0, 0, 0), True, True, True)
    fw = abi.forward(faces, textures, S, 0.1, 100.0, 1e-4, (0, 0, 0), 0, True, True, True, want_face_inv=True)
    check_forward(fw, fn)


@pytest.mark.parametrize('near,far', [(0, 100), (0.0, 2.5), (-1, 100), (-1.5, -0.25), (-100, 1.5)])
def test_near_zero_and_negative_like_the_reference(near, far):
    """The reference accepts any `near` (rasterize.py:331 pastes it as a literal): near = 0 and negative near planes, with
    faces behind the camera (negative depths, drawn when near < 0), faces straddling the camera plane (1 / (w0/z0 + ..) of
    mixed signs: huge, infinite or NaN depths) and a z-fight between +0-ish depths.  The packed z-buffer orders depths
    through a monotone integer key, so every map must agree with the oracle bit for bit, and so must the gradients'
    NaN pattern / values."""
    rng = np.random.default_rng(31)
    faces = edge_case_scene(rng, B=2, F=120)
    faces[:, 20:40, :, 2] = -rng.uniform(0.3, 3.0, (2, 20, 3)).astype(np.float32)       # behind the camera
    faces[:, 40:50, 0, 2] *= -1.0                                                        # straddling the camera plane
    faces[:, 50:55, :, 2] = rng.uniform(1e-30, 1e-20, (2, 5, 3)).astype(np.float32)      # depths just above zero
    faces[:, 55:60, :, 2] = -rng.uniform(1e-30, 1e-20, (2, 5, 3)).astype(np.float32)     # ... and just below
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    for S in (32, 67):
        fn = oracle_forward(faces, textures, S, near, far, 1e-3, (0.3, 0.2, 0.1), True, True, True)
        fw = abi.forward(faces, textures, S, float(near), float(far), 1e-3, (0.3, 0.2, 0.1), 0, True, True, True,
                         want_face_inv=True)
        fi = abi.host(fw['face_index_map'])
        assert int((fi != fn.face_index_map).sum()) == 0
        for name in ('weight_map', 'depth_map', 'face_inv_map', 'rgb_map', 'alpha_map'):
            np.testing.assert_array_equal(abi.host(fw[name]), getattr(fn, name), err_msg=name)  # (NaN == NaN here)
    if near < 0 and far > 0:
        assert (fn.depth_map < 0).any() and (fn.depth_map > 0).any()  # both sides of the camera were drawn
    # gradients on the well-conditioned part of the scene (no straddling faces: their depths are +-inf / NaN)
    faces[:, 40:50, 0, 2] *= -1.0
    faces[:, 50:60, :, 2] = np.abs(faces[:, 50:60, :, 2]) + 1.0
    check_backward(faces, textures, 48, 1e-3, (True, True, True), seed=32, near=near, far=far)


def test_single_face_and_empty_image():
    one = np.array([[[[0.8, 0.8, 1.], [0.0, -0.5, 1.], [0.2, -0.4, 1.]]]], np.float32)
    for faces in (one, one[:, :, ::-1].copy()):  # front-facing / back-facing only (empty image)
        fn = oracle_forward(faces, None, 64, 0.1, 100, 1e-4, None, False, True, True)
        fw = abi.forward(faces, None, 64, return_alpha=True, return_depth=True, want_face_inv=True)
        check_forward(fw, fn)


def test_thin_face_with_far_offscreen_vertices():
    """A face whose height is below 2^-18 of its longest edge counts as a needle (candidates = a 4-pixel strip along the
    edge), but with vertices thousands of NDC units off-screen (a perspective division by z near 0) such a face is still
    several pixels high on screen: rows 125-127 at S 256 for the triangle below.  Beyond half a pixel of slack the whole
    image becomes the candidate set."""
    faces = np.zeros((1, 4, 3, 3), np.float32)
    faces[0, 0] = [[-3000, -0.02, 1.0], [3000, -0.02, 1.2], [0, 0.0, 1.1]]
    faces[0, 1] = [[-0.02, -2000, 2.0], [0.0, 0, 2.0], [-0.02, 2500, 2.0]]          # the same, along y
    faces[0, 2] = [[-40000, 0.3, 1.5], [40000, 0.31, 1.5], [0, 0.45, 1.5]]          # longer still
    faces[0, 3] = [[-0.5, -0.5, 3.0], [0.5, -0.5, 3.0], [0.0, 0.6, 3.0]]            # an ordinary face behind them
    for f in range(3):
        a, b, c = faces[0, f, 0, :2], faces[0, f, 1, :2], faces[0, f, 2, :2]
        if (c[1] - a[1]) * (b[0] - a[0]) < (b[1] - a[1]) * (c[0] - a[0]):
            faces[0, f] = faces[0, f, ::-1]
    rng = np.random.default_rng(17)
    textures = rng.uniform(0, 1, (1, 4, 2, 2, 2, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, 256, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    assert all((fn.face_index_map == f).sum() > 100 for f in range(3))
    check_backward(faces, textures, 256, 1e-3, (True, True, True), seed=18)


@pytest.mark.parametrize('F,S', [(1, 64), (33, 100), (4096, 256), (40000, 96)])
def test_forward_wave_redistribution(F, S):
    """k_face_raster deals a wave's rows, then its inside pixels, to the lanes through LDS lists of 128 / 256 entries: faces of
    ~15 x 15 pixels fill several windows of both lists per wave (64 kept faces -> ~900 rows, ~6000 inside pixels).  The face
    counts cover both wave sizes of the kernel (32 faces below 131 072 faces per launch, 64 above: B * F = 160 000), a launch
    that is not a multiple of either, and a raster that is not a power of two."""
    rng = np.random.default_rng(1000 + F)
    B = 4
    size = 15.0 * 2.0 / S                                     # NDC extent of ~15 pixels
    c = rng.uniform(-0.9, 0.9, (B, F, 1, 2)).astype(np.float32)
    faces = np.zeros((B, F, 3, 3), np.float32)
    faces[..., :2] = c + rng.uniform(-0.5, 0.5, (B, F, 3, 2)).astype(np.float32) * size
    faces[..., 2] = rng.uniform(1.0, 3.0, (B, F, 3)).astype(np.float32)
    flip = ((faces[:, :, 2, 1] - faces[:, :, 0, 1]) * (faces[:, :, 1, 0] - faces[:, :, 0, 0]) <
            (faces[:, :, 1, 1] - faces[:, :, 0, 1]) * (faces[:, :, 2, 0] - faces[:, :, 0, 0]))
    faces[flip] = faces[flip][:, ::-1]                         # all front-facing: every face is kept
    fn = oracle_forward(faces, None, S, 0.1, 100, 1e-4, None, False, True, True)
    fw = abi.forward(faces, None, S, return_alpha=True, return_depth=True, want_face_inv=True)
    check_forward(fw, fn)


def test_many_faces_more_than_one_round():
    """F > 1024 exercises several scan rounds of the tile kernel; dense overlap exercises the tie rule."""
    rng = np.random.default_rng(4)
    faces = H.random_scene(rng, 2, 3000, spread=0.9, size=0.15)
    fn = oracle_forward(faces, None, 96, 0.1, 100, 1e-4, None, False, True, True)
    fw = abi.forward(faces, None, 96, return_alpha=True, return_depth=True, want_face_inv=True)
    check_forward(fw, fn)


# ---------------------------------------------------------------------------------------------------
def grads_for(fn, rng, rgb=True, alpha=True, depth=True):
    s = fn.face_index_map.shape
    g_rgb = rng.normal(size=s + (3,)).astype(np.float32) if rgb else None
    g_alpha = rng.normal(size=s).astype(np.float32) if alpha else None
    g_depth = rng.normal(size=s).astype(np.float32) if depth else None
    return g_rgb, g_alpha, g_depth


def check_backward(faces, textures, S, eps, modes, seed, residual_maps=False, ts_bg=(0.2, 0.4, 0.6), k6_flags=0, near=0.1,
                   far=100):
    rgb, alpha, depth = modes
    rng = np.random.default_rng(seed)
    fn = oracle_forward(faces, textures, S, near, far, eps, ts_bg, rgb, alpha, depth)
    fw = abi.forward(faces, textures, S, float(near), float(far), eps, ts_bg, 0, rgb, alpha, depth,
                     want_face_inv=residual_maps and depth, want_sampling=residual_maps and rgb)
    check_forward(fw, fn)
    # the forward's per-face flags: exactly the faces that own a pixel
    vis = abi.host(fw['visible_faces']).astype(bool)
    ref_vis = np.zeros_like(vis)
    bi = np.broadcast_to(np.arange(vis.shape[0])[:, None, None], fn.face_index_map.shape)
    ref_vis[bi[fn.face_index_map >= 0], fn.face_index_map[fn.face_index_map >= 0]] = True
    np.testing.assert_array_equal(vis, ref_vis)
    g_rgb, g_alpha, g_depth = grads_for(fn, rng, rgb, alpha, depth)
    ref = fn.backward(g_rgb, g_alpha, g_depth)
    ref_gf, ref_gt = ref[0].copy(), (ref[1].copy() if rgb else None)
    # same per-pixel float terms, sums carried in double: isolates term arithmetic from summation order
    ref_dd = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True)
    ref_d, ref_gt_d = ref_dd[0].copy(), (ref_dd[1].copy() if rgb else None)
    noise = H.rel_err(ref_gf, ref_d)       # the reference's own serial-float-sum rounding noise
    err_f = None
    # default kernel and NR_FLAG_EXACT_GRADIENT; with and without the forward's visible-face flags (same bits)
    for flags, bound in ((k6_flags, K6_BOUND_DEFAULT), (k6_flags | EXACT, K6_BOUND_EXACT)):
        gf, gt = abi.backward(fw, g_rgb, g_alpha, g_depth, use_sampling_maps=residual_maps,
                              use_face_inv_map=residual_maps, k6_flags=flags)
        gf = abi.host(gf)
        assert not np.isnan(gf).any()
        err_d = H.rel_err(gf, ref_d)           # ours vs the exactly-summed terms
        err_f = H.rel_err(gf, ref_gf)          # ours vs the literal reference order
        ok = np.abs(ref_d) > 0
        elementwise = float(np.mean(np.abs(gf[ok] - ref_d[ok]) <= RTOL * np.abs(ref_d[ok]))) if ok.any() else 1.0
        report('check_backward', S=S, modes=list(modes), flags=flags, err_vs_double_sum=err_d, err_vs_float_order=err_f,
               reference_sum_noise=noise, frac_within_1e4_elementwise=elementwise)
        if depth:
            bound = max(bound, 1e-5)  # K8's float partial sums
        if flags & K6_GLOBAL:
            bound = K6_BOUND_EXACT
        assert err_d <= bound, 'grad_faces vs double-summed oracle: %g (flags %d)' % (err_d, flags)
        assert err_f <= RTOL + 2 * noise, 'grad_faces rel err %g (reference summation noise %g)' % (err_f, noise)
        # back faces and z (when depth is off) are exactly zero, like the reference
        if not depth:
            assert np.all(gf[..., 2] == 0)
        if rgb or alpha:
            gf2, _ = abi.backward(fw, g_rgb, g_alpha, g_depth, use_sampling_maps=residual_maps,
                                  use_face_inv_map=residual_maps, k6_flags=flags, use_visible=False)
            # same terms either way; what can differ is the order of the line records, hence which segment sums share a float
            # run sum before the double atomics (and the order of K8's float adds): a few 1e-7 of the largest gradient
            assert H.rel_err(abi.host(gf2), gf) <= SAME_TERMS
            # the band kernel's second way to its line records (in-kernel face scan, the fallback of images whose records
            # exceed the buffer): same terms again, in both arithmetic modes
            gf3, _ = abi.backward(fw, g_rgb, g_alpha, g_depth, use_sampling_maps=residual_maps,
                                  use_face_inv_map=residual_maps, k6_flags=flags | K6_SCAN)
            gf3 = abi.host(gf3)
            # (the scan path lives in k_bpm_fast: in the default mode its terms are compared with the same kernel's behind
            # NR_FLAG_K6_LEGACY below -- same terms, another order -- and with k_bpm_row's here as two kernels' roundings of
            # the same quantity; the exact mode's terms are the same bits on either kernel)
            same_kernel = bool(flags & (EXACT | K6_GLOBAL | K6_LEGACY))
            assert H.rel_err(gf3, gf) <= (SAME_TERMS if same_kernel else K6_BOUND_DEFAULT)
            # ... and the same mode on the other band kernel (the call above ran k_bpm_row wherever its band fits;
            # NR_FLAG_K6_LEGACY: k_bpm_fast): against the oracle and against k_bpm_row's
            if not (flags & (K6_GLOBAL | K6_SCAN | K6_LEGACY | K6_PX)):
                for kflag, kname in ((K6_LEGACY, 'k_bpm_fast'),):
                    gf4, _ = abi.backward(fw, g_rgb, g_alpha, g_depth, use_sampling_maps=residual_maps,
                                          use_face_inv_map=residual_maps, k6_flags=flags | kflag)
                    gf4 = abi.host(gf4)
                    assert not np.isnan(gf4).any()
                    err_k = H.rel_err(gf4, ref_d)
                    report('check_backward_' + kname, S=S, modes=list(modes), flags=flags | kflag, err_vs_double_sum=err_k,
                           vs_default=H.rel_err(gf4, gf))
                    assert err_k <= bound, 'grad_faces (%s) vs double-summed oracle: %g' % (kname, err_k)
                    # (two kernels, two ways to round a term -- k_bpm_fast steps t = d1 - d1_cross by additions of 1 along a
                    # piece and forms sum (I - ref) g, k_bpm_row subtracts per pixel like the reference and forms the colour
                    # difference from centred sums: not the same terms in another order, so not SAME_TERMS; each kernel is
                    # within `bound` of the oracle)
                    assert H.rel_err(gf4, gf) <= (K6_BOUND_EXACT if flags & EXACT else K6_BOUND_DEFAULT)
                    assert H.rel_err(gf3, gf4) <= SAME_TERMS  # (the scan path against the line-setup path of the same kernel)
                    if (flags & EXACT) and not depth:
                        # the exact mode: both kernels form the reference's float terms bit for bit and add them in double, so
                        # grad_faces can differ only where a double sum sits within ~1e-16 of a float rounding point (allowed:
                        # two entries).  A term rounded differently would move entries by the hundred -- this is what checks
                        # k_bpm_row's `x * 2. / S` without the division (rasters that are no power of two) against k_bpm_fast's
                        # literal one
                        n_diff = int((gf4 != gf).sum())
                        assert n_diff <= 2, 'exact mode: k_bpm_row and k_bpm_fast differ in %d entries' % n_diff
    if rgb:
        gt = abi.host(gt)
        assert not np.isnan(gt).any(), 'grad_textures has unwritten elements'
        noise_t = H.rel_err(ref_gt, ref_gt_d)
        # K7 keeps per-lane float partial sums (like the reference's float atomics): float-sum noise applies
        assert H.rel_err(gt, ref_gt_d) <= RTOL, 'grad_textures vs double-summed oracle'
        err_t = H.rel_err(gt, ref_gt)
        assert err_t <= RTOL + 2 * noise_t, 'grad_textures rel err %g (summation noise %g)' % (err_t, noise_t)
    return err_f


@pytest.mark.parametrize('modes', [(False, True, False), (True, False, False), (False, False, True),
                                   (True, True, True)], ids=['alpha', 'rgb', 'depth', 'all'])
def test_teapot_views_backward(modes):
    faces, _ = H.teapot_views(3, 128)
    rng = np.random.default_rng(5)
    textures = rng.uniform(0, 1, (3, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    check_backward(faces, textures, 128, 1e-3, modes, seed=6)


@pytest.mark.parametrize('S,modes', [(300, (True, True, False)), (512, (True, True, False)), (512, (False, True, False)),
                                     (1024, (True, False, False))], ids=['S300', 'S512', 'S512_alpha', 'S1024_rgb'])
def test_k6_row_kernel_band_shapes(S, modes):
    """k_bpm_row keeps up to 1024 pixels of a band in LDS, a line per wave: four lines per workgroup up to raster 256, two at 512
    (the reference's default raster: anti-aliasing; small launches narrow the bands further and deal a line's windows to
    several waves), one line of 64 segments at 1024; 300 is no multiple of 16: a line ends in 20 padding pixels.  check_backward
    runs k_bpm_row (the default) and k_bpm_fast (NR_FLAG_K6_LEGACY) against the oracle and against each other."""
    faces, _ = H.teapot_views(2, S)
    rng = np.random.default_rng(300 + S)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    check_backward(faces, textures, S, 1e-3, modes, seed=301 + S)


def test_baseline_config1_teapot_64_silhouette():
    """BASELINE.json configs[0]: teapot, 1 view, 64x64 silhouette (the case tests/test_numpy_naive.py runs through the
    naive NumPy per-pixel loop on the CPU)."""
    faces, _ = H.teapot_views(1, 64)
    check_backward(faces, None, 64, 1e-4, (False, True, False), seed=21)


def test_baseline_config2_teapot_16_views_full_size():
    """BASELINE.json configs[1] at full size: 16 azimuth views, 256x256, RGB + depth + silhouette, fwd + bwd, every
    element against the oracle."""
    faces, _ = H.teapot_views(16, 256)
    rng = np.random.default_rng(22)
    textures = rng.uniform(0, 1, (16, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    check_backward(faces, textures, 256, 1e-3, (True, True, True), seed=23)


def test_backward_with_reference_style_residual_maps():
    """K7 fed by sampling maps and K8 fed by face_inv_map (the reference's residuals) instead of recomputation."""
    faces, _ = H.teapot_views(2, 64)
    rng = np.random.default_rng(7)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
    check_backward(faces, textures, 64, 1e-3, (True, True, True), seed=8, residual_maps=True)


@pytest.mark.parametrize('S', [33, 64])
def test_edge_cases_backward(S):
    rng = np.random.default_rng(9)
    faces = edge_case_scene(rng)
    faces[:, 10, :, 2] = 2.0  # keep depths positive for the depth gradient
    textures = rng.uniform(0, 1, (faces.shape[0], faces.shape[1], 3, 3, 3, 3)).astype(np.float32)
    check_backward(faces, textures, S, 1e-4, (True, True, True), seed=10)


def test_big_triangles_long_sweeps():
    """Few large faces: in-sweeps and out-sweeps are hundreds of pixels long (wave-cooperative path)."""
    rng = np.random.default_rng(11)
    faces = H.random_scene(rng, 2, 12, spread=0.3, size=0.9)
    textures = rng.uniform(0, 1, (2, 12, 2, 2, 2, 3)).astype(np.float32)
    check_backward(faces, textures, 256, 1e-3, (True, True, False), seed=12)


def test_line_buffer_overflow_in_some_images_of_a_batch():
    """Images whose line records exceed the buffer (8 F + 32 S + 1.2 S sqrt(F) per image) beside images that fit: 120 faces that
    span most of a 128 x 128 image have tens of thousands of records against a capacity of ~6 700, the image between them a
    handful.  k_bpm_fast walks such an image by its in-kernel face scan inside the normal grid; behind k_bpm_row the overflow-only
    launch of k_bpm_fast takes it (check_backward runs both kernels, and the exact mode)."""
    rng = np.random.default_rng(321)
    big = H.random_scene(rng, 3, 120, spread=0.4, size=1.2)
    small = H.random_scene(rng, 3, 120, spread=0.5, size=0.05)
    faces = np.stack((big[0], small[1], big[2]))
    textures = rng.uniform(0, 1, (3, 120, 2, 2, 2, 3)).astype(np.float32)
    S, F = 128, 120
    # the records of an image: one per visible face, edge, axis and integer line inside the edge's extent (rasterize.py:567-569)
    fn = oracle_forward(faces, textures, S, 0.1, 100, 1e-3, (0.2, 0.4, 0.6), True, True, False)
    capacity = 8 * F + 32 * S + int(1.2 * S * np.sqrt(F))
    records = []
    for b in range(3):
        vis = np.unique(fn.face_index_map[b][fn.face_index_map[b] >= 0])
        p = (faces[b, vis, :, :2].astype(np.float64) * S + S - 1) / 2  # [V, 3, 2] pixel coordinates
        n = 0
        for e in range(3):
            for ax in range(2):
                lo = np.maximum(np.ceil(np.minimum(p[:, e, ax], p[:, (e + 1) % 3, ax])), 0)
                hi = np.minimum(np.floor(np.maximum(p[:, e, ax], p[:, (e + 1) % 3, ax])), S - 1)
                n += int(np.maximum(hi - lo + 1, 0).sum())
        records.append(n)
    assert min(records[0], records[2]) > 1.5 * capacity and records[1] < capacity // 2, (records, capacity)
    check_backward(faces, textures, S, 1e-3, (True, True, False), seed=322)
    check_backward(faces, textures, S, 1e-3, (True, False, False), seed=323)


@pytest.mark.parametrize('ts,eps', [(2, 1e-3), (2, 1e-10), (3, 1e-3), (6, 1e-3), (9, 1e-3)])
def test_big_faces_every_gather_path(ts, eps):
    """Screen-filling faces mixed with small ones: their texture / depth gradients come from k_backward_big (a workgroup per
    face) in its three texture modes (static taps, LDS accumulators, none) and from the one-face-per-workgroup gather for
    texture_size >= 9; staged and fused backward, all three outputs."""
    rng = np.random.default_rng(200 + ts)
    big = H.random_scene(rng, 2, 6, spread=0.3, size=0.9)
    small = H.random_scene(rng, 2, 60, spread=0.6, size=0.08)
    faces = np.concatenate((small[:, :30], big, small[:, 30:]), axis=1)
    textures = rng.uniform(0, 1, (2, faces.shape[1], ts, ts, ts, 3)).astype(np.float32)
    S = 128
    fn = oracle_forward(faces, textures, S, 0.1, 100, eps, (0.3, 0.1, 0.2), True, True, True)
    fw = abi.forward(faces, textures, S, 0.1, 100.0, eps, (0.3, 0.1, 0.2), 0, True, True, True)
    check_forward(fw, fn)
    g_rgb, g_alpha, g_depth = grads_for(fn, rng)
    ref_gf, ref_gt = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True)
    for run in (abi.backward, abi.backward_fused):
        gf, gt = run(fw, g_rgb, g_alpha, g_depth)
        assert H.rel_err(abi.host(gt), ref_gt) <= RTOL, run.__name__
        assert H.rel_err(abi.host(gf), ref_gf) <= K6_BOUND_DEFAULT, run.__name__


@pytest.mark.parametrize('S', [384, 512, 768, 1024])
def test_band_width_classes(S):
    """Raster sizes whose K6 bands are 2 lines (512, 384: the anti-aliased default of Renderer) or 1 line (768, 1024) wide,
    powers of two and not: every staging / band-width path of the band kernel against the oracle."""
    rng = np.random.default_rng(300 + S)
    faces = H.random_scene(rng, 1, 400, spread=0.7, size=0.15)
    textures = rng.uniform(0, 1, (1, 400, 2, 2, 2, 3)).astype(np.float32)
    check_backward(faces, textures, S, 1e-3, (True, True, True), seed=301 + S)


@pytest.mark.parametrize('S', [4096, 6000])
def test_very_large_raster_alpha_only(S):
    """Raster sizes at which one band line only just fits in LDS (W = 1, alpha only) and the packed segment scan of K6 needs
    shorter line windows (full-segment counts are kept in 16 bits: 256 lines x 2 S / 15 segments would overflow beyond
    S = 1919); 6000 is not a power of two and its 2 x 6000 band counters do not fit the compaction's LDS histogram."""
    rng = np.random.default_rng(4000 + S)
    faces = H.random_scene(rng, 1, 16, spread=0.5, size=0.35)
    check_backward(faces, None, S, 1e-4, (False, True, False), seed=4001 + S)


@pytest.mark.parametrize('S', [2048, 2600])
def test_large_raster_rgb_one_line_bands(S):
    """RGB + alpha at raster sizes whose bands are one line wide (W = 1 from S ~ 1500): 2 x 2048 bands are the most whose
    table k_line_setup keeps in 48 KB of LDS, 2 x 2600 need the raised limit (up to 3072; beyond, the in-kernel line setup)."""
    rng = np.random.default_rng(7000 + S)
    faces = H.random_scene(rng, 1, 24, spread=0.5, size=0.3)
    textures = rng.uniform(0, 1, (1, 24, 2, 2, 2, 3)).astype(np.float32)
    check_backward(faces, textures, S, 1e-3, (True, True, False), seed=7001 + S)


def test_known_answer_gradients_through_renderer():
    """The reference's grad_ref constants (tests/test_rasterize_silhouettes.py:37-99) through the full
    PyTorch-facing API: Renderer -> look_at -> vertices_to_faces -> HIP rasterizer -> autograd."""
    import neural_renderer_amd as nr
    from test_oracle_golden import CASE1, CASE2
    for case in (CASE1, CASE2):
        vertices = torch.tensor(case['vertices'], dtype=torch.float32)
        faces = torch.tensor([[0, 1, 2]], dtype=torch.int32)
        vb, fb = [torch.as_tensor(x) for x in H.to_minibatch((vertices.numpy(), faces.numpy()))]
        vb = vb.cuda().requires_grad_(True)
        renderer = nr.Renderer()
        renderer.image_size = 64
        renderer.anti_aliasing = False
        renderer.perspective = False
        images = renderer.render_silhouettes(vb, fb.cuda())
        loss = torch.sum(torch.abs(images[:, case['pyi'], case['pxi']] - case['target']))
        loss.backward()
        grad = vb.grad.cpu().numpy()
        np.testing.assert_allclose(grad[2], np.array(case['grad_ref']), rtol=1e-2, atol=1e-6)  # reference's rtol
        np.testing.assert_allclose(grad[2], np.array(case['grad_ref']), rtol=1e-4, atol=1e-6)  # ours
        assert np.all(grad[[0, 1, 3]] == 0)


def test_determinism_and_batch_independence():
    """Run twice -> identical forward bits (the packed z-buffer minimum does not depend on the order of its atomics); a view
    rendered inside a batch equals the same view rendered alone (the property the multi-GPU sharding relies on).
    K6 sums its per-line / per-face partials with double-precision atomics whose order is not fixed: grad_faces is
    reproducible up to the rounding of a double sum to float (a last-bit difference in rare, heavily cancelling entries),
    and the default kernel forms float sums over the
    segments that happen to share a run of lanes: the gradients are compared to 1e-5 of the largest one, not bit for bit."""
    faces, _ = H.teapot_views(8, 128)
    rng = np.random.default_rng(13)
    g = rng.normal(size=(8, 128, 128)).astype(np.float32)
    outs = []
    for _ in range(2):
        fw = abi.forward(faces, None, 128, return_alpha=True, return_depth=True)
        gf, _ = abi.backward(fw, g_alpha=g)
        outs.append((abi.host(fw['face_index_map']), abi.host(fw['depth_map']), abi.host(gf)))
    for a, b in list(zip(*outs))[:2]:
        np.testing.assert_array_equal(a, b)
    assert H.rel_err(outs[0][2], outs[1][2]) <= SAME_TERMS
    fw1 = abi.forward(faces[5:6], None, 128, return_alpha=True, return_depth=True)
    gf1, _ = abi.backward(fw1, g_alpha=g[5:6])
    np.testing.assert_array_equal(abi.host(fw1['face_index_map'])[0], outs[0][0][5])
    np.testing.assert_array_equal(abi.host(fw1['depth_map'])[0], outs[0][1][5])
    assert H.rel_err(abi.host(gf1)[0], outs[0][2][5]) <= SAME_TERMS


def test_headline_size_properties():
    """BASELINE.json full size (teapot, 64 views, 256x256): the oracle checks 2 of the 64 views (it is a
    brute-force O(pixels x faces) CPU loop); all 64 are checked through size-independent properties."""
    B, S = 64, 256
    faces, _ = H.teapot_views(B, S)
    fw = abi.forward(faces, None, S, return_alpha=True, return_depth=True)
    fi = abi.host(fw['face_index_map'])
    alpha = abi.host(fw['alpha_map'])
    depth = abi.host(fw['depth_map'])
    F = faces.shape[1]
    assert fi.min() == -1 and fi.max() < F
    np.testing.assert_array_equal(alpha, (fi >= 0).astype(np.float32))
    assert np.all(depth[fi < 0] == 100.0) and np.all((depth[fi >= 0] > 0.1) & (depth[fi >= 0] < 100.0))
    # only front-facing faces can win a pixel
    f0 = faces.reshape(B, F, 9)
    back = (f0[..., 7] - f0[..., 1]) * (f0[..., 3] - f0[..., 0]) < (f0[..., 4] - f0[..., 1]) * (f0[..., 6] - f0[..., 0])
    bi = np.repeat(np.arange(B), S * S).reshape(B, S, S)
    assert not back[bi[fi >= 0], fi[fi >= 0]].any()
    # coverage of each view is within the range the survey measured for the teapot (11.6-12.8 % at this distance)
    cov = alpha.reshape(B, -1).mean(1)
    assert 0.05 < cov.min() and cov.max() < 0.25
    for i in (0, 37):
        fn = oracle_forward(faces[i:i + 1], None, S, 0.1, 100, 1e-4, None, False, True, True)
        assert int((fi[i] != fn.face_index_map[0]).sum()) == 0
        np.testing.assert_array_equal(depth[i], fn.depth_map[0])
    # backward at full size: gradient of sum(alpha * c) is finite, zero on back faces, and linear in c
    rng = np.random.default_rng(14)
    g = rng.normal(size=(B, S, S)).astype(np.float32)
    gf1, _ = abi.backward(fw, g_alpha=g)
    gf2, _ = abi.backward(fw, g_alpha=2 * g)
    gf1, gf2 = abi.host(gf1), abi.host(gf2)
    assert np.isfinite(gf1).all() and np.all(gf1[back] == 0) and np.all(gf1[..., 2] == 0)
    # (linear up to the summation order of two runs: float run sums in front of the double accumulators)
    assert H.rel_err(gf2, 2 * gf1) <= SAME_TERMS
    for i in (0, 37):
        fn = oracle_forward(faces[i:i + 1], None, S, 0.1, 100, 1e-4, None, False, True, False)
        ref_d, = fn.backward(None, g[i:i + 1], None, accumulate_double=True)
        assert H.rel_err(gf1[i], ref_d[0]) <= K6_BOUND_DEFAULT


def test_exact_gradient_through_the_operator():
    """`Rasterize.exact_gradient` (env NR_EXACT_GRADIENT) selects NR_FLAG_EXACT_GRADIENT through the autograd operator:
    the default is within the north star's 1e-4 of the exactly summed reference terms, the exact mode within 2e-6."""
    import neural_renderer_amd as nr
    faces, _ = H.teapot_views(2, 128)
    rng = np.random.default_rng(21)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, 128, 0.1, 100, 1e-3, (0.2, 0.4, 0.6), True, True, False)
    g_rgb, g_alpha, _ = grads_for(fn, rng, True, True, False)
    ref_d, _ = fn.backward(g_rgb, g_alpha, None, accumulate_double=True)
    errs = {}
    for exact in (False, True):
        ft = torch.tensor(faces, device='cuda', requires_grad=True)
        op = nr.Rasterize(128, 0.1, 100, 1e-3, (0.2, 0.4, 0.6), True, True, False)
        op.exact_gradient = exact
        rgb, alpha, _ = op(ft, torch.tensor(textures, device='cuda'))
        torch.autograd.backward([rgb, alpha], [torch.tensor(g_rgb, device='cuda'), torch.tensor(g_alpha, device='cuda')])
        errs[exact] = H.rel_err(ft.grad.cpu().numpy(), ref_d)
    report('exact_vs_default', default=errs[False], exact=errs[True])
    assert errs[True] <= K6_BOUND_EXACT and errs[False] <= K6_BOUND_DEFAULT


def test_global_memory_k6_fallback():
    """The band pipeline needs one band line of all maps in LDS; rasters too large for that use the
    global-memory kernel (k_bpm_global).  NR_FLAG_K6_GLOBAL forces it so that it stays covered."""
    faces, _ = H.teapot_views(2, 96)
    rng = np.random.default_rng(23)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    rgb, alpha, depth = True, True, False
    fn = oracle_forward(faces, textures, 96, 0.1, 100, 1e-3, (0.2, 0.4, 0.6), rgb, alpha, depth)
    fw = abi.forward(faces, textures, 96, 0.1, 100.0, 1e-3, (0.2, 0.4, 0.6), 0, rgb, alpha, depth)
    g_rgb, g_alpha, _ = grads_for(fn, rng, rgb, alpha, depth)
    ref_d, _ = fn.backward(g_rgb, g_alpha, None, accumulate_double=True)
    gf, _ = abi.backward(fw, g_rgb, g_alpha, None, k6_flags=K6_GLOBAL)
    assert H.rel_err(abi.host(gf), ref_d) <= K6_BOUND_EXACT


@pytest.mark.parametrize('S', [64, 256])
def test_unsafe_rasterizer_against_the_k3_oracle(S):
    """SURVEY 8 row a3': with `use_unsafe_rasterizer(True)` the reference runs K3 (rasterize.py:102-236: per-face scan
    conversion, x-sorted face_inv, spin-lock z-buffer).  The one rasterizer of this library serves that switch too; against
    the oracle's restatement of K3 (sequential emulation, pinned by tests/test_oracle_golden.py) on the teapot under the
    default camera it must give the same face indices and coverage, depth within 2.6e-6 and weights within 1e-4 -- the
    differences SURVEY Appendix B measured between the reference's two kernels (K3 builds face_inv from the x-sorted vertices)."""
    import neural_renderer_amd as nr
    v, f = H.teapot()
    faces = O.Renderer().project(v[None], f[None])
    k3 = O.Rasterize(S, 0.1, 100, 1e-4, None, False, True, True)
    k3.unsafe = True
    _, ref_alpha, ref_depth = k3(faces)
    try:
        nr.use_unsafe_rasterizer(True)
        fn = nr.Rasterize(S, 0.1, 100, 1e-4, (0, 0, 0), False, True, True)
        _, alpha, depth = fn(torch.tensor(faces, device='cuda'))
        fi = fn.face_index_map.cpu().numpy()
        wm = fn.weight_map.cpu().numpy()
        fim = fn.face_inv_map.cpu().numpy()
    finally:
        nr.use_unsafe_rasterizer(False)
    assert int((fi != k3.face_index_map).sum()) == 0
    np.testing.assert_array_equal(alpha.cpu().numpy(), ref_alpha)
    d_err = float(np.abs(depth.cpu().numpy() - ref_depth).max())
    w_err = float(np.abs(wm - k3.weight_map).max())
    report('unsafe_vs_k3_oracle', S=S, depth_abs=d_err, weight_abs=w_err)
    assert d_err <= 2.7e-6
    assert w_err <= 1.1e-4
    assert float(np.abs(fim - k3.face_inv_map).max()) <= 1e-5 * float(np.abs(k3.face_inv_map).max())
    # What a user of the switch observes beyond the maps: colours and texture gradients.  K3's weights (x-sorted vertices,
    # rasterize.py:210-214) feed K4's texture coordinates (:399-404): tif = w (ts - 1) depth / z, so a weight difference of 1e-4
    # moves a texture coordinate by (ts - 1) 1e-4 and a trilinear sample of textures in [0, 1] by at most 3 x that.
    ts = 4
    rng = np.random.default_rng(1300 + S)
    textures = rng.uniform(0, 1, (1, faces.shape[1], ts, ts, ts, 3)).astype(np.float32)
    g_rgb = rng.normal(size=(1, S, S, 3)).astype(np.float32)
    k3c = O.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, False, False)
    k3c.unsafe = True
    ref_rgb = k3c(faces, textures)[0]
    ref_gt = k3c.backward(g_rgb, None, None, accumulate_double=True)[1]
    try:
        nr.use_unsafe_rasterizer(True)
        fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, False, False)
        tt = torch.tensor(textures, device='cuda', requires_grad=True)
        rgb = fn(torch.tensor(faces, device='cuda'), tt)[0]
        rgb.backward(torch.tensor(g_rgb, device='cuda'))
    finally:
        nr.use_unsafe_rasterizer(False)
    c_err = float(np.abs(rgb.detach().cpu().numpy() - ref_rgb).max())
    gt_err = H.rel_err(tt.grad.cpu().numpy(), ref_gt)
    report('unsafe_vs_k3_oracle_colours', S=S, ts=ts, rgb_abs=c_err, grad_textures_rel=gt_err)
    assert c_err <= 3 * (ts - 1) * 1.1e-4
    # (grad_textures[texel] = sum over the face's pixels of corner weight x g: every corner weight moves by <= 3 (ts - 1) 1.1e-4 =
    # 1e-3 -- against corner weights of ~0.1 and gradients of random sign that is ~1e-2 of a texel's gradient; measured 0.6 ... 1.3e-2)
    assert gt_err <= 2e-2


def test_unsafe_rasterizer_flag_is_equivalent(monkeypatch):
    """SURVEY 8 row a3' (rasterize.py:15-16, :1063-1065): `use_unsafe_rasterizer(True)` and NEURAL_RENDERER_UNSAFE=1 keep the
    API and select the same rasterizer: images and texture gradients are bit-identical to the default setting, vertex gradients
    equal up to K6's run-to-run summation order."""
    import importlib
    import sys
    import neural_renderer_amd as nr
    R = sys.modules['neural_renderer_amd.rasterize']  # (the package attribute `rasterize` is the function)
    faces, _ = H.teapot_views(2, 96)
    rng = np.random.default_rng(91)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    g = {k: rng.normal(size=s).astype(np.float32) for k, s in
         (('rgb', (2, 3, 96, 96)), ('alpha', (2, 96, 96)), ('depth', (2, 96, 96)))}

    def run():
        ft = torch.tensor(faces, device='cuda', requires_grad=True)
        tt = torch.tensor(textures, device='cuda', requires_grad=True)
        out = nr.rasterize_rgbad(ft, tt, 96, False, 0.1, 100, 1e-3, (0.3, 0.2, 0.1), True, True, True)
        torch.autograd.backward([out[k] for k in ('rgb', 'alpha', 'depth')],
                                [torch.tensor(g[k], device='cuda') for k in ('rgb', 'alpha', 'depth')])
        return [out[k].detach().cpu().numpy() for k in ('rgb', 'alpha', 'depth')] + [ft.grad.cpu().numpy(),
                                                                                      tt.grad.cpu().numpy()]

    base = run()
    assert R.USE_UNSAFE_IMPLEMENTATION is False
    try:
        nr.use_unsafe_rasterizer(True)
        assert R.USE_UNSAFE_IMPLEMENTATION is True
        flagged = run()
    finally:
        nr.use_unsafe_rasterizer(False)
    for k, (a, b) in enumerate(zip(base, flagged)):
        if k == 3:  # grad_faces: the same terms, summed in the order two runs of K6 happen to place their line records in
            assert H.rel_err(a, b) <= SAME_TERMS
        else:
            np.testing.assert_array_equal(a, b)
    # the environment variable is read at import (rasterize.py:15-16)
    monkeypatch.setenv('NEURAL_RENDERER_UNSAFE', '1')
    try:
        importlib.reload(R)
        assert R.USE_UNSAFE_IMPLEMENTATION is True
        ft = torch.tensor(faces, device='cuda')
        img = R.rasterize_silhouettes(ft, 96, False).cpu().numpy()
        np.testing.assert_array_equal(img, base[1])
    finally:
        monkeypatch.delenv('NEURAL_RENDERER_UNSAFE')
        importlib.reload(R)
    assert R.USE_UNSAFE_IMPLEMENTATION is False


def icosphere(level):
    """Subdivided icosahedron (unit sphere): vertices [Nv,3], faces [Nf,3]."""
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.int32)


def project_mesh(v, f, eye, fill_back=True):
    if fill_back:
        f = np.concatenate((f, f[:, ::-1]), axis=0)
    vv = O.perspective(O.look_at(v[None], eye), 30.)
    return O.vertices_to_faces(vv, f[None])[0]


def test_synthetic_shapenet_scale_meshes():
    """BASELINE.json config 4 at test scale: distinct random meshes of ~5k faces (noisy icospheres, fill_back ->
    10 240 faces), per-sample random rotation, texture_size 4 random textures, 128x128 RGB forward + backward."""
    rng = np.random.default_rng(31)
    v0, f0 = icosphere(4)  # 5120 faces
    batch = []
    for _ in range(3):
        v = v0 * (0.55 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        batch.append(project_mesh((v @ q).astype(np.float32), f0, [0.3, 0.4, -2.6]))
    faces = np.stack(batch)
    textures = rng.uniform(0, 1, (3, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
    check_backward(faces, textures, 128, 1e-3, (True, False, False), seed=32)


def test_high_resolution_dense_mesh():
    """BASELINE.json config 5 at test scale: one dense mesh (20 480 -> 40 960 faces with fill_back) whose faces
    are mostly sub-pixel to a few pixels, texture_size 8, 192x192 (non power of two), all outputs."""
    rng = np.random.default_rng(33)
    v0, f0 = icosphere(5)
    v = v0 * (0.6 + 0.05 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
    faces = project_mesh(v.astype(np.float32), f0, [0.0, 0.0, -2.4])[None]
    textures = rng.uniform(0, 1, (1, faces.shape[1], 8, 8, 8, 3)).astype(np.float32)
    check_backward(faces, textures, 192, 1e-3, (True, True, True), seed=34)


@pytest.mark.parametrize('modes', [(True, True, True), (False, True, False), (False, False, True), (True, False, False)],
                         ids=['all', 'alpha', 'depth', 'rgb'])
def test_fused_backward_equals_stage_calls(modes):
    """nr_backward_rasterize == nr_backward_pixel_map + nr_backward_textures + nr_backward_depth_map: the same kernels on the
    same data, so grad_textures and the depth-only grad_faces agree bit for bit; with K6 in play grad_faces agrees up to the
    order of its line records (placed with atomics: which segment sums share a float run sum, the order of the double adds)."""
    rgb, alpha, depth = modes
    faces, _ = H.teapot_views(3, 96)
    rng = np.random.default_rng(41)
    textures = rng.uniform(0, 1, (3, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    fw = abi.forward(faces, textures, 96, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, rgb, alpha, depth)
    g_rgb = rng.normal(size=(3, 96, 96, 3)).astype(np.float32) if rgb else None
    g_alpha = rng.normal(size=(3, 96, 96)).astype(np.float32) if alpha else None
    g_depth = rng.normal(size=(3, 96, 96)).astype(np.float32) if depth else None
    gf_a, gt_a = abi.backward(fw, g_rgb, g_alpha, g_depth)
    gf_b, gt_b = abi.backward_fused(fw, g_rgb, g_alpha, g_depth)
    if rgb or alpha:
        assert H.rel_err(abi.host(gf_a), abi.host(gf_b)) <= SAME_TERMS
    else:
        np.testing.assert_array_equal(abi.host(gf_a), abi.host(gf_b))
    if rgb:
        np.testing.assert_array_equal(abi.host(gt_a), abi.host(gt_b))


@pytest.mark.parametrize('ts', [2, 3, 4, 7, 10, 14])
@pytest.mark.parametrize('modes', [(True, True, True), (True, False, False), (True, False, True)], ids=['all', 'rgb', 'rgb_depth'])
def test_fused_launch_order_does_not_change_the_backward(ts, modes):
    """nr_backward_rasterize puts the K7 / K8 gather and grad_textures' zeros into the launch of K6's line setup, in front of the
    band kernel (default), or runs line setup, band kernel and gather one after the other (NR_FLAG_SERIAL_BACKWARD): the same
    kernel bodies on the same data, K6's rounded sums and K8's meeting in one float addition per element either way --
    grad_textures bit for bit (outputs pre-filled with NaN: every zero is stored), grad_faces up to the order of K6's line
    records (as between any two runs); also with K6's in-kernel face scan, where there is no line setup to share a launch with."""
    rgb, alpha, depth = modes
    faces, _ = H.teapot_views(3, 96)
    rng = np.random.default_rng(43 + ts)
    textures = rng.uniform(0, 1, (3, faces.shape[1], ts, ts, ts, 3)).astype(np.float32)
    fw = abi.forward(faces, textures, 96, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, rgb, alpha, depth)
    g_rgb = rng.normal(size=(3, 96, 96, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(3, 96, 96)).astype(np.float32) if alpha else None
    g_depth = rng.normal(size=(3, 96, 96)).astype(np.float32) if depth else None
    gf_s, gt_s = abi.backward_fused(fw, g_rgb, g_alpha, g_depth, k6_flags=SERIAL)
    # (use_visible False: K6 rebuilds the flags itself; K6_GLOBAL: no band pipeline, hence no lists and no launch to share)
    for flags, use_visible in ((0, True), (0, False), (K6_SCAN, True), (EXACT, True), (K6_GLOBAL, True)):
        gf_o, gt_o = abi.backward_fused(fw, g_rgb, g_alpha, g_depth, k6_flags=flags, use_visible=use_visible)
        assert not np.isnan(abi.host(gf_o)).any()
        assert H.rel_err(abi.host(gf_s), abi.host(gf_o)) <= (SAME_TERMS if flags in (0, K6_SCAN) else K6_BOUND_DEFAULT)
        np.testing.assert_array_equal(abi.host(gt_s), abi.host(gt_o))


@pytest.mark.parametrize('ts,eps', [(2, 0.0), (2, 1e-10), (5, 1e-3), (6, 1e-3), (9, 1e-3), (13, 1e-4), (14, 1e-3), (16, 1e-3)])
def test_texture_size_paths(ts, eps):
    """K4 / K7 over every dispatch class of the texture gradient: texture_size 2 with eps = 0 or below float32 resolution (an index float can then reach 1.0 exactly, so the static-tap fast
    path must not be taken and the zero-weight taps outside the cube must not be touched), 3-5 (16 lanes per face), 6-8 (64), 9-13 (256), >= 14 (per-pixel scatter
    with hardware float atomics, the reference's own formulation) -- staged and fused backward against the oracle."""
    rng = np.random.default_rng(100 + ts)
    B, F, S = 2, 48, 64
    faces = H.random_scene(rng, B, F, spread=0.5, size=0.35)
    textures = rng.uniform(0, 1, (B, F, ts, ts, ts, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, S, 0.1, 100, eps, (0.3, 0.1, 0.2), True, True, True)
    fw = abi.forward(faces, textures, S, 0.1, 100.0, eps, (0.3, 0.1, 0.2), 0, True, True, True)
    check_forward(fw, fn)
    g_rgb, g_alpha, g_depth = grads_for(fn, rng)
    ref_gf, ref_gt = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True)
    for run in (abi.backward, abi.backward_fused):
        gf, gt = run(fw, g_rgb, g_alpha, g_depth)
        gf, gt = abi.host(gf), abi.host(gt)
        assert not np.isnan(gt).any() and not np.isnan(gf).any()
        assert H.rel_err(gt, ref_gt) <= RTOL, (run.__name__, H.rel_err(gt, ref_gt))
        assert H.rel_err(gf, ref_gf) <= K6_BOUND_DEFAULT, (run.__name__, H.rel_err(gf, ref_gf))


def test_culled_images_and_per_batch_background():
    """One image whose faces are all back-facing, one whose faces all lie beyond `far` / before `near`, one ordinary; a
    different background colour per image (rasterize.py:464-465), which K6 sees through the post-background rgb_map."""
    rng = np.random.default_rng(77)
    B, F, S, eps = 3, 60, 48, 1e-3
    faces = H.random_scene(rng, B, F, spread=0.5, size=0.3)
    front = ((faces[0, :, 2, 1] - faces[0, :, 0, 1]) * (faces[0, :, 1, 0] - faces[0, :, 0, 0])
             >= (faces[0, :, 1, 1] - faces[0, :, 0, 1]) * (faces[0, :, 2, 0] - faces[0, :, 0, 0]))
    faces[0, front] = faces[0, front][:, ::-1]          # image 0: every face turned away from the camera
    faces[1, : F // 2, :, 2] = 150.0                    # image 1: beyond far = 100 ...
    faces[1, F // 2:, :, 2] = 0.05                      # ... or closer than near = 0.1
    textures = rng.uniform(0, 1, (B, F, 2, 2, 2, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, S, 0.1, 100, eps, bg, True, True, True)
    assert (fn.face_index_map[0] == -1).all() and (fn.face_index_map[1] == -1).all() and (fn.face_index_map[2] >= 0).any()
    fw = abi.forward(faces, textures, S, 0.1, 100.0, eps, bg, 0, True, True, True)
    check_forward(fw, fn)
    g_rgb, g_alpha, g_depth = grads_for(fn, rng)
    ref_gf, ref_gt = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True)
    for run in (abi.backward, abi.backward_fused):
        gf, gt = run(fw, g_rgb, g_alpha, g_depth)
        gf, gt = abi.host(gf), abi.host(gt)
        assert np.all(gf[:2] == 0) and np.all(gt[:2] == 0)      # nothing visible: exact zeros, no NaN
        assert H.rel_err(gf, ref_gf) <= K6_BOUND_DEFAULT and H.rel_err(gt, ref_gt) <= RTOL


def test_nan_inf_and_zero_depth_vertices_like_the_reference():
    """NaN / Inf coordinates, astronomically large ones and a zero depth: a face whose depth comes out NaN passes the
    near / far test (rasterize.py:331) but can never win a pixel (`zp < depth_min`, :334); maps must agree bit for bit with the
    oracle (NaN == NaN) and the gradients must be NaN in exactly the same places.  No hang, no fault."""
    rng = np.random.default_rng(5)
    B, F, S, eps = 2, 40, 48, 1e-3
    faces = H.random_scene(rng, B, F, spread=0.5, size=0.3)
    faces[0, 3, 1, 0] = np.nan
    faces[0, 7, 2, 1] = np.inf
    faces[0, 9, 0, 2] = np.nan
    faces[1, 2, :, :2] *= 1e20
    faces[1, 5, 0, 0] = -np.inf
    faces[1, 11, 1, 2] = 0.0
    textures = rng.uniform(0, 1, (B, F, 2, 2, 2, 3)).astype(np.float32)
    fn = oracle_forward(faces, textures, S, 0.1, 100, eps, (0.1, 0.2, 0.3), True, True, True)
    fw = abi.forward(faces, textures, S, 0.1, 100.0, eps, (0.1, 0.2, 0.3), 0, True, True, True)
    assert int((abi.host(fw['face_index_map']) != fn.face_index_map).sum()) == 0
    for k in ('weight_map', 'depth_map', 'rgb_map', 'alpha_map'):
        assert np.array_equal(abi.host(fw[k]), getattr(fn, k), equal_nan=True), k
    g_rgb, g_alpha, g_depth = grads_for(fn, rng)
    ref_gf, ref_gt = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True)
    for run in (abi.backward, abi.backward_fused):
        gf, gt = run(fw, g_rgb, g_alpha, g_depth)
        gf, gt = abi.host(gf), abi.host(gt)
        assert np.array_equal(np.isnan(gf), np.isnan(ref_gf)) and np.array_equal(np.isnan(gt), np.isnan(ref_gt))
        ok = np.isfinite(ref_gf)
        assert H.rel_err(gf[ok], ref_gf[ok]) <= K6_BOUND_DEFAULT
        ok = np.isfinite(ref_gt)
        assert H.rel_err(gt[ok], ref_gt[ok]) <= RTOL


def test_fused_forward_equals_stage_calls():
    """nr_forward_rasterize == nr_forward_face_index_map + nr_forward_texture_sampling, bit for bit."""
    faces, _ = H.teapot_views(3, 100)
    rng = np.random.default_rng(43)
    textures = rng.uniform(0, 1, (3, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, (3, 3)).astype(np.float32)
    a = abi.forward(faces, textures, 100, 0.1, 100.0, 1e-3, bg, 1, True, True, True)
    b = abi.forward_fused(faces, textures, 100, 0.1, 100.0, 1e-3, bg, 1, True, True, True)
    for k in ('face_index_map', 'weight_map', 'depth_map', 'rgb_map', 'alpha_map'):
        np.testing.assert_array_equal(abi.host(a[k]), abi.host(b[k]), err_msg=k)


def test_kept_workspace_epochs_equal_the_filling_forward():
    """NR_FLAG_ZBUF_EPOCH (include/nr_hip.h): a forward workspace kept between calls, filled with 0xff once and used with
    falling epoch numbers, gives the maps of the per-call-fill path bit for bit -- over a whole cycle of 255 epochs with the
    scene changing from call to call (stale words of every earlier scene must read as empty), across the refill, with queued
    large faces (the queue counters are reset by the resolve pass), and the operator (which runs this way) agrees too."""
    from neural_renderer_amd import _lib
    import neural_renderer_amd as nr
    rng = np.random.default_rng(77)
    S = 48
    scenes = []
    for k in range(4):
        f = H.random_scene(rng, 2, 60, size=0.2 + 0.2 * k)
        f[:, 0, :, :2] *= 6.0   # a face with a large screen box: goes through the queues
        scenes.append(f)
    tex = rng.uniform(0, 1, (2, 60, 2, 2, 2, 3)).astype(np.float32)
    ref = [abi.forward_fused(f, tex, S, 0.1, 100.0, 1e-3, (0.2, 0.3, 0.4), 0, True, True, True) for f in scenes]
    lib = _lib.load()
    ws = torch.empty(lib.nr_forward_workspace_bytes(2, 60, S), dtype=torch.uint8, device='cuda')
    names = ('face_index_map', 'weight_map', 'depth_map', 'rgb_map', 'alpha_map', 'visible_faces')
    call = 0
    for cycle in range(2):
        ws.fill_(255)
        for epoch in range(254, -1, -1):
            k = call % 4
            call += 1
            if cycle == 1 and epoch < 250:
                break  # (the second cycle only has to prove the refill)
            fw = abi.forward_fused(scenes[k], tex, S, 0.1, 100.0, 1e-3, (0.2, 0.3, 0.4), _lib.NR_FLAG_ZBUF_EPOCH | (epoch << 8),
                                   True, True, True, workspace=ws)
            if epoch % 16 == 0 or epoch > 250 or epoch < 3:
                for name in names:
                    np.testing.assert_array_equal(abi.host(fw[name]), abi.host(ref[k][name]), err_msg='%s epoch %d' % (name, epoch))
    # the operator: several calls in a row on changing scenes
    for k in (0, 1, 2, 3, 0):
        rgb, alpha, depth = nr.Rasterize(S, 0.1, 100, 1e-3, (0.2, 0.3, 0.4), True, True, True)(
            torch.tensor(scenes[k], device='cuda'), torch.tensor(tex, device='cuda'))
        np.testing.assert_array_equal(rgb.cpu().numpy(), abi.host(ref[k]['rgb_map']))
        np.testing.assert_array_equal(depth.cpu().numpy(), abi.host(ref[k]['depth_map']))


def test_vertices_to_faces_gather_and_atomic_scatter():
    """nr_vertices_to_faces / _backward (reference vertices_to_faces.py:4-21 + Chainer get_item backward) through the
    torch-facing function: the gather is exact, the scatter-add matches np.add.at up to float summation order."""
    import neural_renderer_amd as nr
    rng = np.random.default_rng(51)
    v, f = H.teapot()
    f2 = np.concatenate((f, f[:, ::-1]), axis=0)
    B = 3
    vb = rng.normal(size=(B,) + v.shape).astype(np.float32)
    fb = np.repeat(f2[None], B, axis=0)
    vt = torch.tensor(vb, device='cuda', requires_grad=True)
    out = nr.vertices_to_faces(vt, torch.tensor(fb, device='cuda'))
    ref = O.vertices_to_faces(vb, fb)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    g = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(torch.tensor(g, device='cuda'))
    gref = np.zeros_like(vb, dtype=np.float64)
    for b in range(B):
        np.add.at(gref[b], fb[b].reshape(-1), g[b].reshape(-1, 3).astype(np.float64))
    # ~12 float atomics per vertex in arbitrary order (as in the reference's scatter_add): float-sum noise only
    assert H.rel_err(vt.grad.cpu().numpy(), gref) <= RTOL


@pytest.mark.parametrize('aa', [False, True], ids=['no_aa', 'aa'])
def test_public_api_rasterize_rgbad_matches_oracle(aa):
    """The torch-facing `rasterize_rgbad` (2x super-sampling + average pooling, vertical flip, NHWC -> NCHW;
    reference rasterize.py:900-977) against the oracle's restatement, images and gradients."""
    import neural_renderer_amd as nr
    faces, _ = H.teapot_views(2, 64)
    rng = np.random.default_rng(61)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    ft = torch.tensor(faces, device='cuda', requires_grad=True)
    tt = torch.tensor(textures, device='cuda', requires_grad=True)
    out = nr.rasterize_rgbad(ft, tt, 64, aa, 0.1, 100, 1e-3, (0.3, 0.2, 0.1), True, True, True)
    ref = O.rasterize_rgbad(faces, textures, 64, aa, 0.1, 100, 1e-3, (0.3, 0.2, 0.1), True, True, True,
                            return_function=True)
    for k in ('rgb', 'alpha', 'depth'):
        assert out[k].shape == ref[k].shape
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref[k], rtol=1e-6, atol=1e-6, err_msg=k)
    g = {k: rng.normal(size=ref[k].shape).astype(np.float32) for k in ('rgb', 'alpha', 'depth')}
    torch.autograd.backward([out['rgb'], out['alpha'], out['depth']],
                            [torch.tensor(g[k], device='cuda') for k in ('rgb', 'alpha', 'depth')])
    gf, gt = O.rgbad_backward(ref['function'], aa, g['rgb'], g['alpha'], g['depth'])
    assert H.rel_err(ft.grad.cpu().numpy(), gf) <= RTOL
    assert H.rel_err(tt.grad.cpu().numpy(), gt) <= RTOL


def test_none_gradients_and_unused_outputs():
    """A loss that uses only some outputs: the others receive `None` gradients, which the reference replaces by
    zeros (rasterize.py:858-878)."""
    import neural_renderer_amd as nr
    faces, _ = H.teapot_views(2, 64)
    rng = np.random.default_rng(62)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    ft = torch.tensor(faces, device='cuda', requires_grad=True)
    tt = torch.tensor(textures, device='cuda', requires_grad=True)
    rgb, alpha, depth = nr.Rasterize(64, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(ft, tt)
    g_alpha = rng.normal(size=(2, 64, 64)).astype(np.float32)
    alpha.backward(torch.tensor(g_alpha, device='cuda'))
    ref = oracle_forward(faces, textures, 64, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    gf, gt = ref.backward(None, g_alpha, None)
    assert H.rel_err(ft.grad.cpu().numpy(), gf) <= RTOL
    assert tt.grad is None or float(tt.grad.abs().max()) == 0.0
    assert np.all(gt == 0)


def test_graph_replay_equals_the_eager_operator():
    """`use_graph_replay` / `Rasterize.graph_replay` / `Renderer.graph_replay`: the operator replayed from captured HIP graphs
    returns what the eager operator returns -- images bit for bit, gradients up to K6's summation order -- on changing
    inputs of fixed shapes, for every output combination, with outputs that stay valid after later calls; a backward after
    a newer forward of the same shapes raises instead of using the wrong residual maps."""
    import neural_renderer_amd as nr
    rng = np.random.default_rng(404)
    S, B = 64, 3
    faces_np = [H.teapot_views(B, S)[0][:, :600].copy() for _ in range(1)][0]
    tex_np = rng.uniform(0, 1, (B, faces_np.shape[1], 2, 2, 2, 3)).astype(np.float32)
    for modes in ((True, True, True), (False, True, False), (False, False, True), (True, False, False)):
        kept = []
        for it in range(3):
            f_np = faces_np + rng.normal(scale=0.01, size=faces_np.shape).astype(np.float32) * np.array([1, 1, 0], np.float32)
            g = [rng.normal(size=(B, S, S, 3)).astype(np.float32), rng.normal(size=(B, S, S)).astype(np.float32),
                 rng.normal(size=(B, S, S)).astype(np.float32)]
            res = []
            for replay in (False, True):
                ft = torch.tensor(f_np, device='cuda', requires_grad=True)
                tt = torch.tensor(tex_np, device='cuda', requires_grad=True)
                fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), *modes)
                fn.graph_replay = replay
                outs = fn(ft, tt)
                sel = [(o, torch.tensor(gg, device='cuda')) for o, gg in zip(outs, g) if o is not None]
                torch.autograd.backward([o for o, _ in sel], [gg for _, gg in sel])
                res.append(([None if o is None else o.detach().cpu().numpy() for o in outs], ft.grad.cpu().numpy(),
                            None if tt.grad is None else tt.grad.cpu().numpy(), outs))
            (o_e, gf_e, gt_e, _), (o_r, gf_r, gt_r, outs_r) = res
            for a, b in zip(o_e, o_r):
                assert (a is None) == (b is None)
                if a is not None:
                    np.testing.assert_array_equal(a, b)
            if modes[0] or modes[1]:
                assert H.rel_err(gf_r, gf_e) <= SAME_TERMS
            else:
                np.testing.assert_array_equal(gf_r, gf_e)
            if modes[0]:
                np.testing.assert_array_equal(gt_r, gt_e)
            kept.append((outs_r, o_r))
        for outs_r, o_r in kept:  # results of earlier replays are copies: later calls did not change them
            for t, ref in zip(outs_r, o_r):
                if t is not None:
                    np.testing.assert_array_equal(t.detach().cpu().numpy(), ref)
    # two forwards in flight: the first one's backward must refuse
    fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), False, True, False)
    fn.graph_replay = True
    f1 = torch.tensor(faces_np, device='cuda', requires_grad=True)
    a1 = fn(f1)[1]
    a2 = fn(torch.tensor(faces_np, device='cuda', requires_grad=True))[1]
    with pytest.raises(RuntimeError):
        a1.sum().backward()
    a2.sum().backward()
    # and through the Renderer (example 2's call), module-level switch
    v, f = H.teapot()
    r = nr.Renderer()
    r.image_size, r.anti_aliasing = 64, False
    imgs = []
    for replay in (False, True):
        r.graph_replay = replay
        vt = torch.tensor(v[None], device='cuda', requires_grad=True)
        img = r.render_silhouettes(vt, torch.tensor(f[None], device='cuda'))
        img.sum().backward()
        imgs.append((img.detach().cpu().numpy(), vt.grad.cpu().numpy()))
    np.testing.assert_array_equal(imgs[0][0], imgs[1][0])
    assert H.rel_err(imgs[1][1], imgs[0][1]) <= 1e-5


def test_function_protocol_equals_the_autograd_operator():
    """`fn.forward_gpu(inputs)` / `fn.backward_gpu(inputs, grad_outputs)` -- the chainer.Function protocol of the reference
    (rasterize.py:467, :849), no autograd graph -- return what the differentiable call returns: maps bit for bit, gradients
    up to K6's summation order; `None` gradients are zeros; the buffers stay on the instance (:39-58)."""
    import neural_renderer_amd as nr
    rng = np.random.default_rng(77)
    S, B = 96, 3
    faces_np, _ = H.teapot_views(B, S)
    tex_np = rng.uniform(0, 1, (B, faces_np.shape[1], 2, 2, 2, 3)).astype(np.float32)
    g = [torch.tensor(rng.normal(size=(B, S, S, 3)).astype(np.float32), device='cuda'),
         torch.tensor(rng.normal(size=(B, S, S)).astype(np.float32), device='cuda'),
         torch.tensor(rng.normal(size=(B, S, S)).astype(np.float32), device='cuda')]
    for modes in ((True, True, True), (False, True, False), (False, False, True), (True, False, False)):
        ft = torch.tensor(faces_np, device='cuda', requires_grad=True)
        tt = torch.tensor(tex_np, device='cuda', requires_grad=True)
        op = nr.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), *modes)
        outs = op(ft, tt)
        sel = [(o, gg) for o, gg in zip(outs, g) if o is not None]
        torch.autograd.backward([o for o, _ in sel], [gg for _, gg in sel])
        fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0.1, 0.2, 0.3), *modes)
        inputs = (ft.detach(), tt.detach()) if modes[0] else (ft.detach(),)
        outs2 = fn.forward_gpu(inputs)
        for a, b2 in zip(outs, outs2):
            assert (a is None) == (b2 is None)
            if a is not None:
                np.testing.assert_array_equal(a.detach().cpu().numpy(), b2.cpu().numpy())
                assert not b2.requires_grad
        np.testing.assert_array_equal(fn.face_index_map.cpu().numpy(), op.face_index_map.cpu().numpy())
        grads = fn.backward_gpu(inputs, tuple(gg if o is not None else None for o, gg in zip(outs2, g)))
        assert len(grads) == (2 if modes[0] else 1)
        tol = SAME_TERMS if (modes[0] or modes[1]) else 0.0
        assert H.rel_err(grads[0].cpu().numpy(), ft.grad.cpu().numpy()) <= tol
        assert fn.grad_faces is not None and fn.faces is not None and fn.batch_size == B and fn.num_faces == faces_np.shape[1]
        if modes[0]:
            np.testing.assert_array_equal(grads[1].cpu().numpy(), tt.grad.cpu().numpy())
        # no gradient at all: zeros, like rasterize.py:851-853
        zero = fn.backward_gpu(inputs, (None, None, None))
        assert all(float(z.abs().max()) == 0.0 for z in zero)
    with pytest.raises(RuntimeError):
        nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), False, True, False).backward_gpu((ft.detach(),), (None, g[1], None))


def test_lazy_residual_maps_on_the_instance():
    """The reference keeps face_inv_map and the two sampling maps on the Function (rasterize.py:47-48, :57); here the kernels
    recompute them, and reading the attribute runs the per-stage entry point with the optional pointers: same values as the
    oracle's maps."""
    import neural_renderer_amd as nr
    rng = np.random.default_rng(78)
    S, B = 64, 2
    faces_np, _ = H.teapot_views(B, S)
    tex_np = rng.uniform(0, 1, (B, faces_np.shape[1], 3, 3, 3, 3)).astype(np.float32)
    ref = oracle_forward(faces_np, tex_np, S, 0.1, 100, 1e-3, (0.2, 0.4, 0.6), True, True, True)
    fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0.2, 0.4, 0.6), True, True, True)
    assert fn.face_inv_map is None and fn.sampling_index_map is None  # nothing rendered yet
    fn(torch.tensor(faces_np, device='cuda'), torch.tensor(tex_np, device='cuda'))
    # weight_map (rasterize.py:43): the operator's forward stores covered pixels only (NR_FLAG_SPARSE_WEIGHT_MAP); the attribute
    # fills in the zeros
    np.testing.assert_array_equal(fn.weight_map.cpu().numpy(), ref.weight_map)
    assert fn.weight_map is fn.weight_map
    # the flag at the C ABI: covered pixels as without it, the others untouched (the test driver pre-fills with NaN)
    fw = abi.forward_fused(faces_np, tex_np, S, 0.1, 100.0, 1e-3, (0.2, 0.4, 0.6), 32, True, True, True)
    w = abi.host(fw['weight_map'])
    cov = ref.face_index_map >= 0
    np.testing.assert_array_equal(w[cov], ref.weight_map[cov])
    assert np.isnan(w[~cov]).all() and cov.any() and (~cov).any()
    for k in ('depth_map', 'rgb_map', 'alpha_map'):
        np.testing.assert_array_equal(abi.host(fw[k]), getattr(ref, k))
    # (the stage kernels write every element, init values included: whole maps compare)
    np.testing.assert_array_equal(fn.face_inv_map.cpu().numpy(), ref.face_inv_map)
    np.testing.assert_array_equal(fn.sampling_weight_map.cpu().numpy(), ref.sampling_weight_map)
    np.testing.assert_array_equal(fn.sampling_index_map.cpu().numpy(), ref.sampling_index_map)
    assert fn.sampling_weight_map is fn.sampling_weight_map  # computed once per call
    fn2 = nr.Rasterize(S, 0.1, 100, 1e-4, None, False, True, False)
    fn2(torch.tensor(faces_np, device='cuda'))
    assert fn2.sampling_index_map is None and fn2.face_inv_map is not None


def test_workspace_cache_is_bounded_and_can_be_cleared():
    """The kept forward workspaces (8 B per raster pixel): least recently used first out, a byte cap, clear_workspace_cache()."""
    import neural_renderer_amd as nr
    import sys
    R = sys.modules['neural_renderer_amd.rasterize']  # (the package attribute `rasterize` is the function)
    nr.clear_workspace_cache()
    faces = torch.tensor(H.random_scene(np.random.default_rng(5), 1, 8), device='cuda')
    for S in range(8, 8 + 20):
        nr.Rasterize(S, 0.1, 100, 1e-3, None, False, True, False)(faces)
    assert 0 < len(R._ZBUF_CACHE) <= 16
    first = next(iter(R._ZBUF_CACHE))
    nr.Rasterize(first[4], 0.1, 100, 1e-3, None, False, True, False)(faces)  # a hit moves the entry to the young end
    assert next(reversed(R._ZBUF_CACHE)) == first
    cap = R._ZBUF_CACHE_BYTES
    try:
        R._ZBUF_CACHE_BYTES = 1024  # nothing fits: every call takes the per-call fill, results unchanged
        nr.clear_workspace_cache()
        a = nr.Rasterize(64, 0.1, 100, 1e-3, None, False, True, False)(faces)[1]
        assert len(R._ZBUF_CACHE) == 0
    finally:
        R._ZBUF_CACHE_BYTES = cap
    b = nr.Rasterize(64, 0.1, 100, 1e-3, None, False, True, False)(faces)[1]
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    nr.clear_workspace_cache()
    assert len(R._ZBUF_CACHE) == 0


_CAPTURE_AFTER_REPLAY = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
import neural_renderer_amd as nr
R = sys.modules['neural_renderer_amd.rasterize']
dev = torch.device('cuda', 0)
faces, textures = bench.build_scene(dev, 16, 0, 16, 256, 2)   # BASELINE config 2: the scene round 3 crashed on
faces = faces.clone().requires_grad_(True)
textures = textures.clone().requires_grad_(True)
with torch.no_grad():
    outs = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
    grads = [torch.rand_like(o) for o in outs]
def step():
    faces.grad = None
    textures.grad = None
    o = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
    torch.autograd.backward(list(o), grads)
step()
eager = faces.grad.clone()
nr.use_graph_replay(True)
for _ in range(5):
    step()                    # the operator's replay mode captures its graphs and replays them
nr.use_graph_replay(False)
torch.cuda.synchronize()
assert len(R._GRAPH_CACHE) == 1
replay = nr.graph.capture(step, dev)   # round 3: SIGSEGV in here
for _ in range(3):
    replay()
torch.cuda.synchronize()
err = float((faces.grad - eager).abs().max() / eager.abs().max())
print('CAPTURED', err)
assert err <= 1e-5
R.clear_graph_replay_cache()
print('CLEARED', len(R._GRAPH_CACHE))
'''


def test_whole_step_capture_after_operator_replay():
    """Round 3 (ROCm 7.2 / torch 2.10): capturing a whole step after the operator's graph-replay mode had run in the same process
    crashed inside torch's capture (SIGSEGV in run_backward; scripts/graph_crash_probe.py reproduces it with round 3's host
    code and either library).  With the rewritten operator the sequence works; this runs it in a subprocess, so that a crash
    fails the test and not the session, and checks the replayed step against the eager one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, '-c', _CAPTURE_AFTER_REPLAY % root], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-1500:])
    assert 'CAPTURED' in res.stdout and 'CLEARED 0' in res.stdout


def test_operator_does_not_leak_device_memory():
    """The autograd node must not hold its own outputs outside save_for_backward (output -> grad_fn -> node -> output is a cycle
    through C++ that nothing collects): after warm-up, steps of the differentiable call leave the allocated bytes unchanged."""
    import gc
    import neural_renderer_amd as nr
    faces = torch.tensor(H.teapot_views(2, 64)[0], device='cuda', requires_grad=True)
    tex = torch.ones((2, faces.shape[1], 2, 2, 2, 3), device='cuda', requires_grad=True)

    def step():
        faces.grad = tex.grad = None
        fn = nr.Rasterize(64, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
        rgb, alpha, depth = fn(faces, tex)
        (rgb.sum() + alpha.sum() + depth.sum()).backward()

    for _ in range(5):
        step()
    gc.collect()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    for _ in range(20):
        step()
    gc.collect()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= before + (1 << 16), (before, torch.cuda.memory_allocated())


def test_band_kernel_timing_hook():
    """nr_profile_band_kernel (include/nr_hip_profile.h) lives in the MEASUREMENT build of the library only (libnr_hip_prof.so,
    what bench.py's roofline loads): off -> no reading; on -> the band kernel's launch of the last K6 call is bracketed by the
    library's own events, staged and fused entry points alike, and the results are those of the product library."""
    from neural_renderer_amd import _lib
    assert not hasattr(_lib.load(), 'nr_profile_band_kernel')  # (not in the product ABI)
    lib = _lib.load_profile()
    faces, _ = H.teapot_views(2, 64)
    rng = np.random.default_rng(77)
    textures = rng.uniform(0, 1, (2, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    fw = abi.forward(faces, textures, 64, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, True, True, False)
    g_rgb = rng.normal(size=(2, 64, 64, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(2, 64, 64)).astype(np.float32)
    ref, _ = abi.backward_fused(fw, g_rgb, g_alpha, None, k6_flags=EXACT)
    assert lib.nr_profile_band_kernel(0) == 0 and lib.nr_profile_band_kernel_ms() < 0
    prev = _lib._lib
    _lib._lib = lib  # (tests/abi.py calls through _lib.load(): the measurement build for the calls below)
    try:
        assert lib.nr_profile_band_kernel(1) == 0
        assert lib.nr_profile_band_kernel_ms() < 0  # nothing bracketed yet
        got, _ = abi.backward_fused(fw, g_rgb, g_alpha, None, k6_flags=EXACT)
        t_fused = lib.nr_profile_band_kernel_ms()
        abi.backward(fw, g_rgb, g_alpha, None, k6_flags=EXACT)
        t_staged = lib.nr_profile_band_kernel_ms()
        assert lib.nr_profile_band_kernel_which() == 1  # (the exact mode: k_bpm_row as well)
        abi.backward(fw, g_rgb, g_alpha, None)  # (the default mode: k_bpm_row is bracketed as well, and named)
        t_px = lib.nr_profile_band_kernel_ms()
        assert lib.nr_profile_band_kernel_which() == 1
        abi.backward(fw, g_rgb, g_alpha, None, k6_flags=K6_LEGACY)
        assert lib.nr_profile_band_kernel_ms() > 0 and lib.nr_profile_band_kernel_which() == 0
    finally:
        lib.nr_profile_band_kernel(0)
        _lib._lib = prev
    assert 0 < t_fused < 50 and 0 < t_staged < 50 and 0 < t_px < 50  # milliseconds; a 2-view 64 x 64 launch takes tens of microseconds
    np.testing.assert_array_equal(abi.host(got), abi.host(ref))  # (exact mode: the same bits run to run, and in both builds)
    assert lib.nr_profile_band_kernel_ms() < 0
