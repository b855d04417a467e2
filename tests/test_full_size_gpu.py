"""`-m gpu`: parity at the FULL size of the BASELINE.json configurations -- every element of every output against the
C oracle, which for these sizes runs its cache-blocked K2 loop order on all host cores (bit-identical to the literal
loop: tests/test_oracle_threads.py).

  H   the headline batch: teapot, 64 azimuth views, 256x256, rgb + alpha + depth, forward + backward (the configuration
      bench.py times, in the mode it times it)
  C4  per-GPU share of config 4: 64 distinct random meshes x 10 240 faces, texture_size 4, 256x256 textured RGB
  C5  config 5: one 655 360-face mesh, 1024x1024, texture_size 8, rgb + alpha + depth

The oracle's forward is O(pixels x faces) by construction (H: 2.1e10, C4: 4.3e10, C5: 6.9e11 face tests); on the GPU box's
256 host cores that is seconds.  A host with few cores skips C5 (NR_FULL_SIZE_MIN_CORES, default 32)."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import oracle as O
import abi
import helpers as H
from test_hip_parity import icosphere, project_mesh, report, K6_BOUND_DEFAULT, K6_BOUND_EXACT, EXACT, RTOL

pytestmark = pytest.mark.gpu


def _compare(faces, textures, S, eps, modes, seed, bg=(0.0, 0.0, 0.0), double_textures=True):
    rgb, alpha, depth = modes
    t0 = time.time()
    fn = O.Rasterize(S, 0.1, 100, eps, bg, rgb, alpha, depth)
    fn.blocked = True
    fn(faces, textures) if rgb else fn(faces)
    t_fwd = time.time() - t0
    fw = abi.forward_fused(faces, textures, S, 0.1, 100.0, eps, bg, 0, rgb, alpha, depth)
    fi = abi.host(fw['face_index_map'])
    mism = int((fi != fn.face_index_map).sum())
    assert mism == 0, 'face_index_map: %d mismatches' % mism
    for name in ('weight_map', 'depth_map', 'rgb_map', 'alpha_map'):
        ref, got = getattr(fn, name, None), fw.get(name)
        if ref is not None and got is not None:
            np.testing.assert_array_equal(abi.host(got), ref, err_msg=name)
    rng = np.random.default_rng(seed)
    shape = fn.face_index_map.shape
    g_rgb = rng.normal(size=shape + (3,)).astype(np.float32) if rgb else None
    g_alpha = rng.normal(size=shape).astype(np.float32) if alpha else None
    g_depth = rng.normal(size=shape).astype(np.float32) if depth else None
    t0 = time.time()
    if double_textures:
        ref = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True)
        ref_gf, ref_gt = ref[0], (ref[1] if rgb else None)
    else:
        # grad_textures of a 655 360 x 8^3 x 3 tensor: float sums (4 GB) instead of a double twin (8 GB); K6 / K8 in double
        ref_gt = fn.backward(g_rgb, g_alpha, g_depth)[1].copy() if rgb else None
        keep = fn.return_rgb
        ref_gf = fn.backward(g_rgb, g_alpha, g_depth, accumulate_double=True, skip_textures=True)[0]
    t_bwd = time.time() - t0
    out = {}
    # (default mode: k_bpm_row, then the same terms on k_bpm_fast -- NR_FLAG_K6_LEGACY 128 --, then the exact mode)
    for flags, bound in ((0, K6_BOUND_DEFAULT), (128, K6_BOUND_DEFAULT), (EXACT, K6_BOUND_EXACT)):
        gf, gt = abi.backward_fused(fw, g_rgb, g_alpha, g_depth, k6_flags=flags)
        gf = abi.host(gf)
        assert np.isfinite(gf).all()
        err = H.rel_err(gf, ref_gf)
        ok = np.abs(ref_gf) > 0
        out[flags] = dict(err=err, max_abs_err=float(np.abs(gf - ref_gf).max()), max_abs=float(np.abs(ref_gf).max()),
                          frac_within_1e4=float(np.mean(np.abs(gf[ok] - ref_gf[ok]) <= RTOL * np.abs(ref_gf[ok]))))
        assert err <= (max(bound, 1e-5) if depth else bound), (flags, err)
        if rgb:
            gt = abi.host(gt)
            e_t = H.rel_err(gt, ref_gt)
            out[flags]['grad_textures_err'] = e_t
            assert e_t <= RTOL, e_t
            del gt
    report('full_size', S=S, B=int(faces.shape[0]), F=int(faces.shape[1]), modes=list(modes), oracle_fwd_s=t_fwd,
           oracle_bwd_s=t_bwd, threads=O.get_threads(), covered=int((fi >= 0).sum()), visits=fn.visits,
           default=out[0], k_bpm_fast=out[128], exact=out[EXACT])
    return out


def test_headline_64_views_all_outputs():
    """The benchmarked configuration in its benchmarked mode: 64 teapot views, 256x256, texture_size 2, eps 1e-3, rgb +
    alpha + depth; non-uniform textures so that K4 / K7 are observable."""
    faces, _ = H.teapot_views(64, 256)
    rng = np.random.default_rng(640)
    textures = rng.uniform(0, 1, (64, faces.shape[1], 2, 2, 2, 3)).astype(np.float32)
    _compare(faces, textures, 256, 1e-3, (True, True, True), seed=641, bg=(0.1, 0.2, 0.3))


def test_headline_batch_equals_its_shards():
    """The multi-GPU design rests on shards == batch: the benchmarked 64 teapot views at 256 x 256 as ONE call, as
    2 x 32 and as 8 x 8 views (every shard handed the global first view as `faces_z_ref`).  Images and
    grad_textures: the same bits.  grad_faces: every call size takes the same band kernel (k_bpm_row: a record's sums do not
    depend on what else is in the launch), so what can differ is the order of the double atomics that add a face's records
    up -- bound 1e-6 in the parity metric, two orders below the default mode's distance from the oracle (measured: reported to
    gpurun_out/parity_errors.jsonl).  Also with NR_FLAG_K6_LEGACY (k_bpm_fast: float run sums grouped by arrival: 3e-5)."""
    from test_hip_parity import report
    B, S, TS = 64, 256, 2
    faces, _ = H.teapot_views(B, S)
    rng = np.random.default_rng(6464)
    textures = rng.uniform(0, 1, (B, faces.shape[1], TS, TS, TS, 3)).astype(np.float32)
    g_rgb = rng.normal(size=(B, S, S, 3)).astype(np.float32)
    g_alpha = rng.normal(size=(B, S, S)).astype(np.float32)

    def run(sl, k6_flags):
        fw = abi.forward_fused(faces[sl], textures[sl], S, 0.1, 100.0, 1e-3, (0.1, 0.2, 0.3), 0, True, True, False,
                               faces_z_ref=faces[0])
        gf, gt = abi.backward_fused(fw, g_rgb[sl], g_alpha[sl], None, k6_flags=k6_flags)
        return abi.host(fw['rgb_map']), abi.host(fw['alpha_map']), abi.host(gf), abi.host(gt)

    levels = {}
    for k6_flags, bound in ((0, 1e-6), (128, 3e-5)):
        full = run(slice(0, B), k6_flags)
        for n in (2, 8):
            step = B // n
            parts = [run(slice(i * step, (i + 1) * step), k6_flags) for i in range(n)]
            for k, name in enumerate(('rgb_map', 'alpha_map', 'grad_faces', 'grad_textures')):
                got = np.concatenate([p[k] for p in parts])
                if name == 'grad_faces':
                    e = H.rel_err(got, full[k])
                    levels['flags_%d_shards_%d' % (k6_flags, n)] = dict(rel=e, bit_equal=bool(np.array_equal(got, full[k])))
                    assert e <= bound, (k6_flags, n, e)
                else:
                    np.testing.assert_array_equal(got, full[k], err_msg='%s, %d shards' % (name, n))
    report('headline_batch_equals_its_shards', **levels)


def config4_meshes(batch, seed=1234):
    """icosphere (5 120 faces) with per-vertex radial noise around radius 0.5, per-mesh random rotation."""
    rng = np.random.default_rng(seed)
    v0, f0 = icosphere(4)
    out = []
    for _ in range(batch):
        v = v0 * (0.5 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        out.append(project_mesh((v @ q).astype(np.float32), f0, [0.3, 0.4, -2.6]))
    return np.stack(out)


def test_config4_64_meshes_full_size():
    faces = config4_meshes(64)
    assert faces.shape[1] == 10240
    rng = np.random.default_rng(44)
    textures = rng.uniform(0, 1, (64, faces.shape[1], 4, 4, 4, 3)).astype(np.float32)
    _compare(faces, textures, 256, 1e-3, (True, False, False), seed=45)


@pytest.mark.skipif((os.cpu_count() or 1) < int(os.environ.get('NR_FULL_SIZE_MIN_CORES', '32')),
                    reason='the O(pixels x faces) oracle needs many host cores for 1024^2 x 655 360 faces')
def test_config5_655k_faces_1024_full_size():
    rng = np.random.default_rng(55)
    v0, f0 = icosphere(7)  # 327 680 faces -> 655 360 with fill_back
    v = v0 * (0.6 + 0.02 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
    faces = project_mesh(v.astype(np.float32), f0, [0.0, 0.0, -2.4])[None]
    assert faces.shape[1] == 655360
    textures = rng.uniform(0, 1, (1, faces.shape[1], 8, 8, 8, 3)).astype(np.float32)
    _compare(faces, textures, 1024, 1e-3, (True, True, True), seed=56, double_textures=False)
    torch.cuda.empty_cache()
