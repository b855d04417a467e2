#!/usr/bin/env python3
"""bench.py -- rasterize fwd+bwd Mpixels/s on teapot.obj, 256x256, batch 64 (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the rasterizer hot path (Rasterize forward + backward through the product's autograd
operator, i.e. the C-ABI entry points nr_forward_rasterize / nr_backward_rasterize) over one batch of 64 views of the
teapot at raster size 256 with RGB + alpha + depth outputs all enabled.  Inputs (projected faces, lit textures, upstream
gradients) are resident in HBM before the timed region.  With --gpus R the 64-view batch is SPLIT: rank r renders views
[r 64/R, (r+1) 64/R) of the same 64 azimuths (BASELINE.md "Multi-GPU rows"; neural_renderer_amd.distributed.
shard_bounds), with no collective inside forward + backward -- "strong" scaling, `config.views_per_gpu` = 64/R.  The
64-views-PER-GPU job of rounds 1-3 (R x 64 views, "weak" scaling) is timed in the same run and reported as the
`weak_scaling` object of the line.  Rank 0 prints ONE JSON line.

Objects on the line besides the contract's keys:
  roofline      the dominant stage's algorithmic HBM bytes / its measured average launch duration (HIP events on the launch
                stream), against the 8 TB/s HBM3E peak; `whole_step` prices the whole fwd+bwd with SURVEY 8d's compulsory
                bytes (92 P + (108 + 24 ts^3) N for rgb+alpha+depth); `valu_issue` is the secondary bound of the K6 kernel,
                whose floor is instruction issue, not HBM
  cpu_baseline  the C oracle (oracle/nr_oracle.c, a literal port of the reference's algorithm) on this host: one thread on
                a bounded sample (`value`), all cores on the whole batch (`all_cores`), and the north star's naive NumPy
                per-pixel loop on BASELINE configs[0] (`numpy_naive_config1`)
  grad_check    parity of the benchmarked batch against the oracle: face_index_map mismatches (must be 0), max-abs and
                max-rel errors of the gradients, all 64 views
  extra_rows    the same step with anti-aliasing on (raster 512, the Renderer default) and with the all-ones upstream
                gradient of the reference's misc/measure_time.py:60
  shard_rows    (1-GPU run) the step at the per-GPU shard sizes of a 2 / 4 / 8-GPU run -- 32 / 16 / 8 views -- on this one GPU:
                through the autograd operator (torch defaults), with the backward kept on the calling thread, and through the
                operator's chainer.Function protocol (forward_gpu / backward_gpu: no autograd graph); `predicted_strong_scaling`
                is what these times mean for the 64-view job on R GPUs (no collective in the path)
  weak_scaling  (N > 1) 64 views per GPU instead of 64 / N: the job of the earlier rounds' N > 1 points
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
NUM_SIMDS = 1024       # 256 CUs x 4 SIMDs
NS_PER_WAVE_INSTR = 1.07  # FP32 VALU wave-instruction issue interval per SIMD, measured by microbenchmark


def load_teapot():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_fixtures.npz'))
    v = g['teapot_vertices_raw'].astype(np.float32).copy()
    # reference load_obj.py:188-192 normalisation
    v -= v.min(0)[None, :]
    v /= np.abs(v).max()
    v *= 2
    v -= v.max(0)[None, :] / 2
    return v, g['teapot_faces'].astype(np.int32)


def build_scene(device, batch, first_view, total_views, image_size, texture_size):
    """Projected faces [B,F,3,3] and lit textures [B,F,ts,ts,ts,3] for views first_view.. of total_views
    azimuths (elevation 30, distance 2.732, examples/example1.py:26-27) -- the Renderer.render front-end
    (renderer.py:75-103) executed once, outside the timed region."""
    import neural_renderer_amd as nr
    v, f = load_teapot()
    vertices = torch.from_numpy(v).to(device)[None].repeat(batch, 1, 1)
    faces_i = torch.from_numpy(f).to(device)[None].repeat(batch, 1, 1)
    faces_i = torch.cat((faces_i, torch.flip(faces_i, dims=[2])), dim=1)  # fill_back
    textures = torch.ones((batch, f.shape[0], texture_size, texture_size, texture_size, 3), device=device)
    textures = torch.cat((textures, textures.permute(0, 1, 4, 3, 2, 5)), dim=1)
    textures = nr.lighting(nr.vertices_to_faces(vertices, faces_i), textures, 0.5, 0.5, (1, 1, 1), (1, 1, 1), (0, 1, 0))
    eyes = [nr.get_points_from_angles(2.732, 30., 360.0 * (first_view + i) / total_views) for i in range(batch)]
    eye = torch.tensor(eyes, dtype=torch.float32, device=device)
    vertices = nr.perspective(nr.look_at(vertices, eye), 30.)
    faces = nr.vertices_to_faces(vertices, faces_i)
    return faces.contiguous(), textures.contiguous()


def icosphere(level):
    """Subdivided icosahedron on the unit sphere: vertices [Nv,3] float32, faces [Nf,3] int32 (20 * 4^level faces)."""
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.int32)


C4_MESHES = 512  # BASELINE.json configs[3]: 512 random meshes of ~5k faces, sharded over the GPUs


def build_scene_c4(device, first, count, texture_size=4):
    """BASELINE.json configs[3] / SURVEY 8d C4, meshes [first, first + count) of the 512: an icosphere of 5 120 faces with
    seeded per-vertex radial noise around radius 0.55 and a random rotation per mesh, seen from eye (0.3, 0.4, -2.6) through the
    product's look_at / perspective(30) / vertices_to_faces glue with fill_back (10 240 faces per mesh), texture_size 4 textures
    ~ U(0, 1).  Mesh m depends on (1234, m) only: every rank builds exactly its shard."""
    import neural_renderer_amd as nr
    v0, f0 = icosphere(4)
    verts, tex = [], []
    for m in range(first, first + count):
        rng = np.random.default_rng([1234, m])
        v = v0 * (0.55 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        verts.append((v @ q).astype(np.float32))
        tex.append(rng.uniform(0, 1, (2 * f0.shape[0], texture_size, texture_size, texture_size, 3)).astype(np.float32))
    vertices = torch.from_numpy(np.stack(verts)).to(device)
    faces_i = torch.from_numpy(f0).to(device)[None].repeat(count, 1, 1)
    faces_i = torch.cat((faces_i, torch.flip(faces_i, dims=[2])), dim=1)  # fill_back
    eye = torch.tensor([[0.3, 0.4, -2.6]], dtype=torch.float32, device=device).repeat(count, 1)
    vertices = nr.perspective(nr.look_at(vertices, eye), 30.)
    faces = nr.vertices_to_faces(vertices, faces_i)
    return faces.contiguous(), torch.from_numpy(np.stack(tex)).to(device).contiguous()


def upstream_gradients(faces, textures, S, eps, seed, all_ones=False, z_ref=None, modes=(True, True, True)):
    """g = 2 (image - ref) with a seeded uniform reference (dense, both signs); all_ones: the gradient of
    sum(images) as in the reference's misc/measure_time.py:60 (K6's `diff_grad <= 0` branch then skips half the work)."""
    import neural_renderer_amd as nr
    dev = faces.device
    gen = torch.Generator(device='cpu').manual_seed(seed)
    with torch.no_grad():
        fn = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), *modes)
        fn.faces_z_ref = z_ref
        rgb0, alpha0, depth0 = fn(faces, textures) if modes[0] else fn(faces)
        if all_ones:
            return tuple(None if o is None else torch.ones_like(o) for o in (rgb0, alpha0, depth0))
        # (a disabled output has no gradient: None, like the reference's grad_outputs, rasterize.py:858-878)
        g_rgb = None if rgb0 is None else (2 * (rgb0 - torch.rand(rgb0.shape, generator=gen).to(dev))).contiguous()
        g_alpha = None if alpha0 is None else (2 * (alpha0 - torch.rand(alpha0.shape, generator=gen).to(dev))).contiguous()
        g_depth = None if depth0 is None else (2 * (depth0 / 100.0 - torch.rand(depth0.shape, generator=gen).to(dev)) / 100.0).contiguous()
    return g_rgb, g_alpha, g_depth


# dominant kernel of each stage call (the rest of a stage are small helper launches, see profiles/README.md)
STAGE_KERNEL = {
    'forward_face_index_map': 'k_face_raster', 'forward_texture_sampling': 'k_shade', 'backward_pixel_map': 'k_bpm_row',
    'backward_textures': 'k_backward_textures_face', 'backward_depth_map': 'k_backward_depth_face',
}


def k6_band_kernel(B, F, S, rgb, alpha, exact):
    """Which band kernel the library launched for the rgb + alpha stage call of this shape that time_stages() timed through the
    measurement build (nr_profile_band_kernel_which: asked, not re-derived -- k_bpm_row in the default mode wherever its band
    fits, k_bpm_fast in the exact mode; csrc/nr_backward_pixel_map.hip k6_row_band).  None without a measurement of that
    shape (no measurement build in the tree): the roofline then names no kernel and carries no per-kernel traffic."""
    return time_stages.band_kernel.get((B, F, S, bool(exact)))


def algorithmic_bytes(B, F, S, ts):
    """Compulsory HBM traffic per launch of each stage (every input read once, every output written once;
    recomputable intermediates count zero) -- DESIGN.md 'Kernels'."""
    P, N = B * S * S, B * F
    return {
        'forward_face_index_map': N * 36 + P * (4 + 12 + 4),
        'forward_texture_sampling': P * (4 + 12 + 4) + N * 12 * ts ** 3 + P * (12 + 4),
        'backward_pixel_map': P * (4 + 12 + 4 + 12 + 4) + N * 72,
        'backward_textures': P * (4 + 12 + 4 + 12) + N * 24 * ts ** 3,
        'backward_depth_map': P * (4 + 12 + 4 + 4) + N * 72,
    }


def coverage_scaled_bytes(B, F, S, ts, covered, visible):
    """The same figures with the terms a kernel rightly skips scaled down: weight / depth / gradient reads happen only on
    the `covered` fraction of the pixels (12 % of a teapot view), texture reads only for the `visible` fraction of the faces.
    A per-stage fraction of the HBM peak must be priced with THESE bytes (the unscaled figure put k_shade at 0.96 of the
    8 TB/s spec peak, above what the part can deliver); K6 stages whole image bands and the visibility pass writes every
    element, so their figures do not change."""
    P, N = B * S * S, B * F
    c, v = covered, visible
    return {
        'forward_face_index_map': N * 36 + P * (4 + 12 + 4),
        'forward_texture_sampling': P * 4 + c * P * (12 + 4) + v * N * 12 * ts ** 3 + P * (12 + 4),
        'backward_pixel_map': P * (4 + 12 + 4 + 12 + 4) + N * 72,
        'backward_textures': P * 4 + c * P * (12 + 4 + 12) + N * 24 * ts ** 3,
        'backward_depth_map': P * 4 + c * P * (12 + 4 + 4) + N * 72,
    }


def whole_step_bytes(B, F, S, ts):
    """SURVEY 8d 'Algorithmic bytes' for one Rasterize fwd+bwd with rgb + alpha + depth: forward writes rgb 12 + alpha 4 +
    depth 4 + face_index 4 + weight 12, backward reads the three gradients 20, rgb 12, alpha 4 and the residuals 20:
    92 B per pixel; per face 36 (faces, forward) + 36 (faces, backward) + 36 (grad_faces) + 24 ts^3 (textures read,
    grad_textures written)."""
    return 92 * B * S * S + (108 + 24 * ts ** 3) * B * F


def whole_step_bytes_rgb(B, F, S, ts):
    """SURVEY 8d, rgb only (config 4): 76 B per pixel (forward writes rgb 12 + face_index 4 + weight 12 + depth 4, backward reads
    g_rgb 12 + rgb 12 + the residuals 20) + (108 + 24 ts^3) B per face."""
    return 76 * B * S * S + (108 + 24 * ts ** 3) * B * F


def time_stages(faces, textures, S, eps, g_rgb, g_alpha, g_depth, iters, k6_flags=0, only=None):
    """Average duration of each C-ABI stage, measured with events on the stream the kernels are launched on.  `only`: a subset of
    the stage names (the stages are run in the order below: later ones read what earlier ones wrote)."""
    from neural_renderer_amd import _lib
    lib = _lib.load()
    dev = faces.device
    B, F = faces.shape[:2]
    ts = textures.shape[2]
    st = torch.cuda.current_stream(dev).cuda_stream
    fi = torch.empty((B, S, S), dtype=torch.int32, device=dev)
    wm = torch.empty((B, S, S, 3), device=dev)
    dm = torch.empty((B, S, S), device=dev)
    rgb = torch.empty((B, S, S, 3), device=dev)
    am = torch.empty((B, S, S), device=dev)
    vis = torch.empty((B, F), dtype=torch.uint8, device=dev)
    bg = torch.zeros(3, device=dev)
    gf = torch.empty_like(faces)
    gt = torch.empty_like(textures)
    wsb = lib.nr_forward_workspace_bytes(B, F, S)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bwsb = lib.nr_backward_workspace_bytes(B, F, S, 1, 1)
    bws = torch.empty(max(bwsb, 1), dtype=torch.uint8, device=dev)

    calls = {
        'forward_face_index_map': lambda: lib.nr_forward_face_index_map(
            faces.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), None, vis.data_ptr(), B, F, S, 0.1, 100.0,
            ws.data_ptr(), wsb, st),
        'forward_texture_sampling': lambda: lib.nr_forward_texture_sampling(
            faces.data_ptr(), None, textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(), None,
            None, bg.data_ptr(), 0, am.data_ptr(), B, F, S, ts, eps, 0, st),
        'backward_pixel_map': lambda: lib.nr_backward_pixel_map(
            faces.data_ptr(), fi.data_ptr(), rgb.data_ptr(), am.data_ptr(), g_rgb.data_ptr(), g_alpha.data_ptr(),
            gf.data_ptr(), B, F, S, eps, 1, 1, k6_flags, vis.data_ptr(), bws.data_ptr(), bwsb, st),
        'backward_textures': lambda: lib.nr_backward_textures(
            fi.data_ptr(), None, None, faces.data_ptr(), None, wm.data_ptr(), dm.data_ptr(), g_rgb.data_ptr(),
            gt.data_ptr(), B, F, S, ts, eps, 0, st),
        'backward_depth_map': lambda: lib.nr_backward_depth_map(
            faces.data_ptr(), dm.data_ptr(), fi.data_ptr(), None, wm.data_ptr(), g_depth.data_ptr(), gf.data_ptr(),
            B, F, S, st),
    }
    # the two fused entry points the autograd operator actually calls -- the forward as the operator calls it: a kept workspace
    # with falling epoch numbers (filled once, here) and weights stored for covered pixels only
    ws_kept = torch.full((wsb,), 255, dtype=torch.uint8, device=dev)
    epoch = [254]

    def fused_forward():
        e = epoch[0]
        epoch[0] = e - 1 if e > 0 else 254
        if e == 0:
            ws_kept.fill_(255)
        return lib.nr_forward_rasterize(
            faces.data_ptr(), None, textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(),
            am.data_ptr(), vis.data_ptr(), bg.data_ptr(), 0, B, F, S, ts, 0.1, 100.0, eps,
            _lib.NR_FLAG_ZBUF_EPOCH | (e << 8) | _lib.NR_FLAG_SPARSE_WEIGHT_MAP, ws_kept.data_ptr(), wsb, st)
    calls['fused_forward_rasterize'] = fused_forward
    calls['fused_backward_rasterize'] = lambda: lib.nr_backward_rasterize(
        faces.data_ptr(), None, fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(), am.data_ptr(),
        g_rgb.data_ptr(), g_alpha.data_ptr(), g_depth.data_ptr(), gf.data_ptr(), gt.data_ptr(), B, F, S, ts, eps, k6_flags,
        vis.data_ptr(), bws.data_ptr(), bwsb, st)
    out = {}
    for name, call in calls.items():
        if only is not None and name not in only:
            continue
        for _ in range(2):
            _lib.check(call(), name)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = e0.elapsed_time(e1) * 1e3 / iters  # us
    # The path's dominant kernel ALONE (K6's band kernel, without the helper launches of its stage call): the MEASUREMENT build of
    # the library (libnr_hip_prof.so: the same sources with the timing hook of include/nr_hip_profile.h compiled in) brackets its
    # launch with a pair of HIP events on the launch stream.  Calls are issued back to back, the last call's pair is read, five
    # samples; once inside the stage call, once inside the fused backward (where the band workgroups also zero-fill grad_textures).
    try:
        plib = _lib.load_profile()
    except Exception:  # (no measurement build in the tree: the stage call's own duration stands in)
        plib = None
    if plib is not None and (only is None or 'backward_pixel_map' in only):
        pcalls = {
            'backward_pixel_map': lambda: plib.nr_backward_pixel_map(
                faces.data_ptr(), fi.data_ptr(), rgb.data_ptr(), am.data_ptr(), g_rgb.data_ptr(), g_alpha.data_ptr(),
                gf.data_ptr(), B, F, S, eps, 1, 1, k6_flags, vis.data_ptr(), bws.data_ptr(), bwsb, st),
            'fused_backward_rasterize': lambda: plib.nr_backward_rasterize(
                faces.data_ptr(), None, fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(), am.data_ptr(),
                g_rgb.data_ptr(), g_alpha.data_ptr(), g_depth.data_ptr(), gf.data_ptr(), gt.data_ptr(), B, F, S, ts, eps, k6_flags,
                vis.data_ptr(), bws.data_ptr(), bwsb, st)}
        for key, name in (('k6_band_kernel_alone', 'backward_pixel_map'), ('k6_band_kernel_alone_in_fused_backward', 'fused_backward_rasterize')):
            if only is not None and name not in only:
                continue
            samples = []
            _lib.check(plib.nr_profile_band_kernel(1), 'profile hook')
            try:
                for _ in range(5):
                    for _ in range(max(iters, 3)):
                        pcalls[name]()
                    ms = plib.nr_profile_band_kernel_ms()
                    if ms >= 0:
                        samples.append(ms * 1e3)
                        # (which of the two band kernels the library picked for this call: asked, not re-derived)
                        time_stages.band_kernel[(B, F, S, bool(k6_flags & 2))] = ('k_bpm_fast', 'k_bpm_row')[plib.nr_profile_band_kernel_which() == 1]
            finally:
                plib.nr_profile_band_kernel(0)
            if samples:
                out[key] = sum(samples) / len(samples)
    return out


time_stages.band_kernel = {}


def time_step(step, dev, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def renderer_end_to_end(device, batch, first_view, total_views, image_size, texture_size, iters=10):
    """SURVEY 8d "report end-to-end Renderer.render* time separately": the same scene through the public Renderer
    (fused HIP front-end -> rasterizer -> HIP image epilogue, anti-aliasing off so that S = image_size) with a squared-sum
    loss, forward + backward, wall clock per step."""
    import neural_renderer_amd as nr
    v, f = load_teapot()
    vertices = torch.from_numpy(v).to(device)[None].repeat(batch, 1, 1).requires_grad_(True)
    faces = torch.from_numpy(f).to(device)[None].repeat(batch, 1, 1)
    textures = torch.ones((batch, f.shape[0], texture_size, texture_size, texture_size, 3), device=device,
                          requires_grad=True)
    r = nr.Renderer()
    r.image_size, r.anti_aliasing = image_size, False
    r.eye = torch.tensor([nr.get_points_from_angles(2.732, 30., 360.0 * (first_view + i) / total_views)
                          for i in range(batch)], dtype=torch.float32, device=device)

    def rgb():
        vertices.grad = None
        textures.grad = None
        r.render(vertices, faces, textures).square().sum().backward()

    def sil():
        vertices.grad = None
        r.render_silhouettes(vertices, faces).square().sum().backward()

    out = {'render_fwd_bwd_ms': time_step(rgb, device, iters, 3), 'render_silhouettes_fwd_bwd_ms': time_step(sil, device, iters, 3),
           'what': 'Renderer.render / render_silhouettes + squared-sum loss, forward + backward, %d views, %dx%d, '
                   'anti_aliasing off' % (batch, image_size, image_size),
           'frontend': r.last_frontend, 'frontend_calls': dict(r.frontend_calls)}
    # the fused HIP front-end must be what ran: the module-by-module torch path is ~160 launches slower
    assert r.frontend_calls['torch'] == 0 and r.last_frontend == 'fused', r.frontend_calls
    return out


def face_light_row(device, batch=64, image_size=256, texture_size=4, iters=10):
    """Renderer.render + backward on config 4's shape (`batch` distinct ~5k-face spheres, fill_back -> 10 240 faces, random
    textures of `texture_size`) with lit, fill_back-duplicated textures handed to the rasterizer (the reference's data flow,
    renderer.py:77-103) and with per-face light colours instead (Renderer.face_light): ms per step and the
    largest difference between the two modes' images and gradients, relative to their largest value."""
    import neural_renderer_amd as nr
    rng = np.random.default_rng(3)
    # a latitude / longitude sphere of 5 120 triangles (config 4: "~5k faces", 10 240 with fill_back)
    n_lat, n_lon = 41, 64
    th = np.pi * np.arange(1, n_lat) / n_lat
    ph = 2 * np.pi * np.arange(n_lon) / n_lon
    ring = np.stack([np.sin(th)[:, None] * np.cos(ph)[None], np.cos(th)[:, None] * np.ones_like(ph)[None],
                     np.sin(th)[:, None] * np.sin(ph)[None]], axis=2).reshape(-1, 3)
    v0 = np.concatenate(([[0.0, 1.0, 0.0]], ring, [[0.0, -1.0, 0.0]])).astype(np.float32)
    idx = lambda i, j: 1 + i * n_lon + (j % n_lon)
    tri = []
    for j in range(n_lon):
        tri.append((0, idx(0, j + 1), idx(0, j)))
        tri.append((len(v0) - 1, idx(n_lat - 2, j), idx(n_lat - 2, j + 1)))
        for i in range(n_lat - 2):
            tri.append((idx(i, j), idx(i, j + 1), idx(i + 1, j)))
            tri.append((idx(i + 1, j), idx(i, j + 1), idx(i + 1, j + 1)))
    f0 = np.array(tri, np.int32)
    vs = []
    for _ in range(batch):
        vv = v0 * (0.55 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
        vs.append((vv @ q).astype(np.float32))
    vertices0 = torch.from_numpy(np.stack(vs)).to(device)
    faces = torch.from_numpy(f0.astype(np.int32)).to(device)[None].repeat(batch, 1, 1)
    textures0 = torch.rand((batch, f0.shape[0], texture_size, texture_size, texture_size, 3), device=device)
    out = {'what': 'Renderer.render + squared-sum loss, forward + backward: %d meshes x %d faces (fill_back -> %d), '
                   'texture_size %d, %dx%d, anti_aliasing off' % (batch, f0.shape[0], 2 * f0.shape[0], texture_size,
                                                                  image_size, image_size)}
    keep = {}
    for flag in (False, True):
        r = nr.Renderer()
        r.image_size, r.anti_aliasing, r.face_light = image_size, False, flag
        r.eye = torch.tensor([nr.get_points_from_angles(2.732, 30., 360.0 * i / batch) for i in range(batch)],
                             dtype=torch.float32, device=device)
        v = vertices0.clone().requires_grad_(True)
        t = textures0.clone().requires_grad_(True)

        def step():
            v.grad = None
            t.grad = None
            img = r.render(v, faces, t)
            img.square().sum().backward()
            return img

        img = step()
        keep[flag] = (img.detach(), v.grad.clone(), t.grad.clone())
        out['face_light_ms' if flag else 'lit_textures_ms'] = time_step(step, device, iters, 3)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    out['max_rel_diff'] = {k: rel(keep[True][i], keep[False][i]) for i, k in enumerate(('images', 'grad_vertices', 'grad_textures'))}
    return out


def measure_time_protocol(device, batch_size, image_size=256, texture_size=2, caller_thread=False):
    """The reference's own timing protocol (misc/measure_time.py:12-18, 44-90) through the public Renderer with ITS
    defaults (anti-aliasing on => raster 2 x image_size, eye from get_points_from_angles(2.732, 30, azimuth)): 24 azimuths,
    `render_silhouettes` then `render`, forward (to the first pixel on the host) and backward of sum(images) timed
    separately, first iteration dropped.  The reference stops its backward clock without synchronising (Q9: it times the
    launch); here the clock stops after a device synchronize.  Milliseconds.  caller_thread: with torch's autograd kept on
    the calling thread (neural_renderer_amd.graph.backward_on_caller_thread: the hand-over to its device thread is ~100 us
    of a host-bound backward)."""
    import neural_renderer_amd as nr
    v, f = load_teapot()
    vertices = torch.from_numpy(v).to(device)[None].repeat(batch_size, 1, 1).requires_grad_(True)
    faces = torch.from_numpy(f).to(device)[None].repeat(batch_size, 1, 1)
    textures = torch.ones((batch_size, f.shape[0], texture_size, texture_size, texture_size, 3), device=device,
                          requires_grad=True)
    renderer = nr.Renderer()
    renderer.image_size = image_size
    out = {'batch_size': batch_size, 'image_size': image_size, 'raster': 2 * image_size if renderer.anti_aliasing else image_size,
           'anti_aliasing': bool(renderer.anti_aliasing),
           'autograd': 'backward on the calling thread' if caller_thread else 'torch default (device thread)'}
    if caller_thread:
        with nr.graph.backward_on_caller_thread():
            out.update({k: v for k, v in measure_time_protocol(device, batch_size, image_size, texture_size).items()
                        if k.endswith('_ms')})
        return out
    for name, call in (('silhouette', lambda: renderer.render_silhouettes(vertices, faces)),
                       ('texture', lambda: renderer.render(vertices, faces, textures))):
        tf, tb = [], []
        for azimuth in range(0, 360, 15):
            renderer.eye = nr.get_points_from_angles(2.732, 30, azimuth)
            vertices.grad = None
            textures.grad = None
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            images = call()
            _ = images.reshape(-1)[0].item()
            t1 = time.perf_counter()
            loss = images.sum()
            _ = loss.item()
            t2 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize(device)
            t3 = time.perf_counter()
            tf.append(t1 - t0)
            tb.append(t3 - t2)
        out['%s_forward_ms' % name] = float(np.mean(tf[1:]) * 1e3)
        out['%s_backward_ms' % name] = float(np.mean(tb[1:]) * 1e3)
    return out


def shard_rows(device, total_views, S, ts, eps, steps, gpus=(2, 4, 8)):
    """The step at the per-GPU shard sizes of a 2 / 4 / 8-GPU run on THIS GPU: views [0, b) of the `total_views` azimuths with
    global view 0 as the texture-depth reference, i.e. rank 0's share of the strong-scaling job.  Three ways to call it (ms
    per step, wall clock, K steps up to a device synchronize):
      autograd               Rasterize(...)(faces, textures) + torch.autograd.backward, torch's default autograd settings
      autograd_caller_thread the same inside neural_renderer_amd.graph.backward_on_caller_thread() (no hand-over of the
                             backward to autograd's device thread)
      function_protocol      fn.forward_gpu(inputs); fn.backward_gpu(inputs, grad_outputs): the chainer.Function protocol of
                             the reference (rasterize.py:467, :849), no autograd graph -- what Chainer itself calls"""
    import neural_renderer_amd as nr
    rows = []
    for b in [total_views // r for r in gpus if total_views % r == 0 and total_views // r >= 1]:
        faces, textures = build_scene(device, b, 0, total_views, S, ts)
        faces.requires_grad_(True)
        textures.requires_grad_(True)
        z_ref = faces.detach()[0].contiguous().clone()
        grads = upstream_gradients(faces, textures, S, eps, 99 + b, z_ref=z_ref)

        def autograd_step():
            faces.grad = None
            textures.grad = None
            fn = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)
            fn.faces_z_ref = z_ref
            torch.autograd.backward(list(fn(faces, textures)), list(grads))

        fd, td = faces.detach(), textures.detach()

        def protocol_step():
            fn = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)
            fn.faces_z_ref = z_ref
            fn.forward_gpu((fd, td))
            fn.backward_gpu((fd, td), grads)

        n = max(steps, 50)
        row = {'views': b, 'of_total_views': total_views, 'gpus_this_shard_belongs_to': total_views // b}
        for _ in range(30):
            autograd_step()
        # (median of five runs of n steps: host-bound loops jitter with the host's thread wake-ups)
        runs = sorted(time_step(autograd_step, device, n, 5) for _ in range(5))
        row['ms_autograd'], row['ms_autograd_min_max'] = runs[2], [runs[0], runs[-1]]
        with nr.graph.backward_on_caller_thread():
            runs = sorted(time_step(autograd_step, device, n, 5) for _ in range(5))
        row['ms_autograd_caller_thread'] = runs[2]
        runs = sorted(time_step(protocol_step, device, n, 5) for _ in range(5))
        row['ms_function_protocol'] = runs[2]
        for k in ('ms_autograd', 'ms_autograd_caller_thread', 'ms_function_protocol'):
            row[k.replace('ms_', 'mpixel_per_s_')] = b * S * S / (row[k] * 1e-3) / 1e6
        # the row's roofline: the whole step's compulsory bytes over its time, and how much of the step is NOT inside the two
        # fused C-ABI calls when they are issued back to back (HIP events around each on the stream: the kernels plus the gaps
        # between them) -- the launch-latency / host share of a small shard's step as a number
        st = time_stages(fd, td, S, eps, grads[0], grads[1], grads[2], 20, 0, only=('fused_forward_rasterize', 'fused_backward_rasterize'))
        calls_us = st.get('fused_forward_rasterize', 0.0) + st.get('fused_backward_rasterize', 0.0)
        wb = whole_step_bytes(b, faces.shape[1], S, ts)
        row['roofline'] = {'bound': 'hbm', 'algorithmic_bytes': wb, 'achieved': wb / (row['ms_function_protocol'] * 1e-3) / 1e9,
                           'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': wb / (row['ms_function_protocol'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           'fused_calls_us': calls_us, 'step_us': row['ms_function_protocol'] * 1e3,
                           'outside_the_calls_us': row['ms_function_protocol'] * 1e3 - calls_us,
                           'what': 'whole-step compulsory bytes over the function-protocol step; fused_calls_us: nr_forward_rasterize + '
                                   'nr_backward_rasterize timed alone with events on the stream (9 launches at these sizes: '
                                   'profiles/*_step_sequence_shards.txt)'}
        rows.append(row)
        del faces, textures, grads
    return rows


def host_floor(device, steps=300):
    """What a step costs the HOST whatever the batch: one teapot view at 16x16 (a few microseconds of device work per kernel, so
    the loop is bound by issuing it), the three ways of shard_rows, ms per step.  The difference between the first two rows is
    torch's hand-over of the backward to its device thread and back; the third is the operator's own Python + ~8 kernel launches."""
    import neural_renderer_amd as nr
    faces, textures = build_scene(device, 1, 0, 64, 16, 2)
    faces.requires_grad_(True)
    textures.requires_grad_(True)
    grads = upstream_gradients(faces, textures, 16, 1e-3, 5)
    fd, td = faces.detach(), textures.detach()

    def autograd_step():
        faces.grad = None
        textures.grad = None
        fn = nr.Rasterize(16, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
        torch.autograd.backward(list(fn(faces, textures)), list(grads))

    def protocol_step():
        fn = nr.Rasterize(16, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
        fn.forward_gpu((fd, td))
        fn.backward_gpu((fd, td), grads)

    out = {'what': 'one teapot view at 16x16, rgb+alpha+depth forward + backward: host-bound, ms per step (median of 5 runs)'}
    med = lambda fn: sorted(time_step(fn, device, steps, 10) for _ in range(5))[2]
    out['ms_autograd'] = med(autograd_step)
    with nr.graph.backward_on_caller_thread():
        out['ms_autograd_caller_thread'] = med(autograd_step)
    out['ms_function_protocol'] = med(protocol_step)
    return out


def cpu_baseline(faces, textures, S, eps, g_rgb, g_alpha, g_depth, sample_views, light=False, modes=(True, True, True), what='teapot views'):
    """The C oracle on this host: one thread on `sample_views` views (the contract's `value`, kind "port"), all cores on
    the whole batch, and the naive NumPy per-pixel loop on BASELINE configs[0]."""
    from oracle import oracle as O
    O.build()

    def run(n_views, threads, blocked):
        f = faces[:n_views].cpu().numpy()
        t = textures[:n_views].cpu().numpy()
        gr, ga, gd = (None if x is None else x[:n_views].cpu().numpy() for x in (g_rgb, g_alpha, g_depth))
        O.set_threads(threads)
        try:
            t0 = time.perf_counter()
            fn = O.Rasterize(S, 0.1, 100, eps, (0, 0, 0), *modes)
            fn.blocked = blocked
            fn(f, t) if modes[0] else fn(f)
            t1 = time.perf_counter()
            fn.backward(gr, ga, gd)
            t2 = time.perf_counter()
        finally:
            O.set_threads(0)
        return n_views * S * S / (t2 - t0) / 1e6, t1 - t0, t2 - t1

    v1, f1, b1 = run(sample_views, 1, False)
    out = {
        'value': v1, 'unit': 'Mpixel/s', 'cores': 1, 'kind': 'port',
        'sample': '%d of the %d %s, %dx%d, %s fwd+bwd, oracle/nr_oracle.c (gcc -O2, literal loop '
                  'order), fwd %.2f s bwd %.2f s' % (sample_views, faces.shape[0], what, S, S, '+'.join(n for n, m in zip(('rgb', 'alpha', 'depth'), modes) if m), f1, b1),
        'host_cpus': os.cpu_count(),
    }
    n_all = min(int(faces.shape[0]), 64)
    va, fa, ba = run(n_all, 0, True)
    out['all_cores'] = {
        'value': va, 'unit': 'Mpixel/s', 'cores': O.get_threads(), 'kind': 'port',
        'sample': 'all %d views, OpenMP build of the same oracle with the cache-blocked K2 loop order (bit-identical '
                  'results), fwd %.2f s bwd %.2f s' % (n_all, fa, ba)}
    if not light:
        # BASELINE configs[0] / north star "naive NumPy per-pixel loop": teapot, 1 view, 64x64 silhouette
        from oracle import numpy_naive as N
        f64 = build_scene_cpu_view(64)
        t0 = time.perf_counter()
        fi, _, _ = N.forward_face_index_map(f64, 64, 0.1, 100)[:3]
        alpha = N.forward_alpha_map(fi)
        t1 = time.perf_counter()
        g = np.random.default_rng(0).normal(size=alpha.shape).astype(np.float32)
        N.backward_pixel_map(f64, fi, None, alpha, None, g, 1e-4)
        t2 = time.perf_counter()
        out['numpy_naive_config1'] = {
            'value': 64 * 64 / (t2 - t0) / 1e6, 'unit': 'Mpixel/s', 'cores': 1, 'kind': 'port',
            'sample': 'BASELINE configs[0]: teapot, 1 view, 64x64 silhouette, oracle/numpy_naive.py (one Python iteration '
                      'per pixel over all faces), fwd %.2f s bwd %.2f s' % (t1 - t0, t2 - t1)}
    return out


def build_scene_cpu_view(S):
    """One teapot view (azimuth 0) as float32 NumPy faces [1,F,3,3] through the oracle's glue (CPU only)."""
    from oracle import oracle as O
    v, f = load_teapot()
    f = np.concatenate((f, f[:, ::-1]), axis=0)
    eye = O.get_points_from_angles(2.732, 30., 0.)
    vv = O.perspective(O.look_at(v[None], eye), 30.)
    return O.vertices_to_faces(vv, f[None]).astype(np.float32)


def grad_check(nr_fn_fi, faces, textures, S, eps, g_rgb, g_alpha, g_depth, n_views, modes=(True, True, True)):
    """Parity of the benchmarked batch: face_index_map mismatches and gradient errors against the oracle (sums of the
    reference's float terms carried in double), `n_views` views."""
    from oracle import oracle as O
    O.build()
    ref = O.Rasterize(S, 0.1, 100, eps, (0, 0, 0), *modes)
    ref.blocked = True
    ref(faces[:n_views].detach().cpu().numpy(), textures[:n_views].detach().cpu().numpy())
    r_gf, r_gt = ref.backward(*(None if g is None else g[:n_views].cpu().numpy() for g in (g_rgb, g_alpha, g_depth)),
                              accumulate_double=True)
    gf, gt = faces.grad[:n_views].cpu().numpy(), textures.grad[:n_views].cpu().numpy()
    fi = nr_fn_fi[:n_views].cpu().numpy()

    def stats(a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        err = np.abs(a - b)
        nz = np.abs(b) > 0
        rel = err[nz] / np.abs(b[nz])
        floor = 1e-3 * np.abs(b).max()
        return {'max_abs_err': float(err.max()), 'max_abs': float(np.abs(b).max()),
                'max_rel_err_elementwise': float(rel.max()) if rel.size else 0.0,
                'frac_elements_within_1e-4_rel': float(np.mean(rel <= 1e-4)) if rel.size else 1.0,
                'max_rel_err_floor_1e-3_of_max': float((err / np.maximum(np.abs(b), floor)).max())}
    return {'face_index_mismatch': int((fi != ref.face_index_map).sum()), 'grad_faces': stats(gf, r_gf),
            'grad_textures': stats(gt, r_gt),
            'checked': '%d views of rank 0 vs oracle (terms summed in double); elementwise max-rel-err is dominated by entries '
                       'that cancel to ~0 (the tests bound the floor form by the north star\'s 1e-4; NR_FLAG_EXACT_GRADIENT: 2e-6)' % n_views}


def csrc_tree_hash():
    """sha1 over the library's sources (neural_renderer_amd/csrc/*, include/*.h, the compiler flags): the stamp the counter passes
    carry (scripts/pmc_traffic.py) -- counters of another build are not this build's traffic."""
    import hashlib
    sys.path.insert(0, ROOT)
    from neural_renderer_amd import _build
    h = hashlib.sha1()
    for p in sorted(_build.SOURCES + [x for x in _build.HEADERS if x.endswith('.h')]):
        h.update(os.path.basename(p).encode())
        h.update(open(p, 'rb').read())
    h.update(' '.join(_build.HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def raster_512_kernel_traffic(kernel):
    """HBM bytes per launch of the K6 band kernel at raster 512 (the anti-aliasing row) from the newest committed counter file
    of that shape, profiles/r*_pmc_hbm_traffic_S512.json -- (bytes, file), or (None, why) when there is none or it was collected
    on other sources than this tree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic_S512.json')))
    if not files or not kernel:
        return None, 'no counter file of this shape'
    rel = 'profiles/' + os.path.basename(files[-1])
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
    except Exception:
        return None, rel + ' unreadable'
    if (rec.get('_build') or {}).get('csrc_sha1') != csrc_tree_hash():
        return None, rel + ' was collected on other sources than this tree'
    for name, v in rec.get('backward_pixel_map', {}).get('kernels', {}).items():
        if name.startswith(kernel):
            return v['fetch'] + v['write'], rel
    return None, rel + ' holds no ' + kernel


def profile_records():
    """Counter-derived figures of the committed profiles (they cannot be collected inside a timed run): HBM traffic of the
    dominant stage and the VALU instruction count of the K6 kernel.  Labelled with their source file.  The file carries the hash
    of the sources it was collected on (`_build.csrc_sha1`); on another tree the records are dropped (`stale`): the line then
    has null traffic instead of another build's."""
    out = {}
    p = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    if os.path.exists(p):
        try:
            rec = json.load(open(p))
            stamp = rec.get('_build', {}).get('csrc_sha1')
            out['stamp'] = {'file_csrc_sha1': stamp, 'tree_csrc_sha1': csrc_tree_hash()}
            if stamp == out['stamp']['tree_csrc_sha1']:
                out['pmc'] = rec
            else:
                out['stale'] = True
        except Exception:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (defaults: ~50 ms of device time.  The first ~30 ms after an idle period run at ramping clocks -- 20 steps behind 3
    # warm-up steps measured 0.383-0.395 ms per step, 100 behind 30 0.366-0.372 on the same box)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--prewarm-ms', type=float, default=250.0, help='untimed milliseconds of the same step in front of the warm-up '
                    'steps (the device reaches its steady clocks; 0 = none)')
    ap.add_argument('--batch', type=int, default=64, help='views of the whole job: split over the GPUs (strong scaling); the '
                    '`weak_scaling` object of an N > 1 run renders this many PER GPU')
    ap.add_argument('--image-size', type=int, default=256, help='raster size S (anti-aliasing off)')
    ap.add_argument('--texture-size', type=int, default=2)
    ap.add_argument('--cpu-sample-views', type=int, default=32,
                    help='views timed on ONE host thread of the CPU oracle, rank 0 of a 1-GPU run only (0 = skip all CPU '
                         'baselines); 32 views ~ 13 s')
    ap.add_argument('--stage-iters', type=int, default=20)
    ap.add_argument('--check-views', type=int, default=64, help='views compared with the oracle (all host cores)')
    ap.add_argument('--gather', action='store_true', help='also all_gather the rendered images each step (RCCL)')
    ap.add_argument('--graph', action='store_true', help='replay the step from a captured HIP graph')
    ap.add_argument('--exact', action='store_true', help='K6 with the reference\'s own arithmetic (NR_FLAG_EXACT_GRADIENT)')
    ap.add_argument('--no-pin', action='store_true', help='do not pin the process to one L3 group of cores')
    ap.add_argument('--no-shard-rows', action='store_true', help='skip the 32 / 16 / 8-view rows of a 1-GPU run')
    ap.add_argument('--light', action='store_true', help='headline step + stage timings only (no extra rows, no Renderer '
                                                         'end-to-end, oracle check on 2 views)')
    ap.add_argument('--workload', choices=('teapot', 'c4'), default='teapot',
                    help='teapot: the metric\'s configuration (BASELINE.json configs[1] at batch 64); c4: BASELINE.json configs[3], '
                         '512 random meshes x 10 240 faces, texture_size 4, 256x256 textured RGB, the meshes split over the GPUs '
                         '(--gather: the all-gather of the rendered shards the configuration names)')
    args = ap.parse_args()
    c4 = args.workload == 'c4'
    if c4:  # (the job is the configuration's 512 meshes, 4 x 4 x 4 textures, RGB only; the headline's extra rows do not apply)
        args.batch = C4_MESHES if args.batch == 64 else args.batch
        args.texture_size = 4 if args.texture_size == 2 else args.texture_size
        args.light = True
    modes = (True, False, False) if c4 else (True, True, True)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run' % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    from neural_renderer_amd import distributed as nrd
    _, _, dev = nrd.init_from_env()  # one process per GPU; backend "nccl" = RCCL
    # one process per GPU, pinned to the cores of one L3 group near that GPU (rank r: the r-th group): where torch's autograd
    # device thread wakes up decides what its hand-over costs a host-bound step (distributed.pin_to_l3_group; --no-pin: leave it)
    all_cpus = os.sched_getaffinity(0)
    pinned = None if args.no_pin else nrd.pin_to_l3_group(int(os.environ.get('LOCAL_RANK', '0')), dev.index)

    class on_all_cores(object):
        """The CPU oracle (checker / CPU baseline) runs on every core the process was given: its OpenMP threads are created
        inside and inherit the calling thread's affinity of that moment."""

        def __enter__(self):
            self.prev = os.sched_getaffinity(0)
            os.sched_setaffinity(0, all_cpus)

        def __exit__(self, *exc):
            os.sched_setaffinity(0, self.prev)
            return False
    dist = None
    if world > 1 or nrd._force():  # NR_DIST_FORCE=1: one rank, but through RCCL all the same (tests/test_rccl_gpu.py)
        import torch.distributed as dist

    import neural_renderer_amd as nr
    G, S, ts, eps = args.batch, args.image_size, args.texture_size, 1e-3  # G: views of the whole job
    # strong scaling: rank r owns views [start, stop) of the G azimuths 360 i / G
    start, stop = nrd.shard_bounds(G, rank, world)
    B = stop - start
    if B < 1:
        raise SystemExit('--batch %d gives rank %d of %d no view' % (G, rank, world))
    faces, textures = build_scene_c4(dev, start, B, ts) if c4 else build_scene(dev, B, start, G, S, ts)
    F = faces.shape[1]
    faces.requires_grad_(True)
    textures.requires_grad_(True)
    # textures are sampled with the depths of GLOBAL view 0 -- one 177 KB broadcast before the timed region
    # makes the shards of an N > 1 run the same computation as the unsharded batch (tests/test_sharding_gpu.py)
    z_ref = nrd.broadcast_reference_faces(faces.detach()) if (world > 1 or nrd._force()) else None
    g_rgb, g_alpha, g_depth = upstream_gradients(faces, textures, S, eps, 1234 + rank, z_ref=z_ref, modes=modes)

    gather = args.gather and dist is not None
    last = {}

    def make_step(f, t, size, grads, with_gather=False, exact=None, ref=None, total=G):
        def step():
            f.grad = None
            t.grad = None
            fn = nr.Rasterize(size, 0.1, 100, eps, (0, 0, 0), *modes)
            fn.exact_gradient = args.exact if exact is None else exact
            fn.faces_z_ref = ref
            outs = fn(f, t)
            if with_gather:  # the downstream loss wants the whole batch: one all-gather of the rendered shards (RCCL / xGMI)
                last['gathered'] = nrd.all_gather_images(outs[0].detach(), total=total)
            torch.autograd.backward([o for o in outs if o is not None], [g for g in grads if g is not None])
            last['fi'] = fn.face_index_map
        return step

    grads = (g_rgb, g_alpha, g_depth)
    step = make_step(faces, textures, S, grads, with_gather=gather, ref=z_ref)
    local_step = make_step(faces, textures, S, grads, ref=z_ref)  # rank-local work only: no collective inside

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Optional: capture the step once in a HIP graph and replay it (B = 64 is GPU-bound: no gain; small batches are
    # host-bound and gain, see scripts/bench_configs.py).
    run, mode = step, 'eager'
    if args.graph and not gather:
        from neural_renderer_amd.graph import capture
        run, mode = capture(step, dev), 'hipgraph'

    # Timing protocol: all ranks leave a barrier + device synchronize together, each times ITS OWN K steps up to its own
    # device synchronize (wall clock, and HIP events on the launch stream as a cross-check), then a second barrier closes the
    # bracket and the MAX over ranks is reduced.  The closing barrier is not inside the timed region: on RCCL it costs tens to
    # hundreds of microseconds, which would be charged to every N > 1 point of an 8.8 ms measurement.
    # Clocks first: a device that idled through the set-up above runs its first few hundred ms of work at ramping clocks (20
    # steps behind 3 warm-up steps: 0.384 ms per step; behind 80 ms of the same steps 0.374, behind 300 ms 0.369; 100 steps
    # behind 30: 0.367).  W warm-up steps of 0.4 ms do not get there, so the same step runs untimed for --prewarm-ms before
    # the W warm-up steps and the K timed ones (reported in `timing.prewarm`).
    # (Rank-local steps only: the loop is bounded by time, so the ranks run different numbers of them -- no collective inside.)
    def timed(run_fn, pre_fn):
        """(seconds for K steps: MAX over ranks, rank-local HIP-event ms per step, pre-warm steps)."""
        n_pre, t_pre = 0, time.perf_counter()
        if args.prewarm_ms > 0:  # (one-time costs of the very first calls -- allocations, kernel attributes -- are not device work)
            pre_fn()
            torch.cuda.synchronize(dev)
            t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for _ in range(10):
                pre_fn()
            torch.cuda.synchronize(dev)
            n_pre += 10
        for _ in range(args.warmup):
            run_fn()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            run_fn()
        ev1.record()
        torch.cuda.synchronize(dev)
        seconds = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1) / args.steps
        barrier()
        if dist is not None:
            t = torch.tensor([seconds], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        return seconds, ev_ms, n_pre

    # `cold`: the driver's protocol as it stands -- W warm-up steps, K timed steps, no pre-warm -- measured FIRST in the process,
    # on a device that idled through the set-up (ramping clocks: typically 5-10 % above the pre-warmed figure below)
    keep_prewarm, args.prewarm_ms = args.prewarm_ms, 0.0
    cold_elapsed, _, _ = timed(run, local_step if gather else run)
    args.prewarm_ms = keep_prewarm
    cold_ms = cold_elapsed / args.steps * 1e3
    elapsed, event_ms_per_step, prewarm_steps = timed(run, local_step if gather else run)
    eager_ms = None
    if mode == 'hipgraph':  # also report the eager number
        eager_ms = time_step(local_step, dev, args.steps, 2)

    ms_per_step = elapsed / args.steps * 1e3
    total_pixels = G * S * S  # the whole job: every rank's shard
    value = total_pixels / (ms_per_step * 1e-3) / 1e6

    # The job of rounds 1-3 at N > 1: G views PER GPU (rank r: views [r G, (r + 1) G) of N G azimuths), same protocol.
    weak = None
    if world > 1 or nrd._force():
        wf, wt = build_scene_c4(dev, rank * G, G, ts) if c4 else build_scene(dev, G, rank * G, world * G, S, ts)
        wf.requires_grad_(True)
        wt.requires_grad_(True)
        wz = nrd.broadcast_reference_faces(wf.detach())
        wg = upstream_gradients(wf, wt, S, eps, 4242 + rank, z_ref=wz, modes=modes)
        w_step = make_step(wf, wt, S, wg, with_gather=gather, ref=wz, total=world * G)
        w_local = make_step(wf, wt, S, wg, ref=wz)
        w_elapsed, _, _ = timed(w_step, w_local)
        w_ms = w_elapsed / args.steps * 1e3
        weak = {'scaling': 'weak', 'views_per_gpu': G, 'views_total': world * G, 'ms_per_step': w_ms,
                'value': world * G * S * S / (w_ms * 1e-3) / 1e6, 'unit': 'Mpixel/s',
                'what': 'the same step on %d views per GPU (%d in all): the N > 1 points of rounds 1-3' % (G, world * G)}
        del wf, wt, wg, w_step, w_local

    if rank == 0:
        local_step()  # gradients of the checked batch; rank 0 alone runs this, so it must not contain a collective
        torch.cuda.synchronize(dev)
        with on_all_cores():
            check = grad_check(last['fi'], faces, textures, S, eps, g_rgb, g_alpha, g_depth,
                               min(B, 2 if args.light else args.check_views), modes=modes)

        k6_flags = 2 if args.exact else 0
        # (the per-stage microbenchmark drives every stage entry point: with all three upstream gradients whatever the step's mode)
        sg = (g_rgb, g_alpha, g_depth) if all(modes) else upstream_gradients(faces.detach(), textures.detach(), S, eps, 99 + rank, z_ref=z_ref)
        stages = time_stages(faces.detach(), textures.detach(), S, eps, sg[0], sg[1], sg[2], args.stage_iters, k6_flags)
        del sg
        stage_bytes = algorithmic_bytes(B, F, S, ts)
        dominant = max(stage_bytes, key=lambda k: stages[k])
        # `achieved`: the dominant stage's algorithmic bytes over the duration of its dominant KERNEL alone, measured live with
        # HIP events around that kernel's launch (K6: the band kernel; the figure `rocprofv3 --stats` reports for it); the whole
        # stage call, helper launches included, is the `stage_call` object
        kernel_us = stages.get('k6_band_kernel_alone') if dominant == 'backward_pixel_map' else None
        launch_us = kernel_us or stages[dominant]
        achieved = stage_bytes[dominant] / (launch_us * 1e-6) / 1e9
        prof_all = profile_records()
        prof = prof_all.get('pmc', {})
        # (the committed counters are launches of the headline shape: 64 teapot views at 256 x 256)
        traffic_rec = prof.get(dominant, {}) if (B == 64 and S == 256 and not c4) else {}
        # HBM bytes of the dominant KERNEL alone (the scope of `avg_launch_us`) and of the whole stage call (the scope of `stage_call`)
        kname = STAGE_KERNEL.get(dominant, dominant)
        if dominant == 'backward_pixel_map':  # (the stage microbenchmark calls K6 with rgb + alpha)
            kname = k6_band_kernel(B, F, S, True, True, args.exact)
        krec = {k: v for k, v in traffic_rec.get('kernels', {}).items() if kname and k.startswith(kname)}
        kernel_traffic = sum(v['fetch'] + v['write'] for v in krec.values()) if krec else None
        stage_traffic = traffic_rec.get('hbm_bytes_per_launch')
        step_bytes = (whole_step_bytes_rgb if c4 else whole_step_bytes)(G, F, S, ts)  # the whole job's compulsory bytes against its step time
        roofline = {
            'bound': 'hbm', 'kernel': (kname + (' (exact mode)' if args.exact and dominant == 'backward_pixel_map' else '')) if kname else None,
            'stage': dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': kernel_traffic,
            'traffic_ratio': (kernel_traffic / stage_bytes[dominant]) if kernel_traffic else None,
            'traffic_scope': 'HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes) of the dominant kernel '
                             'ALONE per launch -- the scope of avg_launch_us; the whole stage call with its helper launches: stage_call.traffic',
            'traffic_source': {'file': 'profiles/pmc_latest.json', 'stamp': prof_all.get('stamp'),
                               'stale': bool(prof_all.get('stale')),
                               'note': 'separate --pmc passes of scripts/gpu_profile_round.sh on the committed build (counters cannot be '
                                       'collected inside a timed run); null when this run is not the profiled shape (64 teapot views, 256 x 256) or when '
                                       'the file was collected on other sources than this tree (`stale`: the stamps differ)'},
            'algorithmic_bytes_per_launch': stage_bytes[dominant], 'avg_launch_us': launch_us,
            'timing': ('HIP events recorded by the library on the launch stream right in front of and behind the launch of the '
                       'dominant kernel (nr_profile_band_kernel: K6\'s band kernel inside nr_backward_pixel_map, calls issued back to '
                       'back, mean of five samples); the same kernel inside the fused backward: `in_fused_backward_us`'
                       if kernel_us else
                       'HIP events on the launch stream around the stage call nr_%s (the dominant kernel plus its helper '
                       'launches)' % dominant),
            'in_fused_backward_us': stages.get('k6_band_kernel_alone_in_fused_backward') if kernel_us else None,
            'stage_call': {'what': 'the whole stage call nr_%s, helper launches included (HIP events around the call; '
                                   'profiles/README.md lists the per-kernel rocprofv3 durations it adds up from)' % dominant,
                           'avg_us': stages[dominant], 'achieved': stage_bytes[dominant] / (stages[dominant] * 1e-6) / 1e9,
                           'frac': stage_bytes[dominant] / (stages[dominant] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           'traffic': stage_traffic, 'traffic_ratio': (stage_traffic / stage_bytes[dominant]) if stage_traffic else None},
            'whole_step': {
                'algorithmic_bytes': step_bytes,
                'definition': 'SURVEY 8d: 76 B/pixel + (108 + 24 ts^3) B/face for rgb only' if c4 else
                              'SURVEY 8d: 92 B/pixel + (108 + 24 ts^3) B/face for rgb+alpha+depth',
                'achieved': step_bytes / (ms_per_step * 1e-3) / 1e9,
                'frac': step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        fi_map = last['fi']
        covered = float((fi_map >= 0).float().mean().item())
        visible = float(torch.stack([torch.bincount(fi_map[b][fi_map[b] >= 0].flatten().long(), minlength=F) > 0
                                     for b in range(B)]).float().mean().item())
        scaled = coverage_scaled_bytes(B, F, S, ts, covered, visible)
        roofline['stages'] = {
            'covered_pixel_fraction': covered, 'visible_face_fraction': visible,
            'per_stage': {k: {'avg_launch_us': stages[k], 'algorithmic_bytes': stage_bytes[k], 'coverage_scaled_bytes': scaled[k],
                              'achieved_GBps': scaled[k] / (stages[k] * 1e-6) / 1e9,
                              'frac': scaled[k] / (stages[k] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              'hbm_bytes_from_profiles': prof.get(k, {}).get('hbm_bytes_per_launch')} for k in stage_bytes},
            'note': 'frac of each stage call priced with coverage-scaled compulsory bytes (reads the kernels rightly skip on '
                    'uncovered pixels / invisible faces are not counted)'}
        valu = prof.get('_valu_issue', {})
        if valu.get('insts_valu_per_launch'):
            floor_us = valu['insts_valu_per_launch'] * NS_PER_WAVE_INSTR / NUM_SIMDS * 1e-3
            roofline['valu_issue'] = {
                'kernel': valu.get('kernel'), 'insts_valu_per_launch': valu['insts_valu_per_launch'],
                'ns_per_wave_instr_per_simd': NS_PER_WAVE_INSTR, 'simds': NUM_SIMDS, 'issue_floor_us': floor_us,
                'stage_us': stages['backward_pixel_map'], 'frac': floor_us / stages['backward_pixel_map'],
                'source': 'SQ_INSTS_VALU from profiles/pmc_latest.json (' + str(valu.get('source')) + '); stage time measured in this run'}
            ctr = valu.get('counters_per_launch', {})
            if ctr.get('SQ_THREAD_CYCLES_VALU') and ctr.get('SQ_INSTS_VALU'):
                # issued lanes that execute (the exec mask), from the counters; and the share of them that does useful work in the
                # band kernel's visits (pixels inside a sweep / lanes of the steps walked: scripts/row_stats.py, work counters of a
                # -DNR_ROW_STATS build, committed with the counter passes)
                roofline['lane_efficiency'] = {
                    'executing_lanes_per_issued_lane': ctr['SQ_THREAD_CYCLES_VALU'] / (64.0 * ctr['SQ_INSTS_VALU']),
                    'definition': 'SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU) of ' + str(valu.get('kernel')),
                    'visits': valu.get('row_stats')}
        extra_rows, e2e, exact_row = [], None, None
        if not args.light:
            # anti-aliasing on: raster 2 x image_size (the Renderer default), same views
            faces2 = faces.detach().clone().requires_grad_(True)
            tex2 = textures.detach().clone().requires_grad_(True)
            g2 = upstream_gradients(faces2, tex2, 2 * S, eps, 4321)
            ms2 = time_step(make_step(faces2, tex2, 2 * S, g2), dev, max(3, args.steps // 4), 2)
            # the roofline of THIS row (the reference's default configuration: Renderer() renders with anti-aliasing, raster 2 S):
            # its dominant stage's algorithmic bytes over that stage's dominant kernel alone, as for the headline
            st2 = time_stages(faces2.detach(), tex2.detach(), 2 * S, eps, g2[0], g2[1], g2[2], max(3, args.stage_iters // 2), k6_flags,
                              only=('forward_face_index_map', 'forward_texture_sampling', 'backward_pixel_map'))
            sb2 = algorithmic_bytes(B, F, 2 * S, ts)['backward_pixel_map']
            k2_us = st2.get('k6_band_kernel_alone') or st2['backward_pixel_map']
            wb2 = whole_step_bytes(B, F, 2 * S, ts)
            kname2 = k6_band_kernel(B, F, 2 * S, True, True, args.exact)
            tr2, tr2_src = raster_512_kernel_traffic(kname2) if (B, S, ts) == (64, 256, 2) else (None, 'not the profiled shape')
            extra_rows.append({'row': 'anti_aliasing on: raster %dx%d for image_size %d' % (2 * S, 2 * S, S), 'ms_per_step': ms2,
                               'mpixel_per_s_raster': B * 4 * S * S / (ms2 * 1e-3) / 1e6,
                               'mpixel_per_s_image': B * S * S / (ms2 * 1e-3) / 1e6,
                               'roofline': {'bound': 'hbm', 'stage': 'backward_pixel_map',
                                            'kernel': kname2, 'avg_launch_us': k2_us,
                                            'algorithmic_bytes_per_launch': sb2, 'achieved': sb2 / (k2_us * 1e-6) / 1e9,
                                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': sb2 / (k2_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                            'stage_call_us': st2['backward_pixel_map'],
                                            'whole_step': {'algorithmic_bytes': wb2, 'achieved': wb2 / (ms2 * 1e-3) / 1e9,
                                                           'frac': wb2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                            'traffic': tr2, 'traffic_ratio': (tr2 / sb2) if tr2 else None,
                                            'traffic_note': 'the band kernel alone, counter passes of this shape: ' + tr2_src}})
            del faces2, tex2, g2
            ones = upstream_gradients(faces, textures, S, eps, 0, all_ones=True)
            ms3 = time_step(make_step(faces, textures, S, ones), dev, args.steps, 2)
            extra_rows.append({'row': 'all-ones upstream gradient (reference misc/measure_time.py:60)', 'ms_per_step': ms3,
                               'mpixel_per_s_raster': B * S * S / (ms3 * 1e-3) / 1e6})
            if not args.exact:  # the bit-faithful K6 mode on the headline batch (NR_FLAG_EXACT_GRADIENT)
                ms5 = time_step(make_step(faces, textures, S, (g_rgb, g_alpha, g_depth), exact=True), dev, args.steps, 2)
                exact_row = {'ms_per_step': ms5, 'value': G * S * S / (ms5 * 1e-3) / 1e6, 'unit': 'Mpixel/s',
                             'what': 'the same step with NR_FLAG_EXACT_GRADIENT: K6 with the reference\'s own arithmetic per term, '
                                     'sums in double (bound 2e-6 against the exactly summed reference terms)'}
                extra_rows.append({'row': 'headline step with NR_FLAG_EXACT_GRADIENT (K6 with the reference\'s own arithmetic per term)',
                                   'ms_per_step': ms5, 'mpixel_per_s_raster': B * S * S / (ms5 * 1e-3) / 1e6})
            # the same step through the operator's chainer.Function protocol (forward_gpu / backward_gpu: no autograd graph, no
            # hand-over of the backward to torch's device thread): the device-bound figure whatever the host's thread wake-ups cost
            fd, td = faces.detach(), textures.detach()

            def protocol_step():
                fn = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)
                fn.faces_z_ref = z_ref
                fn.forward_gpu((fd, td))
                fn.backward_gpu((fd, td), grads)
            ms6 = time_step(protocol_step, dev, args.steps, 5)
            extra_rows.append({'row': 'headline step through Rasterize.forward_gpu / backward_gpu (the Function protocol, no autograd graph)',
                               'ms_per_step': ms6, 'mpixel_per_s_raster': B * S * S / (ms6 * 1e-3) / 1e6})
            try:  # the same step replayed from a captured HIP graph: what the ~15 launches and allocations cost the host
                from neural_renderer_amd.graph import capture
                ms4 = time_step(capture(local_step, dev), dev, args.steps, 2)
                extra_rows.append({'row': 'headline step replayed from a captured HIP graph (neural_renderer_amd.graph)',
                                   'ms_per_step': ms4, 'mpixel_per_s_raster': B * S * S / (ms4 * 1e-3) / 1e6})
            except Exception as ex:  # pragma: no cover
                extra_rows.append({'row': 'hip graph capture failed: %s' % ex, 'ms_per_step': float('nan')})
            e2e = renderer_end_to_end(dev, B, rank * B, world * B, S, ts)
            e2e['reference_protocol'] = {
                'what': 'misc/measure_time.py:12-18, 44-90 through Renderer() with the reference defaults (anti_aliasing on: raster 512 '
                        'for image_size 256); forward and backward of sum(images) timed separately, ms, mean of 23 azimuths',
                'rows': [measure_time_protocol(dev, 1), measure_time_protocol(dev, B),
                         measure_time_protocol(dev, 1, caller_thread=True)]}
            e2e['face_light'] = face_light_row(dev, B)
        shards = None
        if world == 1 and not args.no_shard_rows and not args.light:
            shards = {'rows': shard_rows(dev, G, S, ts, eps, args.steps),
                      'what': 'rank 0\'s shard of the %d-view job at 2 / 4 / 8 GPUs, timed on this one GPU (ms per step)' % G}
            # no collective in forward + backward: the R-GPU step is the slowest rank's shard step
            shards['host_floor'] = host_floor(dev)
            shards['predicted_strong_scaling'] = [
                {'n_gpus': r['gpus_this_shard_belongs_to'], 'views_per_gpu': r['views'],
                 'value_autograd': G * S * S / (r['ms_autograd'] * 1e-3) / 1e6,
                 'value_autograd_caller_thread': G * S * S / (r['ms_autograd_caller_thread'] * 1e-3) / 1e6,
                 'value_function_protocol': G * S * S / (r['ms_function_protocol'] * 1e-3) / 1e6, 'unit': 'Mpixel/s',
                 'efficiency_vs_this_run_autograd': (ms_per_step / r['ms_autograd']) / r['gpus_this_shard_belongs_to']}
                for r in shards['rows']]
        cpu = None
        if args.cpu_sample_views > 0:
            # (N > 1: rank 0's shard, one thread and all cores; the naive-NumPy row only in the 1-GPU run)
            with on_all_cores():
                cpu = cpu_baseline(faces.detach(), textures.detach(), S, eps, g_rgb, g_alpha, g_depth,
                                   min(2 if c4 else args.cpu_sample_views, B), light=args.light or world > 1, modes=modes,
                                   what='config-4 meshes' if c4 else 'teapot views')
        line = {
            'metric': ('rasterize fwd+bwd Mpixels/sec @256x256, BASELINE configs[3]: 512 meshes x ~5k faces sharded over the GPUs'
                       if c4 else 'rasterize fwd+bwd Mpixels/sec @256x256 batch=64'), 'value': value, 'unit': 'Mpixel/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'data_note': 'synthetic cameras, textures and upstream gradients; the mesh is the reference\'s teapot.obj (embedded in '
                         'tests/golden/reference_fixtures.npz), as BASELINE.json names it',
            'config': {
                'workload': ('BASELINE.json configs[3] (SURVEY 8d C4): %d seeded random meshes (icosphere of 5 120 faces, radial noise, '
                             'random rotation; fill_back -> %d faces) split over %d GPU(s) (%d per GPU), raster %dx%d, texture_size %d '
                             'random textures, RGB forward + backward through the Rasterize autograd operator%s'
                             % (G, F, world, B, S, S, ts, ', one all_gather of the rendered shards per step' if gather else '')) if c4 else
                            'teapot.obj (2464 faces, fill_back -> %d), %d azimuth views split over %d GPU(s) (%d per GPU), raster '
                            '%dx%d (anti_aliasing off), texture_size %d, rgb+alpha+depth forward + backward through the '
                            'Rasterize autograd operator' % (F, G, world, B, S, S, ts),
                'views_total': G, 'views_per_gpu': B, 'image_size': S, 'num_faces': F, 'texture_size': ts, 'eps': eps,
                'k6_numerics': 'exact (reference arithmetic per term, sums in double)' if args.exact else
                               'default (float terms via fused multiply-adds and v_rcp_f32, ~1 ulp per term; tests bound the '
                               'deviation from the exactly summed reference terms by 1e-4; measured levels: grad_check and '
                               'profiles/*_parity_errors.jsonl)',
                'parallelism': 'the batch of views split over %d GPU(s): rank r renders views [r*%d/%d, (r+1)*%d/%d), no collective%s'
                               % (world, G, world, G, world, ' + all_gather(rgb)' if gather else ''),
            },
            'cold': {'ms_per_step': cold_ms, 'value': total_pixels / (cold_ms * 1e-3) / 1e6, 'unit': 'Mpixel/s',
                     'what': 'the same K steps behind W warm-up steps WITHOUT the untimed pre-warm, measured first in the process'},
            'exact': exact_row,
            'weak_scaling': weak, 'shard_rows': shards,
            'roofline': roofline, 'cpu_baseline': cpu, 'stages_us': stages, 'grad_check': check,
            'extra_rows': extra_rows, 'launch_mode': mode, 'eager_ms_per_step': eager_ms, 'renderer_end_to_end': e2e,
            'timing': {'protocol': 'barrier + synchronize | K steps timed per rank up to its own synchronize (wall clock) | barrier; '
                                   'MAX over ranks; no collective inside the timed region' + (' except the requested all_gather' if gather else ''),
                       'rank0_hip_event_ms_per_step': event_ms_per_step,
                       'prewarm': {'ms': args.prewarm_ms, 'steps': prewarm_steps,
                                   'why': 'steady clocks before the W warm-up and K timed steps (untimed)'},
                       'effective_warmup_steps': prewarm_steps + args.warmup,
                       'cpu_affinity': ('pinned to one L3 group: %d logical CPUs from %d' % (len(pinned), min(pinned))) if pinned else 'not pinned',
                       'backend': (dist.get_backend() if dist is not None else None)},
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
