#!/usr/bin/env python3
"""bench.py -- rasterize fwd+bwd Mpixels/s on teapot.obj, 256x256, batch 64 per GPU (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the rasterizer hot path (Rasterize forward + backward through the product's autograd
operator, i.e. the five C-ABI stages) over one batch of 64 views of the teapot at raster size 256 with
RGB + alpha + depth outputs all enabled.  Inputs (projected faces, lit textures, upstream gradients) are
resident in HBM before the timed region.  The batch-of-views dimension shards across GPUs without any
collective ("weak" scaling: 64 views per GPU); rank 0 prints ONE JSON line.

Extra objects on the line (tier contract):
  roofline      the dominant kernel's algorithmic HBM bytes / its measured average launch duration
                (HIP events on the launch stream), against the 8 TB/s HBM3E peak
  cpu_baseline  the C oracle (oracle/nr_oracle.c, a literal single-thread port of the reference's
                algorithm) timed on a bounded sample of the same workload on this host
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


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


# dominant kernel of each stage call (the rest of a stage are small helper launches, see profiles/README.md)
STAGE_KERNEL = {
    'forward_face_index_map': 'k_face_raster', 'forward_texture_sampling': 'k_shade', 'backward_pixel_map': 'k_bpm_band',
    'backward_textures': 'k_backward_textures_face', 'backward_depth_map': 'k_backward_depth_face',
}


def algorithmic_bytes(B, F, S, ts):
    """Compulsory HBM traffic per launch of each stage (every input read once, every output written once;
    recomputable intermediates count zero) -- DESIGN.md 'Kernels'."""
    P, N = B * S * S, B * F
    return {
        'face_setup': N * (36 + 36 + 8),
        'raster_tiles': N * (36 + 36 + 8) + P * (4 + 12 + 4),
        'shade': P * (4 + 12 + 4) + N * 12 * ts ** 3 + P * (12 + 4),
        'backward_pixel_map': P * (4 + 12 + 4 + 12 + 4) + N * 72,
        'backward_textures': P * (4 + 12 + 4 + 12) + N * 24 * ts ** 3,
        'backward_depth_map': P * (4 + 12 + 4 + 4) + N * 72,
    }


def time_stages(faces, textures, S, eps, g_rgb, g_alpha, g_depth, iters):
    """Average duration of each C-ABI stage, measured with events on the stream the kernels are launched on."""
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
    bg = torch.zeros(3, device=dev)
    gf = torch.empty_like(faces)
    gt = torch.empty_like(textures)
    wsb = lib.nr_forward_workspace_bytes(B, F, S)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bwsb = lib.nr_backward_workspace_bytes(B, F, S, 1, 1)
    bws = torch.empty(max(bwsb, 1), dtype=torch.uint8, device=dev)

    calls = {
        'forward_face_index_map': lambda: lib.nr_forward_face_index_map(
            faces.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), None, B, F, S, 0.1, 100.0, ws.data_ptr(),
            wsb, st),
        'forward_texture_sampling': lambda: lib.nr_forward_texture_sampling(
            faces.data_ptr(), textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(), None,
            None, bg.data_ptr(), 0, am.data_ptr(), B, F, S, ts, eps, 0, st),
        'backward_pixel_map': lambda: lib.nr_backward_pixel_map(
            faces.data_ptr(), fi.data_ptr(), rgb.data_ptr(), am.data_ptr(), g_rgb.data_ptr(), g_alpha.data_ptr(),
            gf.data_ptr(), B, F, S, eps, 1, 1, bws.data_ptr(), bwsb, st),
        'backward_textures': lambda: lib.nr_backward_textures(
            fi.data_ptr(), None, None, faces.data_ptr(), wm.data_ptr(), dm.data_ptr(), g_rgb.data_ptr(),
            gt.data_ptr(), B, F, S, ts, eps, 0, st),
        'backward_depth_map': lambda: lib.nr_backward_depth_map(
            faces.data_ptr(), dm.data_ptr(), fi.data_ptr(), None, wm.data_ptr(), g_depth.data_ptr(), gf.data_ptr(),
            B, F, S, st),
    }
    # the two fused entry points the autograd operator actually calls
    calls['fused_forward_rasterize'] = lambda: lib.nr_forward_rasterize(
        faces.data_ptr(), textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(),
        am.data_ptr(), bg.data_ptr(), 0, B, F, S, ts, 0.1, 100.0, eps, 0, ws.data_ptr(), wsb, st)
    calls['fused_backward_rasterize'] = lambda: lib.nr_backward_rasterize(
        faces.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(), am.data_ptr(), g_rgb.data_ptr(),
        g_alpha.data_ptr(), g_depth.data_ptr(), gf.data_ptr(), gt.data_ptr(), B, F, S, ts, eps, 0, bws.data_ptr(), bwsb,
        st)
    out = {}
    for name, call in calls.items():
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
    return out


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

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / iters * 1e3

    def rgb():
        vertices.grad = None
        textures.grad = None
        r.render(vertices, faces, textures).square().sum().backward()

    def sil():
        vertices.grad = None
        r.render_silhouettes(vertices, faces).square().sum().backward()

    return {'render_fwd_bwd_ms': timed(rgb), 'render_silhouettes_fwd_bwd_ms': timed(sil),
            'what': 'Renderer.render / render_silhouettes + squared-sum loss, forward + backward, %d views, %dx%d, '
                    'anti_aliasing off' % (batch, image_size, image_size)}


def cpu_baseline(faces, textures, S, eps, g_rgb, g_alpha, g_depth, sample_views):
    """The C oracle on `sample_views` views of the same workload, one host thread."""
    from oracle import oracle as O
    O.build()
    f = faces[:sample_views].cpu().numpy()
    t = textures[:sample_views].cpu().numpy()
    gr, ga, gd = (x[:sample_views].cpu().numpy() for x in (g_rgb, g_alpha, g_depth))
    t0 = time.perf_counter()
    fn = O.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)
    fn(f, t)
    t1 = time.perf_counter()
    fn.backward(gr, ga, gd)
    t2 = time.perf_counter()
    pixels = sample_views * S * S
    return {
        'value': pixels / (t2 - t0) / 1e6, 'unit': 'Mpixel/s', 'cores': 1, 'kind': 'port',
        'sample': '%d of the 64 teapot views, %dx%d, rgb+alpha+depth fwd+bwd, oracle/nr_oracle.c (gcc -O2), '
                  'fwd %.2f s bwd %.2f s' % (sample_views, S, S, t1 - t0, t2 - t1),
        'host_cpus': os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='views per GPU')
    ap.add_argument('--image-size', type=int, default=256, help='raster size S (anti-aliasing off)')
    ap.add_argument('--texture-size', type=int, default=2)
    ap.add_argument('--cpu-sample-views', type=int, default=32,
                    help='views timed on the CPU oracle, rank 0 of a 1-GPU run only (0 = skip); 32 views ~ 13 s')
    ap.add_argument('--stage-iters', type=int, default=20)
    ap.add_argument('--gather', action='store_true', help='also all_gather the rendered images each step (RCCL)')
    ap.add_argument('--graph', action='store_true', help='replay the step from a captured HIP graph (measured: no gain, the stream is GPU-bound)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run' % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    from neural_renderer_amd import distributed as nrd
    _, _, dev = nrd.init_from_env()  # one process per GPU; backend "nccl" = RCCL
    dist = None
    if world > 1:
        import torch.distributed as dist

    import neural_renderer_amd as nr
    B, S, ts, eps = args.batch, args.image_size, args.texture_size, 1e-3
    faces, textures = build_scene(dev, B, rank * B, world * B, S, ts)
    F = faces.shape[1]
    faces.requires_grad_(True)
    textures.requires_grad_(True)

    # upstream gradients: dense, g = 2 (image - ref) with a seeded uniform reference
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    with torch.no_grad():
        rgb0, alpha0, depth0 = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)(faces, textures)
        g_rgb = (2 * (rgb0 - torch.rand(rgb0.shape, generator=gen).to(dev))).contiguous()
        g_alpha = (2 * (alpha0 - torch.rand(alpha0.shape, generator=gen).to(dev))).contiguous()
        g_depth = (2 * (depth0 / 100.0 - torch.rand(depth0.shape, generator=gen).to(dev)) / 100.0).contiguous()
        del rgb0, alpha0, depth0

    gather_buf = None
    if args.gather and world > 1:
        gather_buf = torch.empty((world * B, S, S, 3), device=dev)

    def step():
        faces.grad = None
        textures.grad = None
        rgb, alpha, depth = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)(faces, textures)
        if gather_buf is not None:
            dist.all_gather_into_tensor(gather_buf, rgb.detach())
        torch.autograd.backward([rgb, alpha, depth], [g_rgb, g_alpha, g_depth])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Optional: capture the step (~15 launches) once in a HIP graph and replay it.  Measured on MI355X: 0.947 ms
    # replayed vs 0.941 ms eager -- the stream is already GPU-bound, so eager launches stay the default.
    run, mode = step, 'eager'
    if args.graph and gather_buf is None:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            run, mode = graph.replay, 'hipgraph'
        except Exception as ex:  # pragma: no cover
            if rank == 0:
                print('graph capture failed (%s); timing eager launches' % ex, file=sys.stderr)
            run, mode = step, 'eager'
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    eager_ms = None
    if mode == 'hipgraph':  # also report the eager number
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    total_pixels = world * B * S * S
    value = total_pixels / (ms_per_step * 1e-3) / 1e6

    if rank == 0:
        # parity of the benchmarked configuration: gradient of view 0 against the oracle
        from oracle import oracle as O
        O.build()
        ref = O.Rasterize(S, 0.1, 100, eps, (0, 0, 0), True, True, True)
        ref(faces[:1].detach().cpu().numpy(), textures[:1].detach().cpu().numpy())
        r_gf, r_gt = ref.backward(g_rgb[:1].cpu().numpy(), g_alpha[:1].cpu().numpy(), g_depth[:1].cpu().numpy())
        gf, gt = faces.grad[:1].cpu().numpy(), textures.grad[:1].cpu().numpy()
        grad_err = {
            'grad_faces_max_abs_err': float(np.abs(gf - r_gf).max()),
            'grad_faces_max_abs': float(np.abs(r_gf).max()),
            'grad_textures_max_abs_err': float(np.abs(gt - r_gt).max()),
            'grad_textures_max_abs': float(np.abs(r_gt).max()),
            'checked': 'view 0 of rank 0 vs oracle',
        }

        stages = time_stages(faces.detach(), textures.detach(), S, eps, g_rgb, g_alpha, g_depth, args.stage_iters)
        ab = algorithmic_bytes(B, F, S, ts)
        # kernel -> owning stage timing.  The forward visibility stage is two kernels (setup + tiles).
        stage_bytes = {
            'forward_face_index_map': ab['face_setup'] + ab['raster_tiles'],
            'forward_texture_sampling': ab['shade'],
            'backward_pixel_map': ab['backward_pixel_map'],
            'backward_textures': ab['backward_textures'],
            'backward_depth_map': ab['backward_depth_map'],
        }
        dominant = max(stage_bytes, key=lambda k: stages[k])
        achieved = stage_bytes[dominant] / (stages[dominant] * 1e-6) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(dominant, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {
            'bound': 'hbm', 'kernel': STAGE_KERNEL.get(dominant, dominant), 'stage': dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
            'algorithmic_bytes_per_launch': stage_bytes[dominant], 'avg_launch_us': stages[dominant],
            'timing': 'HIP events on the launch stream around the stage call nr_%s (the dominant kernel plus its '
                      'helper launches; profiles/README.md lists the per-kernel rocprofv3 durations they add up from)' % dominant,
            'whole_step': {
                'algorithmic_bytes': sum(stage_bytes.values()),
                'achieved': sum(stage_bytes.values()) / (ms_per_step * 1e-3) / 1e9,
                'frac': sum(stage_bytes.values()) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        e2e = renderer_end_to_end(dev, B, rank * B, world * B, S, ts)
        cpu = None
        if args.cpu_sample_views > 0 and world == 1:
            cpu = cpu_baseline(faces.detach(), textures.detach(), S, eps, g_rgb, g_alpha, g_depth,
                               min(args.cpu_sample_views, B))
        line = {
            'metric': 'rasterize fwd+bwd Mpixels/sec @256x256 batch=64', 'value': value, 'unit': 'Mpixel/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'teapot.obj (2464 faces, fill_back -> %d), %d azimuth views per GPU, raster %dx%d '
                            '(anti_aliasing off), texture_size %d, rgb+alpha+depth forward + backward through the '
                            'Rasterize autograd operator' % (F, B, S, S, ts),
                'views_per_gpu': B, 'image_size': S, 'num_faces': F, 'texture_size': ts, 'eps': eps,
                'parallelism': 'batch-of-views sharded over %d GPU(s), no collective%s'
                               % (world, ' + all_gather(rgb)' if gather_buf is not None else ''),
            },
            'roofline': roofline, 'cpu_baseline': cpu, 'stages_us': stages, 'grad_check': grad_err,
            'launch_mode': mode, 'eager_ms_per_step': eager_ms, 'renderer_end_to_end': e2e,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
