#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations other than the headline one (and of the headline scene per mode),
through the product's Rasterize operator.  One JSON line per row; run on the GPU box:
    python scripts/bench_configs.py > gpurun_out/configs.jsonl
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
import neural_renderer_amd as nr  # noqa: E402

dev = torch.device('cuda', 0)


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(name, faces, textures, S, modes, eps, iters=10, graph=False):
    rgb, alpha, depth = modes
    faces = faces.clone().requires_grad_(True)
    if textures is not None:
        textures = textures.clone().requires_grad_(rgb)
    B, F = faces.shape[:2]
    gen = torch.Generator(device='cpu').manual_seed(7)
    with torch.no_grad():
        outs = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), rgb, alpha, depth)(faces, textures if rgb else None)
        grads = [None if o is None else (2 * (o / (100.0 if k == 2 else 1.0) - torch.rand(o.shape, generator=gen).to(dev))
                                         / (100.0 if k == 2 else 1.0)).contiguous() for k, o in enumerate(outs)]
        cov = float((nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), False, True, False)(faces)[1]).mean())

    def step():
        faces.grad = None
        if textures is not None:
            textures.grad = None
        o = nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), rgb, alpha, depth)(faces, textures if rgb else None)
        torch.autograd.backward([x for x in o if x is not None], [g for g in grads if g is not None])

    def fwd():
        with torch.no_grad():
            nr.Rasterize(S, 0.1, 100, eps, (0, 0, 0), rgb, alpha, depth)(faces, textures if rgb else None)

    ms = timeit(step, iters)
    ms_f = timeit(fwd, iters)
    row = {'config': name, 'B': B, 'F': F, 'S': S, 'ts': 0 if textures is None else textures.shape[2],
           'modes': ''.join(c for c, m in zip('rad', modes) if m), 'coverage': round(cov, 4),
           'ms_fwd_bwd': round(ms, 4), 'ms_fwd': round(ms_f, 4), 'mpixel_s': round(B * S * S / ms / 1e3, 1)}
    if graph:  # host-bound sizes only
        # torch's autograd engine hands the backward of CUDA tensors to a device thread: ~100 us of wake-up and GIL traffic per
        # step (scripts/host_profile.py: 283 -> 153 us of host time); a loop that is bound by the host can switch that off
        with torch.autograd.set_multithreading_enabled(False):
            ms_st = timeit(step, iters)
        row.update(ms_fwd_bwd_single_threaded_autograd=round(ms_st, 4), mpixel_s_single_threaded_autograd=round(B * S * S / ms_st / 1e3, 1))
        # the same step captured as a whole by the caller (neural_renderer_amd.graph.capture): no copies.  (First: a whole-step
        # capture AFTER the operator's replay mode has run in the same process crashes on ROCm 7.2 / torch 2.10.)
        replay = nr.graph.capture(step, dev)
        ms_g = timeit(replay, iters)
        row.update(ms_fwd_bwd_hipgraph=round(ms_g, 4), mpixel_s_hipgraph=round(B * S * S / ms_g / 1e3, 1))
        # the plain API with the operator's graph replay on (nr.use_graph_replay: copies in, replay, copies out)
        nr.use_graph_replay(True)
        try:
            ms_r = timeit(step, iters)
        finally:
            nr.use_graph_replay(False)
            sys.modules['neural_renderer_amd.rasterize'].clear_graph_replay_cache()
        row.update(ms_fwd_bwd_graph_replay=round(ms_r, 4), mpixel_s_graph_replay=round(B * S * S / ms_r / 1e3, 1))
    print(json.dumps(row), flush=True)


def main():
    only = [k for k in os.environ.get('ONLY', '').split(',') if k]  # e.g. ONLY=C2,C4
    # like bench.py: the process on the cores of one L3 group (torch's hand-over of every backward to its device thread costs
    # 8 us there and ~130 us across groups: the host-bound rows -- C2, C3, the shards -- measure the host otherwise); NO_PIN=1: off
    if not os.environ.get('NO_PIN'):
        group = nr.distributed.pin_to_l3_group(0, 0)
        print(json.dumps({'note': 'process pinned to one L3 group' if group else 'L3 topology unreadable: not pinned',
                          'cpus': len(group) if group else None}), flush=True)
    rng = np.random.default_rng(1234)
    if not only or 'H' in only:
        # headline scene, per mode, AA off (S 256) and AA on (S 512)
        for S in (256, 512):
            faces, textures = bench.build_scene(dev, 64, 0, 64, S, 2)
            for name, modes, eps in (('H silhouette', (False, True, False), 1e-4), ('H rgb', (True, False, False), 1e-3),
                                     ('H depth', (False, False, True), 1e-4), ('H rgb+alpha+depth', (True, True, True), 1e-3)):
                run('%s S%d' % (name, S), faces, textures, S, modes, eps)
    if not only or 'SH' in only:
        # the per-GPU shards of the 64-view headline job at 2 / 4 / 8 GPUs (views [0, 64/R) of the 64 azimuths, global view 0 as
        # the texture-depth reference), on this one GPU: bench.py's shard_rows, one row each
        for r in bench.shard_rows(dev, 64, 256, 2, 1e-3, 100):
            print(json.dumps({'config': 'SH shard of H for %d GPUs: %d views' % (r['gpus_this_shard_belongs_to'], r['views']), 'B': r['views'],
                              'S': 256, 'modes': 'rad', 'ms_fwd_bwd': round(r['ms_autograd'], 4),
                              'ms_fwd_bwd_min_max_of_5': [round(x, 4) for x in r['ms_autograd_min_max']],
                              'ms_fwd_bwd_single_threaded_autograd': round(r['ms_autograd_caller_thread'], 4),
                              'ms_fwd_bwd_function_protocol': round(r['ms_function_protocol'], 4),
                              'mpixel_s_function_protocol': round(r['mpixel_per_s_function_protocol'], 1)}), flush=True)
    if not only or 'C2' in only:
        # config 2: 16 azimuth views, RGB + depth + silhouette
        faces, textures = bench.build_scene(dev, 16, 0, 16, 256, 2)
        run('C2 teapot 16 views', faces, textures, 256, (True, True, True), 1e-3, iters=30, graph=True)
    if not only or 'C3' in only:
        # config 3: example2, teapot -> rectangle silhouette loss through the public Renderer (256x256, anti-aliasing on), 300 Adam steps
        sys.path.insert(0, os.path.join(ROOT, 'examples'))
        import make_data
        import example2
        make_data.main()
        data = os.path.join(ROOT, 'examples', 'data')
        def example2_run(caller_thread):
            """one fresh optimisation of 300 steps -> (ms per step, first loss, last loss)"""
            model = example2.Model(os.path.join(data, 'teapot.obj'), os.path.join(data, 'example2_ref.png')).to(dev)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            losses = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.autograd.set_multithreading_enabled(not caller_thread):
                for _ in range(300):
                    opt.zero_grad()
                    loss = model()
                    loss.backward()
                    opt.step()
                    losses.append(loss.detach())
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 300 * 1e3, float(losses[0]), float(losses[-1])

        # Five fresh runs per mode behind one untimed run (the first loop of a process pays the one-time costs: allocator
        # growth, kernel attributes); a host-bound loop jitters with the host's thread wake-ups, so median and spread (round 3
        # reported a single run: 1.28 ms once, 0.55-0.64 otherwise).
        example2_run(False)
        for caller_thread, label in ((False, 'C3 example2 vertex optimisation, 300 Adam steps'),
                                     (True, 'C3 example2, eager with torch.autograd.set_multithreading_enabled(False)')):
            runs = sorted(example2_run(caller_thread) for _ in range(5))
            print(json.dumps({'config': label, 'B': 1, 'S': 512, 'ms_per_step': round(runs[2][0], 4),
                              'ms_per_step_min_max_of_5': [round(runs[0][0], 4), round(runs[-1][0], 4)],
                              'loss_first': round(runs[2][1], 2), 'loss_last': round(runs[2][2], 2)}), flush=True)

        # the same loop with the whole step (render, loss, backward, Adam) captured once in a HIP graph
        model = example2.Model(os.path.join(data, 'teapot.obj'), os.path.join(data, 'example2_ref.png')).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        loss_buf = torch.zeros((), device=dev)

        def train_step():
            opt.zero_grad(set_to_none=False)
            loss = model()
            loss.backward()
            opt.step()
            loss_buf.copy_(loss.detach())

        replay = nr.graph.capture(train_step, dev, warmup=3)  # 3 warm-up + 1 captured step are real Adam steps
        first = float(loss_buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(296):
            replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 296 * 1e3
        print(json.dumps({'config': 'C3 example2, whole step replayed from a HIP graph (Adam capturable)', 'B': 1, 'S': 512,
                          'ms_per_step': round(ms, 4), 'loss_after_4_steps': round(first, 2),
                          'loss_last': round(float(loss_buf), 2)}), flush=True)
    # extreme shapes with the headline's pixel count: one 2048 x 2048 view, 1024 views of 32 x 32 (only on request)
    for key, (xb, xs) in (('X1', (1, 2048)), ('X2', (1024, 32)), ('X3', (4, 1024)), ('X4', (256, 128))):
        if key in only:
            faces, textures = bench.build_scene(dev, xb, 0, xb, xs, 2)
            run('%s teapot %d views %dx%d' % (key, xb, xs, xs), faces, textures, xs, (True, True, True), 1e-3)
            del faces, textures
    from test_hip_parity import icosphere, project_mesh
    if not only or 'C4' in only or 'C5' in only:  # (C5's mesh continues C4's random stream: same scenes whatever is selected)
        # config 4 (per-GPU share): 64 distinct ~5k-face meshes (10 240 with fill_back), ts 4 random textures, 256x256 RGB
        v0, f0 = icosphere(4)
        batch = []
        for _ in range(64):
            v = v0 * (0.55 + 0.12 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
            q = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
            batch.append(project_mesh((v @ q).astype(np.float32), f0, [0.3, 0.4, -2.6]))
    if not only or 'C4' in only:
        faces = torch.from_numpy(np.stack(batch)).to(dev)
        textures = torch.rand((64, faces.shape[1], 4, 4, 4, 3), device=dev)
        run('C4 64 random meshes x 10240 faces ts4', faces, textures, 256, (True, False, False), 1e-3)
        del faces, textures
    if not only or 'C5' in only:
        # config 5: one 327 680-face icosphere (655 360 with fill_back), 1024x1024, ts 8
        v0, f0 = icosphere(7)
        v = v0 * (0.6 + 0.02 * rng.normal(size=(v0.shape[0], 1))).astype(np.float32)
        faces = torch.from_numpy(project_mesh(v.astype(np.float32), f0, [0.0, 0.0, -2.4])[None]).to(dev)
        textures = torch.rand((1, faces.shape[1], 8, 8, 8, 3), device=dev)
        run('C5 655k-face mesh 1024x1024 ts8', faces, textures, 1024, (True, True, True), 1e-3, iters=5)


if __name__ == '__main__':
    main()
