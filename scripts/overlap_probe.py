"""Development: how much of the K7/K8 gathers hides behind K6 when they run on a second stream (stage calls of the C-ABI)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import neural_renderer_amd as nr
from neural_renderer_amd import _lib
lib = _lib.load()
B = int(os.environ.get('B', 64)); S = 256; ts = 2; eps = 1e-3
dev = torch.device('cuda', 0)
faces, textures = bench.build_scene(dev, B, 0, B, S, ts)
g_rgb, g_alpha, g_depth = bench.upstream_gradients(faces, textures, S, eps, 1234)
F = faces.shape[1]
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(dev)
fi = torch.empty((B, S, S), dtype=torch.int32, device=dev); wm = torch.empty((B, S, S, 3), device=dev)
dm = torch.empty((B, S, S), device=dev); rgb = torch.empty((B, S, S, 3), device=dev); am = torch.empty((B, S, S), device=dev)
vis = torch.empty((B, F), dtype=torch.uint8, device=dev); bg = torch.zeros(3, device=dev)
gf = torch.empty_like(faces); gf2 = torch.zeros_like(faces); gt = torch.empty_like(textures)
wsb = lib.nr_forward_workspace_bytes(B, F, S); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
bwsb = lib.nr_backward_workspace_bytes(B, F, S, 1, 1); bws = torch.empty(max(bwsb, 1), dtype=torch.uint8, device=dev)
_lib.check(lib.nr_forward_rasterize(faces.data_ptr(), None, textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(),
                                    rgb.data_ptr(), am.data_ptr(), vis.data_ptr(), bg.data_ptr(), 0, B, F, S, ts, 0.1, 100.0, eps, 0,
                                    ws.data_ptr(), wsb, main.cuda_stream), 'fwd')


def k6(st):
    lib.nr_backward_pixel_map(faces.data_ptr(), fi.data_ptr(), rgb.data_ptr(), am.data_ptr(), g_rgb.data_ptr(), g_alpha.data_ptr(),
                              gf.data_ptr(), B, F, S, eps, 1, 1, 0, vis.data_ptr(), bws.data_ptr(), bwsb, st.cuda_stream)


def gathers(st):
    lib.nr_backward_textures(fi.data_ptr(), None, None, faces.data_ptr(), None, wm.data_ptr(), dm.data_ptr(), g_rgb.data_ptr(),
                             gt.data_ptr(), B, F, S, ts, eps, 0, st.cuda_stream)
    lib.nr_backward_depth_map(faces.data_ptr(), dm.data_ptr(), fi.data_ptr(), None, wm.data_ptr(), g_depth.data_ptr(),
                              gf2.data_ptr(), B, F, S, st.cuda_stream)


def sequential():
    k6(main); gathers(main)


def overlapped():
    ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
    gathers(side)
    k6(main)
    ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(iters):
        fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print(json.dumps({'B': B, 'k6_us': round(timeit(lambda: k6(main)), 1), 'gathers_us': round(timeit(lambda: gathers(main)), 1),
                  'sequential_us': round(timeit(sequential), 1), 'overlapped_us': round(timeit(overlapped), 1)}))
