import os, sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
import torch, bench
import neural_renderer_amd as nr
from neural_renderer_amd import _lib
dev = torch.device('cuda', 0)
faces, textures = bench.build_scene(dev, 64, 0, 64, 256, 2)
lib = _lib.load()
B, F, S = 64, faces.shape[1], 256
with torch.no_grad():
    rgb, alpha, depth = nr.Rasterize(S, 0.1, 100, 1e-3, (0,0,0), True, True, False)(faces, textures)
fi = None
fn = nr.Rasterize(S, 0.1, 100, 1e-4, (0,0,0), False, True, False)
with torch.no_grad():
    _, alpha, _ = fn(faces)
fi = fn.face_index_map
g = (2 * (alpha - torch.rand(alpha.shape, device=dev))).contiguous()
g_rgb = (2 * (rgb - torch.rand(rgb.shape, device=dev))).contiguous()
gf = torch.empty_like(faces)
wsb = lib.nr_backward_workspace_bytes(B, F, S, 1, 1); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(call, n=10):
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for flags, label in ((0, 'default'), (2, 'exact')):
    a_only = lambda: lib.nr_backward_pixel_map(faces.data_ptr(), fi.data_ptr(), None, alpha.data_ptr(), None, g.data_ptr(), gf.data_ptr(), B, F, S, 1e-4, 0, 1, flags, None, ws.data_ptr(), wsb, st)
    r_only = lambda: lib.nr_backward_pixel_map(faces.data_ptr(), fi.data_ptr(), rgb.data_ptr(), None, g_rgb.data_ptr(), None, gf.data_ptr(), B, F, S, 1e-3, 1, 0, flags, None, ws.data_ptr(), wsb, st)
    print(label, 'alpha-only K6 us', round(t(a_only), 1), 'rgb-only', round(t(r_only), 1))
