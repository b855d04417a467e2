"""Does it cost GPU time when the forward and the backward of a step are launched from different host threads (torch's
autograd engine runs backward nodes of CUDA tensors on its own device thread)?  Development probe.

A: nr_forward_rasterize + nr_backward_rasterize called back to back from ONE thread, buffers preallocated
B: the same calls, the backward issued by a second thread (handshake through queues)
C: the operator through torch.autograd (bench.py's step)
"""
import os
import queue
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import neural_renderer_amd as nr
from neural_renderer_amd import _lib

dev = torch.device('cuda', 0)
B, S, ts = int(os.environ.get('B', 64)), int(os.environ.get('S', 256)), 2
faces, textures = bench.build_scene(dev, B, 0, B, S, ts)
F = faces.shape[1]
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
fi = torch.empty((B, S, S), dtype=torch.int32, device=dev)
wm = torch.empty((B, S, S, 3), device=dev)
dm = torch.empty((B, S, S), device=dev)
rgb = torch.empty((B, S, S, 3), device=dev)
am = torch.empty((B, S, S), device=dev)
vis = torch.empty((B, F), dtype=torch.uint8, device=dev)
bg = torch.zeros(3, device=dev)
wsf_b = lib.nr_forward_workspace_bytes(B, F, S)
wsf = torch.empty(wsf_b, dtype=torch.uint8, device=dev)
g = [torch.rand((B, S, S, 3), device=dev), torch.rand((B, S, S), device=dev), torch.rand((B, S, S), device=dev)]
gf = torch.empty_like(faces)
gt = torch.empty_like(textures)
wsb_b = lib.nr_backward_workspace_bytes(B, F, S, 1, 1)
wsb = torch.empty(wsb_b, dtype=torch.uint8, device=dev)


wsf.fill_(255)
_epoch = [254]


def fwd():  # as the operator calls it: kept workspace with falling epochs, weights of covered pixels only
    e = _epoch[0]
    _epoch[0] = e - 1 if e > 0 else 254
    if e == 0:
        wsf.fill_(255)
    _lib.check(lib.nr_forward_rasterize(faces.data_ptr(), None, textures.data_ptr(), fi.data_ptr(), wm.data_ptr(), dm.data_ptr(),
                                        rgb.data_ptr(), am.data_ptr(), vis.data_ptr(), bg.data_ptr(), 0, B, F, S, ts, 0.1, 100.0,
                                        1e-3, _lib.NR_FLAG_ZBUF_EPOCH | (e << 8) | _lib.NR_FLAG_SPARSE_WEIGHT_MAP, wsf.data_ptr(),
                                        wsf_b, stream), 'f')


def bwd():
    _lib.check(lib.nr_backward_rasterize(faces.data_ptr(), None, fi.data_ptr(), wm.data_ptr(), dm.data_ptr(), rgb.data_ptr(),
                                         am.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), gf.data_ptr(),
                                         gt.data_ptr(), B, F, S, ts, 1e-3, int(os.environ.get('BWD_FLAGS', 0)), vis.data_ptr(), wsb.data_ptr(), wsb_b, stream), 'b')


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6


def a():
    fwd()
    bwd()


qi, qo = queue.SimpleQueue(), queue.SimpleQueue()


def worker():
    torch.cuda.set_device(dev)
    while True:
        m = qi.get()
        if m is None:
            return
        bwd()
        qo.put(1)


th = threading.Thread(target=worker, daemon=True)
th.start()


def b():
    fwd()
    qi.put(1)
    qo.get()


f_op = faces.clone().requires_grad_(True)
t_op = textures.clone().requires_grad_(True)


def c():
    f_op.grad = None
    t_op.grad = None
    fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    o = fn(f_op, t_op)
    torch.autograd.backward(list(o), g)


def fwd2():
    fwd()
    fwd()


def bwd2():
    bwd()
    bwd()


VARIANTS = (('A one thread, raw calls', a), ('B two threads, raw calls', b), ('C autograd operator', c), ('A again', a),
            ('F two forwards', fwd2), ('G two backwards', bwd2))
only = os.environ.get('PROBE')
for name, fn in VARIANTS:
    if only and not name.startswith(only):
        continue
    gpu, host = timed(fn)
    print('%-28s %.1f us per step on the device, %.1f us host enqueue' % (name, gpu, host), flush=True)
qi.put(None)
