import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
import torch, bench
import neural_renderer_amd as nr
dev = torch.device('cuda', 0)
faces, textures = bench.build_scene(dev, 16, 0, 16, 256, 2)
faces = faces.clone().requires_grad_(True); textures = textures.clone().requires_grad_(True)
with torch.no_grad():
    outs = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
grads = [torch.rand_like(o) for o in outs]
def step():
    faces.grad = None; textures.grad = None
    o = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(faces, textures)
    torch.autograd.backward(list(o), grads)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager', timeit(step), flush=True)
order = os.environ.get('ORDER', 'replay_first')
if order == 'replay_first':
    nr.use_graph_replay(True); print('replay', timeit(step), flush=True); nr.use_graph_replay(False)
    if os.environ.get('CLEAR'):
        sys.modules['neural_renderer_amd.rasterize']._GRAPH_CACHE.clear(); import gc; gc.collect(); torch.cuda.synchronize()
    print('eager again', timeit(step), flush=True)
    r = nr.graph.capture(step, dev); print('captured', flush=True); print('whole-step graph', timeit(r), flush=True)
else:
    r = nr.graph.capture(step, dev); print('whole-step graph', timeit(r), flush=True)
    nr.use_graph_replay(True); print('replay', timeit(step), flush=True); nr.use_graph_replay(False)
