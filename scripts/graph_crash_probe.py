"""Development: does a whole-step HIP-graph capture after the operator's graph-replay mode crash (round 3: "host segfault inside
torch's capture"), and does dropping the operator's graphs first avoid it?  Each case in a subprocess; the scene is round 3's
crashing one (BASELINE config 2: 16 teapot views, rgb + alpha + depth, the step of scripts/bench_configs.py).
    python scripts/graph_crash_probe.py [tree]      tree: another checkout of the package to import (default: this one)"""
import os
import subprocess
import sys

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
import neural_renderer_amd as nr
R = sys.modules['neural_renderer_amd.rasterize']
mode = sys.argv[1]
if 'guard' in mode:   # round 3's device guard: torch.cuda.device(dev) around every launch, also on autograd's device thread
    R._on_device = lambda d: torch.cuda.device(d)
if 'stream' in mode:  # round 3's stream query: a torch.cuda.Stream object per call
    R._stream_ptr = lambda d: torch.cuda.current_stream(d).cuda_stream
dev = torch.device('cuda', 0)
faces, textures = bench.build_scene(dev, 16, 0, 16, 256, 2)
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
nr.use_graph_replay(True)
for _ in range(5):
    step()
nr.use_graph_replay(False)
torch.cuda.synchronize()
print('replay mode ran', len(R._GRAPH_CACHE), flush=True)
import inspect
if mode == 'clear':
    if hasattr(R, 'clear_graph_replay_cache'):
        R.clear_graph_replay_cache()
    else:
        R._GRAPH_CACHE.clear()
    rep = nr.graph.capture(step, dev)
elif mode.startswith('bypass'):
    rep = nr.graph.capture(step, dev)
for _ in range(3):
    rep()
torch.cuda.synchronize()
print('OK', float(faces.grad.abs().sum()))
''' % ROOT
for mode in (('bypass', 'clear') if len(sys.argv) < 3 else sys.argv[2].split(',')):
    res = subprocess.run([sys.executable, '-c', CODE, mode], capture_output=True, text=True, timeout=300, cwd=ROOT)
    err = [l for l in res.stderr.strip().splitlines() if 'amdgpu.ids' not in l]
    print(os.path.basename(ROOT), mode, 'rc', res.returncode, res.stdout.strip().replace('\n', ' | ')[-100:], '||', (err[-1][-200:] if err else ''))
