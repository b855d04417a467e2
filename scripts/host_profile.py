"""Host-side cost of one Rasterize forward + backward at the headline size (development helper): cProfile over N steps,
the device kept busy enough that no call blocks.  python scripts/host_profile.py [N]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import neural_renderer_amd as nr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda', 0)
B = int(os.environ.get('B', 64))
faces, textures = bench.build_scene(dev, B, 0, B, 256, 2)
faces.requires_grad_(True)
textures.requires_grad_(True)
g = [torch.rand((B, 256, 256, 3), device=dev), torch.rand((B, 256, 256), device=dev), torch.rand((B, 256, 256), device=dev)]


def step():
    faces.grad = None
    textures.grad = None
    fn = nr.Rasterize(256, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    rgb, alpha, depth = fn(faces, textures)
    torch.autograd.backward([rgb, alpha, depth], g)


for _ in range(5):
    step()
torch.cuda.synchronize()
# host time alone: enqueue without waiting (the queue absorbs N small steps?  no: it back-pressures; so time a few steps only)
t0 = time.perf_counter()
for _ in range(3):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('3 steps: host enqueue %.1f us per step, until idle %.1f us per step' % ((t1 - t0) / 3 * 1e6, (t2 - t0) / 3 * 1e6))
pr = cProfile.Profile()
# single-threaded backward: the Python backward of the operator then runs on this thread and shows up in the profile
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(3):
        step()
    pr.enable()
    for _ in range(N):
        step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
