"""Development: per-phase cycle sums of k_face_raster from a -DNR_FWD_PHASES build of the library.
    python -m neural_renderer_amd._build fphases NR_FWD_PHASES=1
    NR_HIP_LIB=neural_renderer_amd/libnr_hip_fphases.so python scripts/fwd_phases.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import neural_renderer_amd as nr
raw = ctypes.CDLL(os.environ['NR_HIP_LIB'])
raw.nr_debug_fwd_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device('cuda', 0)
B, S = int(os.environ.get('B', 64)), int(os.environ.get('S', 256))
faces, textures = bench.build_scene(dev, B, 0, B, S, 2)
fn = nr.Rasterize(S, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
with torch.no_grad():
    fn(faces, textures)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    raw.nr_debug_fwd_phases(buf, 1)
    for _ in range(5):
        fn(faces, textures)
    torch.cuda.synchronize()
raw.nr_debug_fwd_phases(buf, 1)
names = ['0 load, box, queues', '1 inverse, LDS, row scan', '2 row items', '3 row search', '4 pixel scan + items', '5 pixel evaluation']
tot = float(sum(buf[:6])) or 1.0
waves = max(int(buf[7]), 1)
print('B', B, 'waves per launch', waves // 5, 'mean wave life %.2f us (100 MHz clock)' % (buf[6] / 100.0 / waves),
      'mean cycles per wave %.0f' % (tot / waves))
for n, v in zip(names, buf[:6]):
    print('%-26s %14d cycles  %5.1f %%  %8.0f per wave' % (n, v, 100.0 * v / tot, v / waves))
