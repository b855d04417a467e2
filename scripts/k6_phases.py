"""Development: per-phase cycle sums of k_bpm_fast from a -DNR_K6_PHASES build of the library.
    python -m neural_renderer_amd._build phases NR_K6_PHASES=1
    NR_HIP_LIB=neural_renderer_amd/libnr_hip_phases.so python scripts/k6_phases.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import neural_renderer_amd as nr
from neural_renderer_amd import _lib
lib = _lib.load()
raw = ctypes.CDLL(os.environ['NR_HIP_LIB'])
raw.nr_debug_k6_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device('cuda', 0)
B, S = int(os.environ.get('B', 64)), int(os.environ.get('S', 256))
faces, textures = bench.build_scene(dev, B, 0, B, S, 2)
g_rgb, g_alpha, g_depth = bench.upstream_gradients(faces, textures, S, 1e-3, 1234)
buf = (ctypes.c_ulonglong * 24)()
raw.nr_debug_k6_phases(buf, 1)
st = bench.time_stages(faces, textures, S, 1e-3, g_rgb, g_alpha, g_depth, 5)
torch.cuda.synchronize()
raw.nr_debug_k6_phases(buf, 1)
names = ['0 count/early-exit', '1 staging', '2 face scan', '3 record compaction', '4 line setup', '5 segment scan', '6 sweeps', '7 flush']
tot = float(sum(list(buf)[:8])) or 1.0
print('stage us', {k: round(v, 1) for k, v in st.items() if 'pixel' in k or 'fused_back' in k})
for n, v in zip(names, list(buf)[:8]):
    print('%-22s %14d cycles  %5.1f %%' % (n, v, 100.0 * v / tot))
wn = ['8 classify+scan', '9 barrier wait', '10 fill', '11 decode', '12 U loop', '13 M loop', '14 G loop', '15 flush']
wt = float(sum(list(buf)[8:16])) or 1.0
print('per-wave cycles inside fast_sweeps (all launches of the run):')
for n, v in zip(wn, list(buf)[8:16]):
    print('%-22s %14d cycles  %5.1f %%' % (n, v, 100.0 * v / wt))
print('wave-rounds  U %d  M %d  G %d  idle %d' % tuple(list(buf)[16:20]))
if buf[16]:
    print('cycles per wave-round: U %.0f  M %.0f  G %.0f' % (buf[12] / max(buf[16], 1), buf[13] / max(buf[17], 1), buf[14] / max(buf[18], 1)))
