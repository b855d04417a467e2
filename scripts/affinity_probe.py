"""Development: is torch's hand-over of the backward to its device thread slow on some hosts because the two threads sit on cores
that share no L3?  Host-bound step (bench.host_floor: one view at 16x16) with the process's default affinity and pinned to the
cores that share the L3 of the core it runs on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes

import torch

import bench

_libc = ctypes.CDLL(None)


def getcpu():
    return int(_libc.sched_getcpu())


def l3_group(cpu):
    try:
        txt = open('/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list' % cpu).read().strip()
    except OSError:
        return None
    out = set()
    for part in txt.split(','):
        a, _, b = part.partition('-')
        out.update(range(int(a), int(b or a) + 1))
    return out


dev = torch.device('cuda', 0)
cpu = getcpu()
print('cpus', os.cpu_count(), 'allowed', len(os.sched_getaffinity(0)), 'running on', cpu, 'L3 group', sorted(l3_group(cpu) or []))
print('default affinity ', {k: round(v, 4) for k, v in bench.host_floor(dev, 200).items() if k != 'what'})
grp = l3_group(getcpu())
if grp:
    os.sched_setaffinity(0, grp & os.sched_getaffinity(0))
    print('pinned to L3 group', {k: round(v, 4) for k, v in bench.host_floor(dev, 200).items() if k != 'what'})
    node = set(range(0, 64)) | set(range(128, 192)) if min(grp) < 64 or 128 <= min(grp) < 192 else set(range(64, 128)) | set(range(192, 256))
    os.sched_setaffinity(0, node)
    print('pinned to the NUMA node', {k: round(v, 4) for k, v in bench.host_floor(dev, 200).items() if k != 'what'})
    os.sched_setaffinity(0, {min(grp)})
    print('pinned to one core ', {k: round(v, 4) for k, v in bench.host_floor(dev, 200).items() if k != 'what'})
