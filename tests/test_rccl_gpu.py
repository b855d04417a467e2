"""`-m gpu`: the RCCL leg of the multi-GPU path on the ONE GPU a test box has.

torch.distributed's "nccl" backend IS RCCL on ROCm.  RCCL refuses two ranks on one device, so the collectives cannot be
exchanged between ranks here; what CAN be proven is that every call the N > 1 path makes -- `init_process_group(backend=
"nccl", device_id=...)`, `barrier`, `all_reduce(MAX)`, `all_gather_into_tensor`, the padded `all_gather` of uneven shards,
`broadcast`, the `all_reduce(SUM)` of shared parameter gradients -- loads librccl, accepts its arguments and completes on
device tensors, with WORLD_SIZE=1 and the `NR_DIST_FORCE=1` hook of neural_renderer_amd.distributed (which makes a
one-rank group issue the collectives instead of short-cutting them).  The second test runs bench.py's distributed branch
the same way (`--gather` included).  Logs are kept under gpurun_out/ (copied to profiles/ by the round script)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from neural_renderer_amd import distributed as nrd
import neural_renderer_amd as nr

rank, world, dev = nrd.init_from_env()          # backend nccl (= RCCL), device_id=dev
assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1 and dev.type == 'cuda'
log = {'backend': dist.get_backend(), 'device': str(dev), 'rccl_loaded': any('librccl' in l for l in open('/proc/self/maps'))}
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.5
# a rendered shard, gathered: even path (all_gather_into_tensor) and the padded path of uneven shards (all_gather)
faces = torch.tensor([[[[0.8, 0.8, 1.], [0.0, -0.5, 1.], [0.2, -0.4, 1.]]]] * 3, device=dev)
faces[1] *= 0.5
faces[1, :, :, 2] = 1.0
images = nr.rasterize_silhouettes(faces, image_size=32, anti_aliasing=False)
for kwargs in ({'total': 3}, {}, {'force_padded': True}):
    g = nrd.all_gather_images(images, **kwargs)
    assert g.is_cuda and g.shape == images.shape and torch.equal(g, images), kwargs
log['all_gather_images'] = 'even (all_gather_into_tensor), size exchange, padded (all_gather): ok, %%d bytes' %% (images.numel() * 4)
ref = nrd.broadcast_reference_faces(torch.rand(2, 5, 3, 3, device=dev))
assert ref.shape == (5, 3, 3) and ref.is_cuda
p1 = torch.nn.Parameter(torch.ones(7, 3, device=dev)); p1.grad = torch.full_like(p1, 2.0)
p2 = torch.nn.Parameter(torch.ones(4, device=dev))  # no gradient on this rank: contributes zeros, same collective sequence
nrd.all_reduce_shared_grads([p1, p2])
assert torch.equal(p1.grad, torch.full_like(p1, 2.0)) and torch.equal(p2.grad, torch.zeros_like(p2))
torch.cuda.synchronize(dev)
dist.barrier()
dist.destroy_process_group()
print('RCCL_OK ' + json.dumps(log))
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    return dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
                NR_DIST_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')


def _keep(name, text):
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, name), 'w') as f:
            f.write(text)


def test_rccl_collectives_on_one_rank():
    res = subprocess.run([sys.executable, '-c', WORKER % {'root': ROOT}], cwd=ROOT, env=_env(), capture_output=True, text=True,
                         timeout=300)
    _keep('rccl_one_rank.log', '$ WORLD_SIZE=1 NR_DIST_FORCE=1 python -c <tests/test_rccl_gpu.py WORKER>\n' + res.stdout +
          '\n--- stderr ---\n' + res.stderr[-4000:])
    assert res.returncode == 0, res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('RCCL_OK')]
    assert line, res.stdout
    log = json.loads(line[0][len('RCCL_OK '):])
    assert log['backend'] == 'nccl' and log['rccl_loaded']


@pytest.mark.parametrize('gather', [False, True], ids=['no_collective', 'all_gather'])
def test_bench_distributed_branch_over_rccl(gather):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--cpu-sample-views', '0', '--stage-iters', '2', '--light'] + (['--gather'] if gather else [])
    res = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    _keep('rccl_bench_one_rank%s.log' % ('_gather' if gather else ''),
          '$ WORLD_SIZE=1 NR_DIST_FORCE=1 ' + ' '.join(cmd) + '\n' + res.stdout + '\n--- stderr ---\n' + res.stderr[-4000:])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['timing']['backend'] == 'nccl'
    assert ('all_gather' in d['config']['parallelism']) == gather
