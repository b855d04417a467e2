"""`-m gpu`: the N > 1 path of bench.py exactly as the driver launches it (`python -m torch.distributed.run --nnodes=1
--nproc-per-node 2 ... bench.py --gpus 2`), on a ONE-GPU box: both ranks are pinned to device 0 (NR_DIST_DEVICE) and
rendezvous over gloo (RCCL refuses two ranks on one device).  Proves that the sharded scene construction, the barrier /
max-over-ranks timing, the `--gather` all-gather of the rendered shards and the one-JSON-line contract execute; it is not a
scaling measurement.  The log is kept under gpurun_out/ (copied to profiles/ by the round script)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('gather', [False, True], ids=['no_collective', 'all_gather'])
def test_bench_two_ranks_on_one_gpu(gather):
    env = dict(os.environ, NR_DIST_DEVICE='0', NR_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--cpu-sample-views', '0', '--stage-iters', '2', '--light']
    if gather:
        cmd.append('--gather')
    out_dir = os.path.join(ROOT, 'gpurun_out')
    try:
        res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as ex:  # a rank waiting in a collective the other never enters
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, 'two_ranks_one_gpu_TIMEOUT.log'), 'w') as f:
                f.write(' '.join(cmd) + '\n' + str(ex.stdout)[-4000:] + '\n--- stderr ---\n' + str(ex.stderr)[-4000:])
        raise
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'two_ranks_one_gpu%s.log' % ('_gather' if gather else '')), 'w') as f:
            f.write('$ NR_DIST_DEVICE=0 NR_DIST_BACKEND=gloo ' + ' '.join(cmd) + '\n' + res.stdout + '\n--- stderr ---\n' +
                    res.stderr[-4000:])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    # the 64-view batch is SPLIT (BASELINE.md "Multi-GPU rows": B / R views per rank); 64 per GPU is the weak_scaling object
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['steps'] == 2 and d['scaling'] == 'strong'
    assert d['config']['views_total'] == 64 and d['config']['views_per_gpu'] == 64 // 2
    assert abs(d['value'] - 64 * 256 * 256 / (d['ms_per_step'] * 1e-3) / 1e6) <= 1e-6 * d['value']
    w = d['weak_scaling']
    assert w['scaling'] == 'weak' and w['views_per_gpu'] == 64 and w['views_total'] == 128 and w['value'] > 0
    assert ('all_gather' in d['config']['parallelism']) == gather
    assert d['cpu_baseline'] is None  # (--cpu-sample-views 0 in this test; an N > 1 run otherwise carries rank 0's CPU rows)
