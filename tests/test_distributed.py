"""CPU coverage of the N > 1 path: world_size-2 gloo processes exercise the batch sharding, the image
all-gather (even and uneven shards) and the shared-parameter gradient all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neural_renderer_amd import distributed as nrd


def test_shard_bounds_partition():
    for total in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [nrd.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _render_stub(view_ids, size=8):
    """Stands in for the HIP rasterizer on CPU: an 'image' that is a deterministic function of the view index
    (the sharding property under test is that a view rendered in a shard equals the view rendered in the full
    batch -- the GPU suite checks that property for the real kernels, tests/test_hip_parity.py)."""
    g = torch.arange(size * size, dtype=torch.float32).reshape(1, 1, size, size)
    return torch.sin(g * 0.1 + view_ids.reshape(-1, 1, 1, 1).float())


def _worker(rank, world, port, total, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    r, w, dev = nrd.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    start, stop = nrd.shard_bounds(total, rank, world)
    local = _render_stub(torch.arange(start, stop))
    full = nrd.all_gather_images(local, total=total)
    ref = _render_stub(torch.arange(total))
    ok_gather = bool(torch.equal(full, ref))
    full2 = nrd.all_gather_images(local)  # sizes discovered with a collective
    ok_gather2 = bool(torch.equal(full2, ref))
    # shared parameter: loss = sum over local views of <param, view-dependent vector>
    p = torch.nn.Parameter(torch.ones(3))
    loss = sum((p * float(v + 1)).sum() for v in range(start, stop)) if stop > start else (p * 0).sum()
    loss.backward()
    nrd.all_reduce_shared_grads([p])
    expect = float(sum(v + 1 for v in range(total)))
    ok_grad = bool(torch.allclose(p.grad, torch.full((3,), expect)))
    # a shared parameter that received no gradient on rank 1 (e.g. the mesh is invisible in that shard): both ranks must
    # still issue the same collectives, and the sum is rank 0's gradient
    q = torch.nn.Parameter(torch.ones(2))
    if rank == 0:
        (q * 3.0).sum().backward()
    nrd.all_reduce_shared_grads([q, p])
    ok_grad = ok_grad and bool(torch.allclose(q.grad, torch.full((2,), 3.0)))
    # Q1 under sharding: every rank ends up with rank 0's first projected view
    faces_local = torch.arange((stop - start) * 4 * 9, dtype=torch.float32).reshape(stop - start, 4, 3, 3) + 1000.0 * rank
    ref_faces = nrd.broadcast_reference_faces(faces_local if stop > start else torch.zeros(1, 4, 3, 3))
    ok_ref = bool(torch.equal(ref_faces, torch.arange(36, dtype=torch.float32).reshape(4, 3, 3)))
    dist.barrier()
    dist.destroy_process_group()
    out_q.put((rank, ok_gather, ok_gather2, ok_grad and ok_ref))


@pytest.mark.parametrize('total', [8, 5])
def test_two_process_gloo(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g1, g2, gr in results:
        assert g1 and g2 and gr, (rank, g1, g2, gr)


def test_pin_to_l3_group_picks_one_cache_group_per_rank():
    """distributed.pin_to_l3_group: one process per GPU pinned to the logical CPUs that share one L3 (rank r: the r-th group);
    every thread of the process follows; None -- and no change -- where the topology cannot be read."""
    import os
    from neural_renderer_amd import distributed as nrd
    before = os.sched_getaffinity(0)
    try:
        groups = nrd.l3_groups()
        got = nrd.pin_to_l3_group(local_rank=1)
        if not groups:
            assert got is None and os.sched_getaffinity(0) == before
            return
        assert got == groups[1 % len(groups)] and got <= before
        assert os.sched_getaffinity(0) == got
        assert all(a.isdisjoint(b) for i, a in enumerate(groups) for b in groups[i + 1:])
        assert set().union(*groups) == before
    finally:
        for tid in os.listdir('/proc/self/task'):
            os.sched_setaffinity(int(tid), before)
