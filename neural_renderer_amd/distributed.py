"""Multi-GPU support for the rasterizer: the batch-of-views dimension shards over the GPUs of a node.

The reference has no multi-GPU path at all.  Every kernel of the hot path reads and writes
only its own batch element (rasterize.py:287, 386-390, 533-536, 770-772, 814-821), so the forward and backward
of a shard need NO communication: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI),
each rendering views [start, stop) of the global batch.  Two optional collectives live outside the rasterizer:

  all_gather_images   when the downstream loss needs the whole batch of rendered images (one all-gather of
                      [B/R, C, H, W] per rank; 6.3 MB per rank for the headline RGB case);
  all_reduce_shared_grads   when mesh parameters are shared by all views (reference mesh.py:29-34 broadcasts
                      them), their gradients are summed over ranks -- a tiny [Nv,3] / [F,ts^3,3] all-reduce.

SURVEY quirk Q1 under sharding: the reference samples textures with the vertex depths of *batch element 0*
(rasterize.py:389).  A shard's element 0 is not the global one, so a sharded run with per-view cameras and non-uniform
textures (BASELINE configs 2 and 4) would differ from the unsharded run.  `broadcast_reference_faces` ships rank 0's
first projected view to every rank once per step (F*36 bytes: 177 KB for the teapot); passed as `Rasterize.faces_z_ref` /
`Renderer.faces_z_ref` / `rasterize(..., faces_z_ref=)` it makes sharded and unsharded images, texture gradients AND vertex
gradients identical bit for bit: every call size takes the same K6 band kernel (k_bpm_row), whose per-record sums do not depend
on what else is in the launch (tests/test_sharding_gpu.py; at the metric's shape -- 64 views vs 2 x 32 vs 8 x 8 --
tests/test_full_size_gpu.py::test_headline_batch_equals_its_shards.  With NR_FLAG_K6_LEGACY, k_bpm_fast's float run sums are
grouped by the arrival order of its line records: <= 1.2e-5 of the largest gradient between two calls, sharded or not).  With
fix_batch_z (NR_FIX_TEXTURE_BATCH_Z=1) nothing needs to be exchanged.
"""
import os

import torch
import torch.distributed as dist


def _force():
    """NR_DIST_FORCE=1 (test hook): initialise the process group and issue the collectives even for ONE rank, so that the
    RCCL code path -- init_process_group(device_id=), barrier, all_reduce, all_gather_into_tensor / all_gather, broadcast on
    device tensors -- can be executed on a one-GPU box (tests/test_rccl_gpu.py)."""
    return os.environ.get('NR_DIST_FORCE', '0') not in ('', '0')


def _active():
    """True when collectives have to be issued: an initialised group of more than one rank (or a forced single rank)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force())


def env_rank_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment. Returns (rank, world, device)."""
    rank, world, local_rank = env_rank_world()
    use_cuda = torch.cuda.is_available()
    # test hooks: NR_DIST_DEVICE pins every rank to one device and NR_DIST_BACKEND picks the backend, so that the
    # multi-process control flow can be exercised on a single-GPU box (RCCL refuses two ranks on one device)
    if 'NR_DIST_DEVICE' in os.environ:
        local_rank = int(os.environ['NR_DIST_DEVICE'])
    backend = backend or os.environ.get('NR_DIST_BACKEND')
    device = torch.device('cuda', local_rank) if use_cuda else torch.device('cpu')
    if use_cuda:
        torch.cuda.set_device(device)
    if (world > 1 or _force()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        kwargs = {}
        if use_cuda and (backend or 'nccl') == 'nccl':
            kwargs['device_id'] = device
        dist.init_process_group(backend or ('nccl' if use_cuda else 'gloo'), rank=rank, world_size=world, **kwargs)
    return rank, world, device


def _cpu_list(text):
    out = set()
    for part in text.strip().split(','):
        if part:
            a, _, b = part.partition('-')
            out.update(range(int(a), int(b or a) + 1))
    return out


def l3_groups(cpus=None):
    """The sets of logical CPUs that share an L3 cache (a CCX on EPYC hosts), restricted to `cpus` (default: the CPUs this
    process may run on), in ascending order of their first CPU.  [] where the topology cannot be read."""
    allowed = set(os.sched_getaffinity(0)) if cpus is None else set(cpus)
    groups, seen = [], set()
    for cpu in sorted(allowed):
        if cpu in seen:
            continue
        try:
            with open('/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list' % cpu) as f:
                grp = _cpu_list(f.read()) & allowed
        except OSError:
            return []
        if grp:
            groups.append(grp)
            seen |= grp
    return groups


def gpu_numa_cpus(device_index):
    """Logical CPUs of the NUMA node GPU `device_index` hangs off (sysfs), or None when that cannot be told."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open('/sys/bus/pci/devices/%s/numa_node' % bdf) as f:
            node = int(f.read())
        if node < 0:
            return None
        with open('/sys/devices/system/node/node%d/cpulist' % node) as f:
            return _cpu_list(f.read())
    except Exception:
        return None


def pin_to_l3_group(local_rank=0, device_index=None):
    """Pin this process (every thread it has now; later ones inherit) to the cores of ONE L3 group -- on the GPU's NUMA node when
    that is known, rank r of a node taking the r-th group.  One process per GPU is the deployment model here, and where its
    threads run matters to a host-bound loop: torch's autograd hands every backward to a device thread and waits for it, two
    wake-ups per step, and on a two-socket EPYC host with 16 L3 groups those cost 8 us when the two threads share an L3 and
    ~130 us when they do not (`scripts/affinity_probe.py`: a one-view 16 x 16 step 0.235 -> 0.107 ms; pinning to the whole NUMA
    node does not help, to a single core neither).  Returns the CPU set, or None when nothing was changed (topology unreadable)."""
    allowed = set(os.sched_getaffinity(0))
    near = gpu_numa_cpus(device_index) if device_index is not None else None
    groups = l3_groups(allowed & near) if near and (allowed & near) else []
    if not groups:
        groups = l3_groups(allowed)
    if not groups:
        return None
    group = groups[int(local_rank) % len(groups)]
    try:
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), group)
            except OSError:
                pass
    except OSError:
        os.sched_setaffinity(0, group)
    return group


def shard_bounds(total, rank, world):
    """Contiguous, balanced partition of `total` views: rank r owns [start, stop)."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard(tensor, rank, world, dim=0):
    """Rank `rank`'s slice of `tensor` along `dim`.  With fewer views than ranks the trailing ranks get an EMPTY slice:
    the rasterizer needs at least one view per call (NR_E_SIZE), so such ranks must skip the render."""
    start, stop = shard_bounds(tensor.shape[dim], rank, world)
    return tensor.narrow(dim, start, stop - start)


def broadcast_reference_faces(projected_faces, src=0):
    """[F,3,3] faces of the GLOBAL batch element 0 on every rank: rank `src` (the owner of view 0) contributes
    `projected_faces[0]` of its shard [b,F,3,3] (the rasterizer's input, i.e. after look_at / perspective /
    vertices_to_faces).  One tiny broadcast; not differentiable (the reference back-propagates nothing through these
    depths either: K7 treats the sampling weights as constants)."""
    ref = projected_faces[0].detach().contiguous().clone()
    if _active():
        dist.broadcast(ref, src=src)
    return ref


def all_gather_images(images, total=None, force_padded=False):
    """Gather per-rank image shards [b_r, ...] into the full batch [B, ...] on every rank (not differentiable
    through the collective; use it on detached images or gather the per-view losses instead).  `force_padded` (tests): take
    the pad / all_gather / trim path of uneven shards even when the shards are equal."""
    if not _active():
        return images
    world, rank = dist.get_world_size(), dist.get_rank()
    images = images.contiguous()
    if images.is_cuda and dist.get_backend() == 'gloo':
        # gloo (CPU rendezvous, e.g. several ranks sharing one GPU in a test) gathers host tensors: stage through the host
        return all_gather_images(images.cpu(), total).to(images.device)
    if total is None:
        n = torch.tensor([images.shape[0]], device=images.device, dtype=torch.int64)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1 and not force_padded:
        out = images.new_empty((sum(sizes),) + tuple(images.shape[1:]))
        dist.all_gather_into_tensor(out, images)
        return out
    # uneven shards: pad to the largest, gather, trim
    m = max(sizes)
    padded = images.new_zeros((m,) + tuple(images.shape[1:]))
    padded[:images.shape[0]] = images
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def all_reduce_shared_grads(parameters):
    """Sum the gradients of parameters shared by all views (vertices / textures of one mesh) over the ranks."""
    if not _active():
        return
    # every rank must issue the same sequence of collectives: a parameter without a gradient on this rank (an unused
    # output, a shard in which the mesh is not visible) contributes zeros instead of being skipped
    for p in parameters:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
