"""Multi-GPU plumbing: one process per GPU, independent video streams sharded across ranks.

The tracking path has no per-frame exchange: the only shared data are the
read-only weights, broadcast ONCE from rank 0 as a single flat buffer (RCCL over xGMI with
backend "nccl" on ROCm; gloo on CPU for tests).  There is deliberately no all-reduce.
The reference's only multi-GPU inference mechanism is process fan-out via mpiexec
(scripts/test_epochs_usot.py:19-49), which sends no message at all.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


LAST_BROADCAST = {}          # what broadcast_weights last did: backend, bytes, seconds (bench.py: config.weights)


def init(backend=None, device_index=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1).
    backend None = RCCL ("nccl") on a GPU box, gloo on CPU.  device_index: the GPU of this rank
    (default LOCAL_RANK).

    One rank per GPU is the product configuration: when every rank HAS its own GPU (world <= visible devices) the
    weights must travel over RCCL / xGMI - a gloo group there (a mis-set backend, a torch build without RCCL) would still
    "work", through host memory, and hide that the RCCL branch never ran.  That case raises instead of falling back."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'gloo' and torch.cuda.is_available() and world <= torch.cuda.device_count() \
                and os.environ.get('USOT_ALLOW_GLOO_ON_GPUS') != '1':
            raise RuntimeError('streams.init: %d ranks on a box with %d GPUs must broadcast over RCCL (backend "nccl"), not gloo; '
                               'set USOT_ALLOW_GLOO_ON_GPUS=1 to force the host path' % (world, torch.cuda.device_count()))
        if backend == 'nccl':
            torch.cuda.set_device(local if device_index is None else device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if dist.get_backend() != backend:
            raise RuntimeError('streams.init: asked for backend %r, torch.distributed resolved %r' % (backend, dist.get_backend()))
    return rank, local, world


def host_thread_cap(world, quota=None):
    """Host threads one rank may use: the cores this process group is allowed (affinity mask, cgroup CPU quota) divided by the
    ranks that share them - eight ranks spin-polling their result tags under a 16-CPU quota are the one resource streams on
    different GPUs DO share."""
    if quota is None:
        quota = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        try:
            with open('/sys/fs/cgroup/cpu.max') as f:
                q, period = f.read().split()
            if q != 'max':
                quota = min(quota, max(1, int(int(q) / int(period))))
        except Exception:
            pass
    return max(1, int(quota) // max(1, int(world)))


def backend_name():
    if not (dist.is_available() and dist.is_initialized()):
        return 'no'
    b = dist.get_backend()
    return 'RCCL' if b == 'nccl' else b


def broadcast_summary():
    """'RCCL broadcast 117835636 B in 0.0123 s (torch.distributed backend nccl, 8 ranks)' for the bench line."""
    if not LAST_BROADCAST:
        return '%s broadcast 0 B' % backend_name()
    return '%s broadcast %d B in %.4f s (torch.distributed backend %s, %d ranks)' % (
        backend_name(), LAST_BROADCAST['bytes'], LAST_BROADCAST['seconds'], LAST_BROADCAST['backend'], LAST_BROADCAST['world'])


def _collective_device(device):
    """gloo moves host memory; RCCL moves device memory (a caller that names no device, or the CPU, gets this rank's GPU)."""
    if dist.get_backend() == 'gloo':
        return torch.device('cpu')
    if device is None or torch.device(device).type != 'cuda':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device(device)


def broadcast_weights(model, src=0, device=None):
    """One flat float32 broadcast of every parameter and float buffer (29.4 M params +
    BN statistics = 117.8 MB) plus one tiny int64 broadcast for num_batches_tracked."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        LAST_BROADCAST.clear()
        return 0
    import time
    t0 = time.perf_counter()
    sd = model.state_dict()
    fkeys = [k for k, v in sd.items() if v.is_floating_point()]
    ikeys = [k for k, v in sd.items() if not v.is_floating_point()]
    dev = _collective_device(device if device is not None else sd[fkeys[0]].device)
    flat = torch.cat([sd[k].detach().reshape(-1).float().to(dev) for k in fkeys])
    dist.broadcast(flat, src=src)
    ints = torch.stack([sd[k].detach().reshape(()).to(dev) for k in ikeys]) if ikeys else None
    if ints is not None:
        dist.broadcast(ints, src=src)
    out, off = {}, 0
    for k in fkeys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].reshape(sd[k].shape).to(sd[k].device)
        off += n
    for i, k in enumerate(ikeys):
        out[k] = ints[i].to(sd[k].device)
    model.load_state_dict(out, strict=True)
    if flat.is_cuda:
        torch.cuda.synchronize(flat.device)
    LAST_BROADCAST.update(backend=dist.get_backend(), bytes=flat.numel() * 4, seconds=round(time.perf_counter() - t0, 4),
                          world=dist.get_world_size())
    return flat.numel() * 4


def shard(items, rank, world):
    """Stream s runs on rank s mod world."""
    return list(items)[rank::world]


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device or 'cpu'))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # timing scalar only, not on the data path
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
