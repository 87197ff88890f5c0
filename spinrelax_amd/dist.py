"""
Multi-GPU sharding of the hot path: every output of the path is independent per
vector / residue, so rank r owns a contiguous vector range and runs the whole pipeline on it with no
data-path collective; the only exchange is ONE all-gather of the per-shard results at the end (RCCL over
xGMI with backend "nccl"; "gloo" for the CPU tests).  Messages are <= 10 MB per rank, so latency, not link
bandwidth, matters: one collective per result array, never a hand-rolled ring.
"""
import numpy as np


def shard_range(V, rank, world):
    """Contiguous, balanced vector range [v0, v0 + nV) of `rank`: the first V % world ranks get one more."""
    base, extra = divmod(V, world)
    nV = base + (1 if rank < extra else 0)
    v0 = rank * base + min(rank, extra)
    return v0, nV


def shard_sizes(V, world):
    return [shard_range(V, r, world)[1] for r in range(world)]


def init_from_env(backend=None):
    """torch.distributed initialisation from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend=backend)
    return rank, world


def gather_vector_axis(local, V, axis, device=None):
    """All-gather shards of an array that is split along `axis` by shard_range.  `local` is a numpy array or a
    torch tensor; every rank receives the full array (numpy in, numpy out; tensor in, tensor out).
    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = shard_sizes(V, world)
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local
    if device is not None:
        t = t.to(device)
    t = t.movedim(axis, 0).contiguous()
    assert t.shape[0] == sizes[dist.get_rank()], (t.shape, sizes, dist.get_rank())
    nmax = max(sizes)
    if t.shape[0] < nmax:
        pad = torch.zeros((nmax - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    full = torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0).movedim(0, axis)
    if is_np:
        return full.cpu().numpy()
    return full
