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
        init_group(backend)
    return rank, world


def init_group(backend):
    """init_process_group with this rank BOUND to its GPU when the backend is RCCL ("nccl"): the current device is set and the
    communicator is created for that device (`device_id`), so that no collective -- the barrier of finish() included -- has to
    guess the device from the rank (torch's fallback is rank modulo device count, wrong as soon as LOCAL_RANK / SPINRELAX_DEVICE
    say otherwise)."""
    import torch
    import torch.distributed as dist
    if backend == 'nccl':
        dev = local_device()
        torch.cuda.set_device(dev)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group(backend=backend)


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


# ---- helpers for the drop-in scripts: "am I one of several ranks, which rows are mine, give everybody everything" ----
def _dist_or_none():
    """torch.distributed when a process group is up, None otherwise -- without importing torch for a plain single run."""
    import os
    import sys
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 and 'torch.distributed' not in sys.modules:
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    d = _dist_or_none()
    return d.get_world_size() if d is not None else 1


def rank():
    d = _dist_or_none()
    return d.get_rank() if d is not None else 0


def is_root():
    return rank() == 0


def local_device():
    """GPU of this rank: SPINRELAX_DEVICE if set (rehearsals with several ranks on one GPU), else LOCAL_RANK, else 0."""
    import os
    return int(os.environ.get('SPINRELAX_DEVICE', os.environ.get('LOCAL_RANK', '0')))


def start(backend=None):
    """Called first thing by the drop-in scripts: under torchrun (WORLD_SIZE > 1) join the process group -- RCCL when every
    rank has its own GPU, gloo when SPINRELAX_DIST_BACKEND=gloo or several ranks share a device -- otherwise do nothing.
    Returns (rank, world)."""
    import os
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return 0, 1
    return init_from_env(backend or os.environ.get('SPINRELAX_DIST_BACKEND'))


def finish():
    import shutil
    while _SCRATCH:
        shutil.rmtree(_SCRATCH.pop(), ignore_errors=True)
    d = _dist_or_none()
    if d is not None:
        if d.get_backend() == 'nccl':
            d.barrier(device_ids=[local_device()])
        else:
            d.barrier()
        d.destroy_process_group()


def my_range(n):
    """Rows [i0, i0 + nloc) of this rank out of n (everything when not distributed)."""
    return shard_range(n, rank(), world())


def replicate_sharding(V):
    """True when there are fewer vectors than ranks: C(t) is then sharded over the replicate CHUNKS instead (every rank all
    vectors of its chunk range), the per-replicate values are gathered and the mean / two-pass std over the replicates is
    formed from all of them (calculate-Ct-from-traj.py:226-228; an all-reduce of sums and squares would round dC(t)
    differently) -- SURVEY.md section 8(e), last paragraph."""
    return world() > 1 and V < world()


def my_chunk_range(R):
    """Chunks [r0, r0 + nR) of this rank out of R."""
    return shard_range(R, rank(), world())


def _collective_device():
    import torch
    import torch.distributed as dist
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', local_device())
    return None


def gather_rows(local, n, axis=0):
    """Every rank hands in its my_range(n) slice of an array along `axis` and receives the whole array."""
    if world() == 1:
        return local
    return gather_vector_axis(np.ascontiguousarray(local), n, axis, device=_collective_device())


def gather_chunk_axis(local, R):
    """All-gather of (V, nR, L) raw sums along the chunk axis (my_chunk_range(R) per rank) -> (V, R, L) on every rank."""
    if world() == 1:
        return local
    return gather_vector_axis(np.ascontiguousarray(local), R, 1, device=_collective_device())


_SCRATCH = []


def output_prefix(out_pref):
    """The output prefix a drop-in script should write to: rank 0 the user's, every other rank a private scratch directory
    that finish() removes -- all ranks run the same code down to the writers, one set of reference-named files appears."""
    if is_root():
        return out_pref
    import os
    import tempfile
    d = tempfile.mkdtemp(prefix='spinrelax_rank%d_' % rank())
    _SCRATCH.append(d)
    return os.path.join(d, os.path.basename(out_pref) or 'out')


def relax(ctx, model, D, omega, f_DD, f_CSA, time_fact, gamma_ratio, S2, C, tau, nComps, binvecs=None, weights=None,
          resvecs=None, noe_mode=0, want_J=False, weights_dev_ptr=None, want_stats=False):
    """Context.relax (sr_jomega_relax_f64) over ALL residues with the residues split across the ranks of the process group:
    every argument indexed by residue is sliced to this rank's range, the (E, nRes, ...) results are all-gathered, every
    rank returns the complete arrays.  Global-parameter fits (--opt Diso | Daniso | zeta | CSA: scipy's Powell on a chi^2
    over all residues, spectral_densities.py:1360-1369, calculate-relaxations-from-Ct.py:865-1000) therefore evaluate the
    SAME objective value on every rank and walk the same path in lock-step; the exchange per objective call is this one
    all-gather of E x nRes x 8 doubles (SURVEY.md section 8(e) allows either that or an all-reduce of E partial chi^2
    sums -- the gather keeps the summation order, hence the optimiser's path, independent of the number of ranks).
    Single process: a plain call."""
    kw = dict(binvecs=binvecs, weights=weights, resvecs=resvecs, noe_mode=noe_mode, want_J=want_J, want_stats=want_stats)
    if world() == 1:
        return ctx.relax(model, D, omega, f_DD, f_CSA, time_fact, gamma_ratio, S2, C, tau, nComps, weights_dev_ptr=weights_dev_ptr, **kw)
    if weights_dev_ptr:
        raise ValueError('dist.relax: device-resident weights belong to one rank; pass host weights when several ranks share the residues')
    S2 = np.asarray(S2, dtype=np.float64)
    n = S2.size
    omega = np.atleast_2d(np.asarray(omega, dtype=np.float64))
    E = omega.shape[0]
    i0, nloc = my_range(n)
    sl = slice(i0, i0 + nloc)
    f_CSA = np.broadcast_to(np.asarray(f_CSA, dtype=np.float64), (E, n))
    C = np.atleast_2d(np.asarray(C, dtype=np.float64))
    tau = np.atleast_2d(np.asarray(tau, dtype=np.float64))
    if nloc > 0:
        kw['weights'] = None if weights is None else np.asarray(weights)[sl]
        kw['resvecs'] = None if resvecs is None else np.asarray(resvecs)[sl]
        res = ctx.relax(model, D, omega, f_DD, f_CSA[:, sl], time_fact, gamma_ratio, S2[sl], C[sl], tau[sl],
                        np.asarray(nComps)[sl], **kw)
    else:
        res = (np.empty((E, 0, 4, 2)), np.empty((E, 0, 5, 2)) if want_J else None) + ((np.empty((E, 0, 12)),) if want_stats else ())
    out = [None if r is None else gather_rows(r, n, axis=1) for r in res]
    return tuple(out)
