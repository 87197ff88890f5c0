"""
CPU tests: the C-ABI shared library loads and exports every symbol include/spinrelax_hip.h declares (no
compute calls); the product fails loudly without a GPU (no CPU fallback); the multi-rank path (vector
sharding + one all-gather of results) works with 2 ranks over gloo.
"""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, 'include', 'spinrelax_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(sr_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from spinrelax_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        from spinrelax_amd import build
        build.build(verbose=False)
    names = header_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), 'libspinrelax_hip.so does not export %s' % n
    # the ctypes binding covers exactly the header
    assert sorted(_lib.SIGNATURES.keys()) == names
    bound = _lib.load()
    assert bound.sr_abi_version() == _lib.ABI_VERSION
    assert bound.sr_ct_psum_stride(4096) >= 2049                   # pure host helper, safe without a GPU


def test_build_id_and_flag_stamps(tmp_path, monkeypatch):
    """The library reports the id of the sources / flags it was built from (bench.py ties the committed PMC figures to it), and an
    object whose flags changed is rebuilt: a flag change re-labels sr_core.o, so it must not leave the other objects behind."""
    from spinrelax_amd import _lib, build
    if os.path.isfile(_lib.LIB_PATH) and not os.environ.get('SR_FIT_DEV_FAST'):
        assert _lib.load().sr_build_id().decode() == build.build_id(), 'libspinrelax_hip.so is not the build of these sources'
    stamp = tmp_path / 'x.o.cmd'
    assert not build._same_flags(str(stamp), 'a b')                 # no stamp: rebuild
    stamp.write_text('a b')
    assert build._same_flags(str(stamp), 'a b')
    assert not build._same_flags(str(stamp), 'a b -DX')             # other flags: rebuild
    before = build.build_id()
    monkeypatch.setitem(build.EXTRA, 'sr_fit.hip', build.EXTRA['sr_fit.hip'] + ['-DX'])
    assert build.build_id() != before                               # the flags are part of the id


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from spinrelax_amd.hip import Context, SpinRelaxHipError
    with pytest.raises(SpinRelaxHipError) as ei:
        Context(0)
    assert 'no CPU fallback' in str(ei.value) or 'no HIP device' in str(ei.value)
    from spinrelax_amd import ct as hostct
    with pytest.raises(SpinRelaxHipError):
        hostct.calculate_Ct_from_files([np.zeros((100, 2, 3), np.float32)], 10.0, 500.0)


def test_product_does_not_import_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, 'spinrelax_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith('.py') or fn.endswith('.hip') or fn.endswith('.h'):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'sr_oracle' not in src and 'libsr_oracle' not in src and 'ref_loader' not in src, fn
    for fn in os.listdir(os.path.join(ROOT, 'scripts')):
        if fn.startswith('calculate-'):
            src = open(os.path.join(ROOT, 'scripts', fn)).read()
            assert 'oracle' not in src, fn


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'oracle'))
import torch, torch.distributed as dist
import sr_oracle as o
from spinrelax_amd import synth, dist as srdist
rank, world = srdist.init_from_env(backend='gloo')
assert world == 2
V = {V}
vecs = synth.synth_vectors(1000, V, seed=1)
full_Ct, full_dCt = o.calculate_Ct_Palmer(o.reformat_vecs_by_tau([vecs], 10.0, 1000.0))
v0, nV = srdist.shard_range(V, rank, world)
# each rank runs the path on ITS vectors only (here: the oracle stands in for the GPU kernels)
loc_Ct, loc_dCt = o.calculate_Ct_Palmer(o.reformat_vecs_by_tau([vecs[:, v0:v0 + nV]], 10.0, 1000.0))
g_Ct = srdist.gather_vector_axis(loc_Ct, V, axis=1)
g_dCt = srdist.gather_vector_axis(torch.from_numpy(loc_dCt), V, axis=1).numpy()
hist = np.arange(nV * 6, dtype=np.float64).reshape(nV, 2, 3) + 1000 * rank
g_hist = srdist.gather_vector_axis(hist, V, axis=0)
assert g_Ct.shape == full_Ct.shape and np.array_equal(g_Ct, full_Ct), 'Ct gather mismatch'
assert np.array_equal(g_dCt, full_dCt)
sizes = srdist.shard_sizes(V, world)
assert g_hist.shape == (V, 2, 3) and g_hist[sizes[0], 0, 0] == 1000.0 and g_hist[0, 0, 0] == 0.0
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('V', [32, 33])
def test_two_rank_gloo_shard_and_gather(tmp_path, V):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, V=V))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out)
        assert 'rank %d ok' % r in out


def _relax_worker(rank, world, port, q):
    import os
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    from spinrelax_amd import dist as srdist

    class FakeCtx:                       # stands in for hip.Context.relax: a residue-wise function of its inputs
        def relax(self, model, D, omega, f_DD, f_CSA, time_fact, gamma_ratio, S2, C, tau, nComps, binvecs=None, weights=None,
                  resvecs=None, noe_mode=0, want_J=False, weights_dev_ptr=None, want_stats=False):
            E, n = np.atleast_2d(omega).shape[0], len(S2)
            base = np.asarray(S2)[None, :, None, None] + np.asarray(f_CSA)[:, :, None, None] + np.asarray(weights).sum(axis=1)[None, :, None, None]
            out = base + np.arange(8).reshape(4, 2)[None, None]
            J = out[:, :, :, :1].repeat(5, axis=2)[:, :, :5] if want_J else None
            return (out, J, out.reshape(E, n, 8).repeat(2, axis=2)[:, :, :12]) if want_stats else (out, J)
    n, E = 7, 3
    rng = np.random.default_rng(3)
    S2, fcsa, w = rng.random(n), rng.random((E, n)), rng.random((n, 5))
    C, tau, K = rng.random((n, 2)), rng.random((n, 2)), np.full(n, 2)
    args = (2, [1.0, 2.0], rng.random((E, 5)), np.ones(E), fcsa, np.ones(E), np.ones(E), S2, C, tau, K)
    got = srdist.relax(FakeCtx(), *args, binvecs=np.zeros((5, 3)), weights=w, want_stats=True)
    i0, nloc = srdist.my_range(n)
    rows = srdist.gather_rows(np.arange(n * 2.0).reshape(n, 2)[i0:i0 + nloc], n)
    q.put((rank, [None if g is None else g.copy() for g in got], rows, srdist.world(), srdist.is_root(), srdist.output_prefix('/x/out')))
    srdist.finish()


def test_distributed_relax_and_row_gather_three_ranks():
    """spinrelax_amd.dist.relax / gather_rows with 3 gloo ranks and 7 residues (uneven shards 3/2/2): every rank receives
    exactly what one process computes; only rank 0 keeps the user's output prefix."""
    import multiprocessing as mp
    import numpy as np
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world, port = 3, 29733
    ps = [ctx.Process(target=_relax_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, E = 7, 3
    rng = np.random.default_rng(3)
    S2, fcsa, w = rng.random(n), rng.random((E, n)), rng.random((n, 5))
    base = S2[None, :, None, None] + fcsa[:, :, None, None] + w.sum(axis=1)[None, :, None, None]
    want = base + np.arange(8).reshape(4, 2)[None, None]
    for rank, got, rows, wd, root, pref in res:
        assert wd == 3 and root == (rank == 0)
        assert np.array_equal(got[0], want) and got[1] is None and got[2].shape == (E, n, 12)
        assert np.array_equal(rows, np.arange(n * 2.0).reshape(n, 2))
        assert (pref == '/x/out') == (rank == 0) and pref.endswith('out')


def _chunk_worker(rank, world, port, q):
    import os
    import sys
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sr_oracle as o
    from spinrelax_amd import synth, ct as hostct, dist as srdist
    F, V = 100, 3
    files = [synth.synth_vectors(437, V, seed=41), synth.synth_vectors(290, V, seed=42)]        # 4 + 2 chunks, tails dropped
    assert srdist.replicate_sharding(V) and not srdist.replicate_sharding(world)

    def sums_fn(chunks):                 # the oracle stands in for kernel 1: raw sums per (vector, chunk, lag)
        v4 = np.stack(chunks).astype(np.float64)
        return np.stack([np.einsum('rjv,rjv->vr', *(2 * [np.einsum('rjvk,rjvk->rjv', v4[:, :-d], v4[:, d:])])) for d in range(1, F // 2 + 1)], -1)

    def finalize_fn(sums):               # calculate-Ct-from-traj.py:225-228 on ALL replicates
        n = F - np.arange(1, F // 2 + 1)
        p = 1.5 * sums / n - 0.5                                   # (V, R, L)
        return np.mean(p, axis=1).T, (np.std(p, axis=1) / (np.sqrt(p.shape[1]) - 1.0)).T

    r0, nR = srdist.my_chunk_range(6)
    Ct, dCt = hostct.calculate_Ct_chunk_sharded(files, F, sums_fn=sums_fn, finalize_fn=finalize_fn)
    q.put((rank, r0, nR, Ct, dCt))
    srdist.finish()


def test_replicate_chunk_sharding_four_ranks_three_vectors():
    """Fewer vectors than ranks (V = 3, world = 4; SURVEY.md section 8(e), last paragraph): the ranks own ranges of the six
    replicate chunks (2 / 2 / 1 / 1, cut from two files whose tails are dropped), the per-replicate raw sums are gathered
    along the chunk axis and mean / two-pass std over ALL replicates are formed from them (calculate-Ct-from-traj.py:225-228)
    -- every rank gets, bit for bit, what one process computes from the whole array.  (The oracle stands in for kernel 1
    and for the chunk-statistics kernel; tests/test_gpu_multirank.py runs the same plumbing with the real kernels.)"""
    import multiprocessing as mp
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sr_oracle as o
    from spinrelax_amd import synth
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    world = 4
    ps = [ctx.Process(target=_chunk_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 2), (2, 2), (4, 1), (5, 1)]
    F, V = 100, 3
    files = [synth.synth_vectors(437, V, seed=41), synth.synth_vectors(290, V, seed=42)]
    v4 = o.reformat_vecs_by_tau(files, 1.0, float(F)).astype(np.float64)
    assert v4.shape == (6, F, V, 3)
    # one process, same arithmetic on the whole array
    n = F - np.arange(1, F // 2 + 1)
    sums = np.stack([np.einsum('rjv,rjv->vr', *(2 * [np.einsum('rjvk,rjvk->rjv', v4[:, :-d], v4[:, d:])])) for d in range(1, F // 2 + 1)], -1)
    p = 1.5 * sums / n - 0.5
    Ct1, dCt1 = np.mean(p, axis=1).T, (np.std(p, axis=1) / (np.sqrt(6) - 1.0)).T
    Cr, dCr = o.calculate_Ct_Palmer(v4)
    assert np.max(np.abs(Ct1 - Cr)) < 1e-14 and np.max(np.abs(dCt1 - dCr)) < 1e-14          # and it IS the reference's C(t)
    for rank, r0, nR, Ct, dCt in res:
        assert np.array_equal(Ct, Ct1) and np.array_equal(dCt, dCt1), rank
