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
