"""
Multi-rank PRODUCT path on the GPU: the drop-in scripts under torchrun -- rank r computes its
contiguous range of vectors / residues (spinrelax_amd.dist), the results are all-gathered, rank 0 writes the files -- must
write byte-identical files to a single-process run.  Rehearsed with the ranks sharing this box's one GPU over gloo
(SPINRELAX_DEVICE=0, SPINRELAX_DIST_BACKEND=gloo); with one GPU per rank the same code runs over RCCL (backend nccl).
Covers the global-parameter fit (--opt Diso: one collective per objective evaluation, all ranks walk Powell's path in
lock-step) and the residue-specific CSA step (residues searched by their owner, then gathered).
"""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden
from spinrelax_amd import synth

pytestmark = pytest.mark.gpu
SCR = os.path.join(ROOT, 'scripts')
PORT = [29611]


def run(script, args, nproc=1):
    env = dict(os.environ, SPINRELAX_DEVICE='0', SPINRELAX_DIST_BACKEND='gloo', GPU_MAX_HW_QUEUES='4')
    if nproc == 1:
        cmd = [sys.executable, os.path.join(SCR, script)]
    else:
        PORT[0] += 1
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
               '--master-port', str(PORT[0]), os.path.join(SCR, script)]
    p = subprocess.run(cmd + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    return p.stdout.decode()


def chain(d, vecfn, expfiles, nproc):
    s = synth.config_shapes(1)
    out = os.path.join(d, 'rotdif')
    quat = ' '.join('%.6f' % x for x in synth.Q_EXT)
    run('calculate-Ct-from-traj.py', ['-s', 'ref.pdb', '-f', vecfn, '--tau', s['tau_memory'], '-o', out, '--vecRot', quat, '--vecHist',
                                      '--binary', '--vecAvg', '--S2', '--Ct'], nproc)
    run('calculate-fitted-Ct.py', ['-f', out + '_Ctint.dat', '-o', out], nproc)
    D = '%g %g' % (synth.DISO, synth.DANI)
    run('calculate-relaxations-from-Ct.py', ['-f', out + '_fittedCt.dat', '-o', out + '-600', '--distfn', out + '_vecHistogram.npz', '-F',
                                             '600.133e6', '--tu', 'ps', '--zeta', synth.ZETA, '--D', D], nproc)
    common = ['-f', out + '_fittedCt.dat', '--distfn', out + '_vecHistogram.npz', '--zeta', synth.ZETA, '-D', synth.DISO, '--aniso', synth.DANI]
    run('calculate-relaxations-multi-field.py', common + ['-o', out + '-optrsCSA', '--opt', 'rsCSA'] + expfiles, nproc)
    run('calculate-relaxations-multi-field.py', common + ['-o', out + '-optDiso', '--opt', 'Diso'] + expfiles, nproc)
    return sorted(f for f in os.listdir(d) if f.startswith('rotdif'))


@pytest.mark.parametrize('nproc', [2, 3])
def test_scripts_under_torchrun_write_the_single_process_files(tmp_path, synth_cache, nproc):
    s = synth.config_shapes(1)
    vecfn = str(tmp_path / 'solute.npz')
    np.savez(vecfn, vecs=synth_cache(1), names=np.arange(2, 34), dt=s['dt'])
    g = golden('cfg1_chain.npz')
    rs = golden('cfg1_rscsa.npz')
    exps = []
    for kind, MHz, vals, errs in zip(rs['expt_kind'], rs['expt_MHz'], rs['expt_vals'], rs['expt_errs']):
        fn = str(tmp_path / ('expt_%s_%d.dat' % (kind, round(float(MHz)))))
        with open(fn, 'w') as fp:
            fp.write('# Type %s\n# NucleiA 15N\n# NucleiB 1H\n# Frequency %.3f\n' % (kind, float(MHz)))
            for nm, v, e in zip(g['names'], vals, errs):
                fp.write('%d %.12g %.12g\n' % (nm, v, e))
        exps.append(fn)
    one, many = str(tmp_path / 'one'), str(tmp_path / 'many')
    os.makedirs(one)
    os.makedirs(many)
    files1 = chain(one, vecfn, exps, 1)
    filesN = chain(many, vecfn, exps, nproc)
    assert files1 == filesN and len(files1) >= 12, (files1, filesN)
    for f in files1:
        if f.endswith('.npz'):
            a, b = np.load(os.path.join(one, f), allow_pickle=True), np.load(os.path.join(many, f), allow_pickle=True)
            for k in ('names', 'data'):
                assert np.array_equal(a[k], b[k]), (f, k)
        else:
            assert filecmp.cmp(os.path.join(one, f), os.path.join(many, f), shallow=False), f
    # nothing but rank 0's files: no scratch directories left behind
    import glob
    import tempfile
    assert not glob.glob(os.path.join(tempfile.gettempdir(), 'spinrelax_rank*'))


def test_more_ranks_than_vectors_shards_the_replicate_chunks(tmp_path, synth_cache):
    """V = 3 vectors on 4 ranks: C(t) is sharded over the replicate chunks (every rank all vectors of its chunk range), the
    raw sums are gathered and the chunk-statistics kernel runs on all of them; the histogram / mean vector / S2 keep the
    vector sharding (one rank idles).  The files must be byte-identical to a single process's -- dC(t) included, which an
    all-reduce of sums and squares would round differently.  Two input files with tails, lab-frame vectors of their own."""
    s = synth.config_shapes(1)
    F = int(s['tau_memory'] / s['dt'])
    fns = []
    for k, (n, seed) in enumerate(((4 * F + 37, 51), (3 * F + 90, 52))):
        fn = str(tmp_path / ('part%d.npz' % k))
        np.savez(fn, vecs=synth.synth_vectors(n, 3, seed=seed), vecs_lab=synth.synth_vectors(n, 3, seed=seed + 10), names=np.array([7, 8, 11]), dt=s['dt'])
        fns.append(fn)
    quat = ' '.join('%.6f' % x for x in synth.Q_EXT)
    outs = {}
    for nproc in (1, 4):
        d = str(tmp_path / ('n%d' % nproc))
        os.makedirs(d)
        out = os.path.join(d, 'rotdif')
        run('calculate-Ct-from-traj.py', ['-s', 'ref.pdb', '-f'] + fns + ['--tau', s['tau_memory'], '-o', out, '--vecRot', quat, '--vecHist', '--binary',
                                                                     '--vecAvg', '--S2', '--Ct'], nproc)
        outs[nproc] = d
    files = sorted(os.listdir(outs[1]))
    assert files == sorted(os.listdir(outs[4])) and len(files) == 5
    for f in files:
        if f.endswith('.npz'):
            a, b = np.load(os.path.join(outs[1], f), allow_pickle=True), np.load(os.path.join(outs[4], f), allow_pickle=True)
            assert np.array_equal(a['data'], b['data']) and np.array_equal(a['names'], b['names'])
        else:
            assert filecmp.cmp(os.path.join(outs[1], f), os.path.join(outs[4], f), shallow=False), f
    assert open(os.path.join(outs[1], 'rotdif_Ctext.dat')).read() != open(os.path.join(outs[1], 'rotdif_Ctint.dat')).read()


def test_rccl_world_size_one_beside_a_live_pipeline():
    """The first RCCL initialisation of the product path must not happen on the 8-GPU node: backend "nccl" through
    spinrelax_amd.dist.init_group (device bound with torch.cuda.set_device + device_id), a world of ONE rank, in the same
    process as a live Context + GroupedPipeline; an all-gather of every group's table queued behind its last launch
    (bench.py's consumer), the product's finish() (barrier with device_ids, destroy).  tests/rccl_world1_worker.py."""
    PORT[0] += 1
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(PORT[0]),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('SPINRELAX_DIST_BACKEND', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_world1_worker.py')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600, env=env)
    assert p.returncode == 0 and b'RCCL_WORLD1_OK' in p.stdout, p.stdout.decode()[-3000:]
