"""
GPU tests for SURVEY.md section 8(f)-2: the difference-quaternion lag-correlation kernel (csrc/sr_dq.hip) through the C
ABI against the oracle and the fixtures of the reference's own reductions (tests/golden/dq*_dq.npz), the whole analysis
(spinrelax_amd.dq_distribution.analyse + fit_and_write) against the reference-written files, the drop-in CLI, and
size-independent properties at a trajectory length the oracle would need minutes for.
"""
import filecmp
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, files_equal_numeric, golden
import sr_oracle as o
from spinrelax_amd import synth
from spinrelax_amd import dq_distribution as dq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from spinrelax_amd.hip import Context
    c = Context(0)
    yield c
    c.close()


def _q_of(tag, g):
    q = synth.synth_orientation(int(g['nframes']), 11) if tag == 'dqA' else g['q32']
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(g['q_sha'])
    return q


def test_dq_moments_float64_quaternions_keep_their_precision(ctx):
    """The reference's gmx-rotmat route holds float64 quaternions (rotmatrix_to_quaternion, calculate-dq-distribution.py:
    406-423) and reduces them in float64.  sr_dq_moments_f64 does the same: against the reference's reductions on float64
    input (tests/golden/dqC_dq.npz) to 1e-12 -- while the same data pushed through float32 first is off by ~1e-5 at the
    shortest lags (|v| ~ 1e-2: the deviation the advisor pointed at)."""
    g = golden('dqC_dq.npz')
    q64, lags, nch = g['q64'], g['lags'], int(g['num_chunk'])
    assert q64.dtype == np.float64
    m = ctx.dq_moments(q64, lags, nch)
    tot = m.sum(axis=1)
    scale = np.abs(g['moi']).max(axis=(1, 2))[:, None, None]
    assert np.max(np.abs(dq.moments_to_tensor(tot) - g['moi']) / scale) < 1e-12
    assert np.max(np.abs(np.moveaxis(dq.moments_to_tensor(m), 1, 0) - g['chunk_moi']) / scale[None]) < 1e-12
    assert np.max(np.abs(dq.average_LegendreP1quat(tot) / g['iso'] - 1)) < 1e-12
    assert np.max(np.abs(dq.average_LegendreP1quat(m).T / g['chunk_iso'] - 1)) < 1e-12
    # analyse() keeps float64 data in float64 (time column + quaternions, as rotmatrix_to_quaternion returns them)
    data = np.concatenate((np.arange(q64.shape[0], dtype=np.float64)[None] * 10.0, q64.T))
    res = dq.analyse(data, min_dt=10.0, max_dt=380.0, skip_dt=30.0, num_chunk=nch, ctx=ctx)
    direct = dq.average_LegendreP1quat(ctx.dq_moments(q64, res['lags'], 1).sum(axis=1))
    assert len(res['lags']) >= 10 and np.max(np.abs(res['out_isolist'] / direct - 1)) < 1e-13
    res32 = dq.analyse(data.astype(np.float32), min_dt=10.0, max_dt=380.0, skip_dt=30.0, num_chunk=nch, ctx=ctx)
    assert np.max(np.abs(res32['out_isolist'] / direct - 1)) > 1e-9          # float32 data stay float32 (PLUMED's precision)
    # the float32 round trip is visibly worse at short lags -- which is why the float64 entry point exists
    m32 = ctx.dq_moments(q64.astype(np.float32), lags, nch).sum(axis=1)
    err32 = np.abs(dq.moments_to_tensor(m32) - g['moi']) / scale
    assert err32.max() > 1e-10          # (the fixture quaternions sit 1e-9 off the float32 grid)


@pytest.mark.parametrize('tag', ['dqA', 'dqB'])
def test_dq_moments_vs_reference_and_oracle(ctx, tag):
    g = golden('%s_dq.npz' % tag)
    q = _q_of(tag, g)
    nch = int(g['num_chunk'])
    lags = g['lags']
    m = ctx.dq_moments(q, lags, nch)
    assert m.shape == (len(lags), nch, 7)
    # counts: integer-exact chunk sizes of average_*_chunk
    for k, d in enumerate(lags):
        assert [int(x) for x in m[k, :, 6]] == [b - a for a, b in o.dq_chunk_ranges(int(g['nframes']) - int(d), nch)]
    tot = m.sum(axis=1)
    # reference tensors (float64 evaluation of the reference's functions on the float32 quaternions): 1e-12 relative to
    # the tensor's scale; the north-star bar is 1e-6
    scale = np.abs(g['moi']).max(axis=(1, 2))[:, None, None]
    assert np.max(np.abs(dq.moments_to_tensor(tot) - g['moi']) / scale) < 1e-12
    ch = np.moveaxis(dq.moments_to_tensor(m), 1, 0)
    assert np.max(np.abs(ch - g['chunk_moi']) / scale[None]) < 1e-12
    assert np.max(np.abs(dq.average_LegendreP1quat(tot) / g['iso'] - 1)) < 1e-12
    assert np.max(np.abs(dq.average_LegendreP1quat(m).T / g['chunk_iso'] - 1)) < 1e-12
    assert np.max(np.abs(dq.average_anisotropic_tensor(tot, g['q_frame']) - g['moiR']) / scale) < 1e-12
    # oracle on a few lags, including nchunk = 1
    sub = lags[::7]
    mo = o.dq_moments(q, sub, nch)
    md = ctx.dq_moments(q, sub, nch)
    assert np.max(np.abs(md - mo) / np.maximum(np.abs(mo), 1e-30)[..., :1].max()) < 1e-12
    m1 = ctx.dq_moments(q, sub, 1)
    assert np.allclose(m1[:, 0], md.sum(axis=1), rtol=1e-13, atol=0)


@pytest.mark.filterwarnings('ignore::DeprecationWarning')
def test_analysis_writes_the_reference_files(ctx, tmp_path):
    """vectors of quaternions -> device moments -> eigen-frames, Powell fits, writers: -aniso2.dat byte-identical to the file
    the reference's writers produced from the reference's reductions (same frame quaternion, same decay times)."""
    g = golden('dqA_dq.npz')
    q = _q_of('dqA', g)
    n = q.shape[0]
    data = np.asfortranarray(np.vstack([(np.arange(n) * synth.DT_PS).astype(np.float32), q.T.astype(np.float32)]))
    res = dq.analyse(data, min_dt=100.0, max_dt=5000.0, skip_dt=100.0, num_chunk=4, bDoIso=True, bDoAniso=True, ctx=ctx)
    assert np.array_equal(res['lags'], g['lags'])
    assert np.allclose(res['q_frame'], g['q_frame'], atol=1e-12)
    a2 = np.stack([1 - 2 * g['moiR'][:, i, i] for i in range(3)])
    assert np.max(np.abs(res['out_aniso2list'] - a2)) < 1e-14
    pref = str(tmp_path / 'o')
    fitted = dq.fit_and_write(res, pref, num_chunk=4, bDoIso=False, bDoAniso=True)
    assert np.max(np.abs(fitted['aniso_taus'] / g['taus'] - 1)) < 1e-9
    assert filecmp.cmp(pref + '-aniso2.dat', os.path.join(GOLD, 'dqA-aniso2.dat'), shallow=False)
    # -aniso_q.dat: first line carries the PAF quaternion run-all.bash:393 reads
    first = open(pref + '-aniso_q.dat').readline().split()
    assert np.allclose([float(x) for x in first[1:5]], g['q_frame'], atol=1e-5) and float(first[0]) == 100.0
    assert open(pref + '-moi.xyz').readline().strip() == '3'


@pytest.mark.filterwarnings('ignore::DeprecationWarning')
def test_cli_dropin_on_plumed_file(tmp_path):
    """run-all.bash:383-387 argument vector (without --iso: the reference's isotropic list makes its own fit raise for
    this file, tests/test_dq_hostlogic.py) on the PLUMED file of the de-tumbling fixture."""
    g = golden('dqB_dq.npz')
    out = str(tmp_path / 'rotdif')
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'calculate-dq-distribution.py'), '--aniso', '-f',
           os.path.join(GOLD, 'cfg1_colvar-qorient'), '-o', out, '--mindt', '10', '--skip', '10', '--maxdt', '400', '--num_chunk', '3']
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()
    # same text up to a printed last digit on a rounding boundary (one of ~700 numbers differs in its 7th digit when the
    # moments are summed in another order: 7.485824e+06 / 7.485825e+06)
    ok, why = files_equal_numeric(out + '-aniso2.dat', os.path.join(GOLD, 'dqB-aniso2.dat'), rtol=1e-6)
    assert ok, why
    q_line = [float(x) for x in open(out + '-aniso_q.dat').readline().split()]
    assert np.allclose(q_line[1:], g['q_frame'], atol=1e-5)
    # with --iso the mirror fails where the reference fails (ValueError: math domain error), not silently
    p = subprocess.run(cmd + ['--iso'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode != 0 and 'math domain error' in p.stdout.decode()


def test_dq_large_trajectory_properties(ctx):
    """10^6 quaternions x 100 lags x 5 chunks (1e8 sample pairs; minutes for the numpy oracle): chunk sums add up to
    the single-chunk run, the moments obey |v|^2 <= 1 and Cauchy-Schwarz, two spot lags equal the oracle."""
    q = synth.synth_orientation(1000000, 12)
    lags = np.arange(10, 1001, 10)
    m5 = ctx.dq_moments(q, lags, 5)
    m1 = ctx.dq_moments(q, lags, 1)
    assert np.array_equal(m5[..., 6].sum(axis=1), 1000000 - lags) and np.array_equal(m1[:, 0, 6], 1000000 - lags)
    assert np.allclose(m5.sum(axis=1), m1[:, 0], rtol=1e-12, atol=0)
    tr = (m1[:, 0, 0] + m1[:, 0, 1] + m1[:, 0, 2]) / m1[:, 0, 6]
    assert np.all(tr > 0) and np.all(tr < 1) and np.all(np.diff(tr) > 0)        # rotational diffusion: <|v|^2> grows with the lag
    assert np.all(m1[:, 0, 3] ** 2 <= m1[:, 0, 0] * m1[:, 0, 1])
    for k in (0, 57):
        mo = o.dq_moments(q, lags[k:k + 1], 5)
        assert np.max(np.abs(m5[k] - mo[0]) / np.abs(mo[0]).max()) < 1e-12
