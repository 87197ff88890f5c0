"""
GPU tests of the trajectory front end (csrc/sr_traj.hip): raw coordinates ->
unit X-H vectors (bit-identical to the reference's numpy expression, fixture from the reference's vecnorm_NDarray) and the
per-frame least-squares superposition (against an independent float64 SVD-Kabsch in the oracle; MDTraj, whose superpose
the reference calls, is absent from the image), then through the drop-in script down to C(t).
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden
import sr_oracle as o
from spinrelax_amd import synth
from spinrelax_amd import ct as hostct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from spinrelax_amd.hip import Context
    c = Context(0)
    yield c
    c.close()


def test_xh_vectors_bit_identical_to_reference_expression(ctx):
    g = golden('frontend_xh.npz')
    d = synth.synth_coordinates(int(g['nframes']), int(g['nvec']), int(g['seed']))
    assert hashlib.sha256(d['xyz'].tobytes()).hexdigest() == str(g['xyz_sha'])
    lab = hostct.obtain_XHvecs(d['xyz'], g['indexX'], g['indexH'], ctx=ctx, bSuppressPrint=True)
    assert lab.dtype == np.float32 and lab.shape == g['vecXH'].shape
    assert np.array_equal(lab.view(np.uint32), g['vecXH'].view(np.uint32))        # float32 bits, incl. the 0/0 -> 0 bond
    assert np.array_equal(lab, o.obtain_XHvecs(d['xyz'], g['indexX'], g['indexH']))
    with pytest.raises(SystemExit):
        hostct.obtain_XHvecs(d['xyz'], g['indexX'][:3], g['indexH'], ctx=ctx, bSuppressPrint=True)


@pytest.mark.parametrize('nframes,nvec,seed', [(400, 16, 21), (3000, 64, 22)])
def test_superposition_vs_svd_kabsch(ctx, nframes, nvec, seed):
    d = synth.synth_coordinates(nframes, nvec, seed)
    lab, fitv, quat = hostct.superpose_XHvecs(d['xyz'], d['ref_xyz'], d['fit_indices'], d['indexX'], d['indexH'], ctx=ctx,
                                              want_quat=True)
    want, R = o.superposed_XHvecs(d['xyz'], d['ref_xyz'], d['fit_indices'], d['indexX'], d['indexH'])
    assert fitv.dtype == np.float32
    # float32 output of a float64 computation: half an ulp of a unit-vector component, 6e-8; bar 1e-6 (north star)
    assert np.max(np.abs(fitv - want)) < 2e-7
    # the rotation itself: device quaternion (closed form, Jacobi) vs SVD, as matrices
    w, x, y, z = quat.T
    Rq = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                   np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                   np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    assert np.max(np.abs(Rq - R)) < 1e-12
    assert np.all(quat[:, 0] >= 0) and np.max(np.abs((quat ** 2).sum(1) - 1)) < 1e-14
    # it undoes the tumbling: fitted vectors == the body-frame vectors the trajectory was built from, up to the jitter
    assert np.max(np.abs(fitv - d['body'])) < 5e-3
    assert np.array_equal(lab, o.obtain_XHvecs(d['xyz'], d['indexX'], d['indexH']))
    # identity: a frame equal to the reference needs no rotation
    one = np.repeat(d['ref_xyz'][None], 3, axis=0)
    _, f1, q1 = hostct.superpose_XHvecs(one, d['ref_xyz'], d['fit_indices'], d['indexX'], d['indexH'], ctx=ctx, want_quat=True)
    assert np.allclose(q1, [[1, 0, 0, 0]] * 3, atol=1e-12)


def test_script_from_raw_coordinates_to_Ct(ctx, tmp_path):
    """calculate-Ct-from-traj.py fed raw coordinates: front end + C(t) on the GPU == the reference's C(t) function on the
    oracle's superposed vectors (float32 like the reference holds them)."""
    s = synth.config_shapes(1)
    d = synth.synth_coordinates(s['frames'], 12, 23)
    fn = str(tmp_path / 'traj.npz')
    np.savez(fn, xyz=d['xyz'], ref_xyz=d['ref_xyz'], indexX=d['indexX'], indexH=d['indexH'], fit_indices=d['fit_indices'],
             names=np.arange(2, 14), dt=s['dt'])
    out = str(tmp_path / 'o')
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'calculate-Ct-from-traj.py'), '-s', 'ref.pdb', '-f', fn, '--tau',
           str(s['tau_memory']), '-o', out, '--Ct']
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()
    want, _ = o.superposed_XHvecs(d['xyz'], d['ref_xyz'], d['fit_indices'], d['indexX'], d['indexH'])
    lab, fitv = hostct.superpose_XHvecs(d['xyz'], d['ref_xyz'], d['fit_indices'], d['indexX'], d['indexH'], ctx=ctx)
    from spinrelax_amd import general_scripts as gs
    legs, t, Ct, dCt = gs.load_sxydylist(out + '_Ctint.dat', 'legend')
    v4 = o.reformat_vecs_by_tau([fitv], s['dt'], s['tau_memory'])
    Cr, dCr = o.calculate_Ct_Palmer(v4)
    assert [int(x) for x in legs] == list(range(2, 14))
    assert np.max(np.abs(np.array(Ct).T / Cr - 1)) < 1e-6            # the file keeps 8 significant digits
    v4w = o.reformat_vecs_by_tau([want.astype(np.float32)], s['dt'], s['tau_memory'])
    Cw, _ = o.calculate_Ct_Palmer(v4w)
    assert np.max(np.abs(Cr / Cw - 1)) < 1e-5                        # float32 rounding of the vectors: at most one ulp apart
    legs, t, Cx, dCx = gs.load_sxydylist(out + '_Ctext.dat', 'legend')
    Cl, _ = o.calculate_Ct_Palmer(o.reformat_vecs_by_tau([lab], s['dt'], s['tau_memory']))
    assert np.max(np.abs(np.array(Cx).T / Cl - 1)) < 1e-6


def test_round2_entry_points_refuse_bad_arguments(ctx):
    """Shapes and index ranges are checked on the host before anything is launched (a kernel fed an out-of-range atom
    index or lag would fault): every refusal is a SpinRelaxHipError carrying the library's message."""
    from spinrelax_amd.hip import SpinRelaxHipError
    d = synth.synth_coordinates(8, 4, 5)
    with pytest.raises(SpinRelaxHipError, match='out of range'):
        ctx.xh_vectors(d['xyz'], np.array([0, 2, 4, 99999], dtype=np.int32), d['indexH'])
    with pytest.raises(SpinRelaxHipError, match='fit atom'):
        ctx.xh_vectors(d['xyz'], d['indexX'], d['indexH'], fit_indices=np.array([0, 1, 10 ** 6], dtype=np.int32), ref_xyz=d['ref_xyz'])
    with pytest.raises(SpinRelaxHipError, match='3 fit atoms'):
        ctx.xh_vectors(d['xyz'], d['indexX'], d['indexH'], fit_indices=np.array([0, 1], dtype=np.int32), ref_xyz=d['ref_xyz'])
    with pytest.raises(ValueError):
        ctx.xh_vectors(d['xyz'][..., :2], d['indexX'], d['indexH'])
    q = synth.synth_orientation(50, 3)
    with pytest.raises(SpinRelaxHipError, match='out of range'):
        ctx.dq_moments(q, [1, 50])                      # a lag must leave at least one pair
    with pytest.raises(SpinRelaxHipError, match='out of range'):
        ctx.dq_moments(q, [0])
    with pytest.raises(ValueError):
        ctx.dq_moments(q[:, :3], [1])
    # rsCSA search: per-experiment arrays must match, the column code must be 0 / 1 / 2
    st = np.ones((2, 3, 12))
    ok = dict(stats=st, column=[0, 1], csa_prefactor=[1.0, 1.0], noe_factor=[1.0, 1.0], f_DD=[1.0, 1.0], target=np.ones((2, 3)),
              dtarget=np.zeros((2, 3)), cover=np.ones((2, 3), dtype=np.uint8), has_err=True, csa0=np.full(3, 1e-4), step=1e-5)
    csa, vals, errs, fopt, nfev = ctx.rscsa_search(**ok)
    assert csa.shape == (3,) and np.all(nfev > 0) and np.all(np.isfinite(vals))
    with pytest.raises(SpinRelaxHipError, match='column'):
        ctx.rscsa_search(**dict(ok, column=[0, 3]))
    with pytest.raises(ValueError):
        ctx.rscsa_search(**dict(ok, target=np.ones((2, 4))))
    with pytest.raises(ValueError):
        ctx.rscsa_search(**dict(ok, stats=np.ones((2, 3, 11))))
