"""
CPU tests for SURVEY.md section 8(f)-2 (global rotational diffusion): the oracle's restatement of the reference's
difference-quaternion reductions against fixtures the reference's own functions produced (oracle/gen_golden_dq.py), and the
host side of spinrelax_amd/dq_distribution.py (fits, headers, writers: byte-identical files) fed with the fixture's
per-lag lists -- no GPU involved.
"""
import filecmp
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, golden
import sr_oracle as o
from spinrelax_amd import synth
from spinrelax_amd import dq_distribution as dq
from spinrelax_amd import quaternions as qops


def _q_of(tag, g):
    q = synth.synth_orientation(int(g['nframes']), 11) if tag == 'dqA' else g['q32']
    assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == str(g['q_sha'])
    return q


@pytest.mark.parametrize('tag', ['dqA', 'dqB'])
def test_oracle_dq_reductions_vs_reference(tag):
    g = golden('%s_dq.npz' % tag)
    q = _q_of(tag, g)
    nch = int(g['num_chunk'])
    lags = g['lags'][:: max(1, len(g['lags']) // 8)]           # a spread of lags keeps this test in seconds
    sel = [list(g['lags']).index(d) for d in lags]
    m = o.dq_moments(q, lags, nch)
    tot = m.sum(axis=1)
    assert np.array_equal(tot[:, 6], int(g['nframes']) - lags)
    moi = dq.moments_to_tensor(tot)
    assert np.max(np.abs(moi - g['moi'][sel])) < 1e-15
    ch = dq.moments_to_tensor(m)                                 # (nl, nch, 3, 3)
    assert np.max(np.abs(np.moveaxis(ch, 1, 0) - g['chunk_moi'][:, sel])) < 1e-15
    iso = dq.average_LegendreP1quat(tot)
    assert np.max(np.abs(iso - g['iso'][sel]) / np.abs(g['iso'][sel])) < 1e-12
    chi = dq.average_LegendreP1quat(m)                           # (nl, nch)
    assert np.max(np.abs(chi.T - g['chunk_iso'][:, sel]) / np.abs(g['chunk_iso'][:, sel])) < 1e-12
    # rotating every sample into the frame and averaging (what the reference does) == R M R^T
    R = dq.average_anisotropic_tensor(tot, g['q_frame'])
    assert np.max(np.abs(R - g['moiR'][sel])) < 1e-15
    # and the functions themselves on one lag
    v = o.obtain_self_dq(q, int(lags[0]))[:, 1:4]
    assert abs(o.average_LegendreP1quat(v) - g['iso'][sel[0]]) < 1e-12 * abs(g['iso'][sel[0]])
    assert np.max(np.abs(o.average_anisotropic_tensor(v) - g['moi'][sel[0]])) < 1e-16


def _res_from_golden(g):
    nl = len(g['lags'])
    nch = int(g['num_chunk'])
    a2 = np.stack([1 - 2 * g['moiR'][:, i, i] for i in range(3)])
    ch_a2 = np.stack([np.stack([1 - 2 * g['chunk_moiR'][c, :, i, i] for i in range(3)]) for c in range(nch)])
    return dict(out_dtlist=g['lags'] * float(g['dt_ps']), out_isolist=g['iso'], out_aniso2list=a2, chunk_isolist=g['chunk_iso'],
                chunk_aniso2list=ch_a2, out_qlist=np.zeros((4, nl)), out_moilist=np.zeros((nl, 3, 3)), q_frame=g['q_frame'])


@pytest.mark.filterwarnings('ignore::DeprecationWarning')      # math.exp of the 1-element array fmin_powell hands over, as in the reference
@pytest.mark.parametrize('tag', ['dqA', 'dqB'])
def test_fits_headers_and_writers_byte_exact(tag, tmp_path, capsys):
    """conduct_exponential_fit (Powell, the reference's point-by-point objective), format_header, print_model_fits_gen:
    same decay times to the last bit and byte-identical -aniso2.dat / -iso.dat."""
    g = golden('%s_dq.npz' % tag)
    res = _res_from_golden(g)
    nch = int(g['num_chunk'])
    pref = str(tmp_path / tag)
    fitted = dq.fit_and_write(res, pref, num_chunk=nch, bDoIso=False, bDoAniso=True)
    assert np.array_equal(fitted['aniso_taus'], g['taus'])
    assert np.array_equal(fitted['aniso_chunk_taus'], g['chunk_taus'])
    assert filecmp.cmp(pref + '-aniso2.dat', os.path.join(GOLD, '%s-aniso2.dat' % tag), shallow=False)
    dq.fit_and_write(res, pref + 'n', num_chunk=0, bDoIso=False, bDoAniso=True)
    assert filecmp.cmp(pref + 'n-aniso2.dat', os.path.join(GOLD, '%s-aniso2_nochunk.dat' % tag), shallow=False)
    if int(g['iso_fit_error']):
        # the reference's isotropic list is 1 - (2/3) sum |v|^2 (not a mean): its own initial guess takes the log of a
        # negative number and raises; so does the mirror
        with pytest.raises(ValueError):
            dq.fit_and_write(res, pref + 'i', num_chunk=nch, bDoIso=True, bDoAniso=False)
    else:
        f2 = dq.fit_and_write(res, pref + 'i', num_chunk=nch, bDoIso=True, bDoAniso=False)
        assert f2['iso_tau'] == float(g['tau_iso'])
        assert filecmp.cmp(pref + 'i-iso.dat', os.path.join(GOLD, '%s-iso.dat' % tag), shallow=False)


def test_quat_frame_transform_properties():
    """transforms3d is absent, so the reference's quat_frame_transform_min cannot run here: its restatement is pinned by
    what it must do -- bring the frame's z axis onto +-z and its x axis onto +-x, by the smaller of the two rotations."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        A = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        if np.linalg.det(A) < 0:
            A[2] *= -1
        q = qops.quat_frame_transform_min(A)
        assert abs(np.dot(q, q) - 1) < 1e-12
        R = qops.rotation_matrix(q)
        z = R @ A[2]
        x = R @ A[0]
        assert abs(abs(z[2]) - 1) < 1e-12 and abs(abs(x[0]) - 1) < 1e-12
        for v in (A[0], A[1], A[2]):
            assert np.allclose(R @ v, qops.rotate_vector(v, q), atol=1e-14)
    # conjugation identities used by dq_distribution.average_anisotropic_tensor
    q = qops.axangle2quat([1.0, 2.0, -0.5], 0.7)
    M = rng.standard_normal((3, 3))
    M = M @ M.T
    v = rng.standard_normal((100, 3))
    rot = np.array([qops.rotate_vector(x, q) for x in v])
    R = qops.rotation_matrix(q)
    assert np.allclose(np.einsum('ij,ik->jk', rot, rot), R @ np.einsum('ij,ik->jk', v, v) @ R.T, atol=1e-12)
    assert qops.nearly_equivalent(q, -q) and not qops.nearly_equivalent(q, qops.qeye())
    assert np.allclose(qops.mat2quat(R), q if q[0] > 0 else -q, atol=1e-12)
    assert np.allclose(qops.qmult(q, qops.qinverse(q)), qops.qeye(), atol=1e-15)


def test_frame_intervals_like_reference():
    # float32 time step as the PLUMED reader yields it; run-all.bash passes --mindt t100 --skip t100 --maxdt tau
    dt = np.float32(10.0)
    assert dq.frame_intervals(dt, 100.0, 5000.0, 100.0) == (10, 500, 10)
    assert dq.frame_intervals(dt, 0.0, 1000.0, 0.0) == (1, 100, 1)
    assert dq.calculate_anisotropies(np.array([3.0, 1.0, 2.0]))[0] == 2.0
