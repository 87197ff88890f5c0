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


def _quat_to_matrix(quat):
    w, x, y, z = quat.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)


def _fit_rmsd(R, xyz, ref_xyz, fit):
    P = xyz[:, fit].astype(np.float64)
    Q = ref_xyz[fit].astype(np.float64)
    P = P - P.mean(axis=1, keepdims=True)
    Q = Q - Q.mean(axis=0)
    return np.sqrt((((np.einsum('nab,nib->nia', R, P) - Q[None]) ** 2).sum(-1)).mean(-1))


def test_superposition_edge_cases_vs_svd_kabsch(ctx):
    """The per-frame superposition (calculate-Ct-from-traj.py:466-467: MDTraj center_coordinates + superpose(ref, frame=0,
    atom_indices=fit)) on the inputs where a least-squares rotation is delicate, against the oracle's SVD Kabsch (MDTraj is
    absent from the image: this step stays "parity unpinned", DESIGN.md section 2 -- the oracle is the mathematics):
      (a) a fit selection DIFFERENT from the vector atoms (--fitsel, :466): only atoms that carry no bond vector;
      (b) MIRROR-IMAGE frames: the best orthogonal map is improper (det = -1), the best PROPER rotation is what superpose
          applies (SVD with the determinant correction; Horn's quaternion is proper by construction);
      (c) a PLANAR fit set (rank-2 covariance: one singular value 0) -- the rotation is still unique;
      (d) a COLLINEAR fit set: the rotation about the line is undetermined, so what must agree is the residual (both are
          optimal) and that the device returns a proper rotation and finite unit vectors."""
    d = synth.synth_coordinates(300, 16, 41)
    xyz, ref = d['xyz'], d['ref_xyz']
    iX, iH = d['indexX'], d['indexH']
    nat = xyz.shape[1]
    extra = np.arange(2 * 16, nat, dtype=np.int32)                     # the scaffold atoms without bond vectors

    def both(xyz_, ref_, fit_):
        lab, fitv, quat = hostct.superpose_XHvecs(xyz_, ref_, fit_, iX, iH, ctx=ctx, want_quat=True)
        want, R = o.superposed_XHvecs(xyz_, ref_, fit_, iX, iH)
        Rq = _quat_to_matrix(quat)
        assert np.max(np.abs(np.linalg.det(Rq) - 1)) < 1e-12 and np.all(np.isfinite(fitv))
        assert np.max(np.abs(np.linalg.norm(fitv.astype(np.float64), axis=2) - 1)) < 2e-7
        return fitv, want, Rq, R

    # (a) fit on the extra atoms only
    fitv, want, Rq, R = both(xyz, ref, extra)
    assert np.max(np.abs(Rq - R)) < 1e-12 and np.max(np.abs(fitv - want)) < 2e-7
    # ... and it is a different superposition from the all-heavy-atom one
    _, _, Rall, _ = both(xyz, ref, d['fit_indices'])
    assert np.max(np.abs(Rq - Rall)) > 1e-6
    # (b) mirror images: x -> -x about the reference centroid, then the trajectory's own tumbling
    mir = xyz.copy()
    mir[..., 0] = -mir[..., 0]
    fitv, want, Rq, R = both(mir, ref, d['fit_indices'])
    assert np.max(np.abs(Rq - R)) < 1e-10 and np.max(np.abs(fitv - want)) < 2e-7
    # the improper optimum would fit better: the proper one leaves a residual of the molecule's size
    assert np.min(_fit_rmsd(R, mir, ref, d['fit_indices'])) > 0.1
    # (c) planar fit set: flatten the extra atoms into the plane z = 1.5 in the reference and in every frame's body frame
    ref_p = ref.copy()
    ref_p[extra, 2] = 1.5
    body = synth._rotate(np.concatenate([d['q'][:, :1], -d['q'][:, 1:]], axis=1)[:, None, :],
                         xyz.astype(np.float64) - xyz[:, d['fit_indices']].astype(np.float64).mean(axis=1, keepdims=True))
    body[:, extra, 2] = 0.0
    xyz_p = (synth._rotate(d['q'][:, None, :], body) + 3.0).astype(np.float32)
    fitv, want, Rq, R = both(xyz_p, ref_p, extra)
    assert np.max(np.abs(Rq - R)) < 1e-9 and np.max(np.abs(fitv - want)) < 2e-7
    # (d) collinear fit set: four atoms on a line
    line = extra[:4]
    ref_l = ref.copy()
    ref_l[line] = np.array([[0.5, 1.0, 1.0], [1.0, 1.0, 1.0], [1.7, 1.0, 1.0], [2.5, 1.0, 1.0]], dtype=np.float32)
    body_l = body.copy()
    body_l[:, line] = (ref_l[line] - ref_l[line].mean(axis=0))[None]
    xyz_l = (synth._rotate(d['q'][:, None, :], body_l) + 3.0).astype(np.float32)
    fitv, want, Rq, R = both(xyz_l, ref_l, line)
    ra, rb = _fit_rmsd(Rq, xyz_l, ref_l, line), _fit_rmsd(R, xyz_l, ref_l, line)
    assert np.max(np.abs(ra - rb)) < 5e-7 and np.max(ra) < 1e-6          # both optimal (float32 coordinates: 1e-7 nm)


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


@pytest.mark.parametrize('dt_ps,tau_ps', [(10.0, 1005.0), (float(np.float32(0.1)), 100.5 * float(np.float32(0.1))), (float(np.float32(0.1)), 10.0)])
def test_mdtraj_loader_streams_chunks_into_resident_vectors(tmp_path, dt_ps, tau_ps):
    """load_mdtraj of the drop-in script EXECUTED (with tests/fake_mdtraj standing in for the absent MDTraj: it reads files
    and resolves selections, nothing else): two trajectory files read in --split chunks of 128 frames, every chunk's
    coordinates turned into bond vectors + superposition on the GPU and appended to the rank's resident vectors, each
    file's tail cut to whole blocks of memory time.  The files must equal, byte for byte, what the same script writes from
    the same coordinates given as .npz arrays (the whole-array path).
    The first file holds 4 * 128 + 1 frames -- its last chunk is ONE frame, for which MDTraj's .timestep raises -- and the
    second case has a 0.1 ps time step, whose float32 frame times give every chunk a time step that differs in its last
    bits: the time step is a property of a file's first chunk only (reference :436-438), compared between files.
    Third case: tau an exact multiple of a 0.1 ps step.  MDTraj's time step is np.float32 and the reference divides by it as
    it is (:241, :255): under NumPy >= 2 that is a float32 division, int(10 / np.float32(0.1)) = 100 frames per chunk, where the
    same division in float64 gives 99 -- the script must keep the reference's arithmetic."""
    s = dict(synth.config_shapes(1))
    s['dt'], s['tau_memory'] = dt_ps, tau_ps
    F = int(s['tau_memory'] / np.float32(s['dt']))
    assert F == 100 and (tau_ps != 10.0 or int(tau_ps / float(np.float32(dt_ps))) == 99)
    n1, n2 = 4 * 128 + 1, 3 * F + 63                                   # 13 and 63 frames of tail to drop
    d1 = synth.synth_coordinates(n1, 12, 31)
    d2 = synth.synth_coordinates(n2, 12, 32)
    natoms = d1['xyz'].shape[1]
    resseq = np.zeros(natoms, dtype=int)
    resseq[d1['indexH']] = np.arange(2, 14)
    sel = {'sel:name H': d1['indexH'], 'sel:name N and not resname PRO': d1['indexX'], 'sel:name CA': d1['fit_indices']}
    np.savez(str(tmp_path / 'ref.npz'), xyz=d1['ref_xyz'][None], resseq=resseq, dt=s['dt'], **sel)
    outs = {}
    for mode in ('mdtraj', 'arrays'):
        files = []
        for k, d in enumerate((d1, d2)):
            fn = str(tmp_path / ('%s_%d.%s' % (mode, k, 'xtc.npz' if mode == 'mdtraj' else 'npz')))
            if mode == 'mdtraj':
                np.savez(fn, xyz=d['xyz'], resseq=resseq, dt=s['dt'], **sel)
                fn2 = fn[:-4]                                             # a name that does not end in .npz: the MDTraj branch
                os.rename(fn, fn2)
                fn = fn2
            else:
                np.savez(fn, xyz=d['xyz'], ref_xyz=d1['ref_xyz'], indexX=d['indexX'], indexH=d['indexH'],
                         fit_indices=d1['fit_indices'], names=np.arange(2, 14), dt=s['dt'])
            files.append(fn)
        out = str(tmp_path / ('o_' + mode))
        cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'calculate-Ct-from-traj.py'), '-s', str(tmp_path / 'ref.npz'), '-f'] + files + \
              ['--tau', str(s['tau_memory']), '-o', out, '--Ct', '--vecHist', '--binary', '--vecAvg', '--S2', '--fitsel', 'name CA',
               '--vecRot', '0.866165 0.392069 -0.308123 -0.033159']
        env = dict(os.environ)
        if mode == 'mdtraj':
            cmd += ['--split', '128']
            env['PYTHONPATH'] = os.path.join(ROOT, 'tests', 'fake_mdtraj') + os.pathsep + env.get('PYTHONPATH', '')
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        if mode == 'mdtraj':
            assert b'%d frames read, %d kept' % (n1, 5 * F) in p.stdout and b'%d frames read, %d kept' % (n2, 3 * F) in p.stdout
        outs[mode] = out
    for suffix in ('_Ctext.dat', '_Ctint.dat', '_avgvec.dat', '_S2.dat'):
        a, b = open(outs['mdtraj'] + suffix, 'rb').read(), open(outs['arrays'] + suffix, 'rb').read()
        assert a == b and len(a) > 100, suffix
    za, zb = np.load(outs['mdtraj'] + '_vecHistogram.npz', allow_pickle=True), np.load(outs['arrays'] + '_vecHistogram.npz', allow_pickle=True)
    assert np.array_equal(za['data'], zb['data']) and za['data'].sum() == 8 * F * 12


def test_host_vector_uploads_move_only_the_ranks_columns(ctx):
    """a rank receives all frames of ITS vector slice -- a strided copy, row pitch 12 V, width 12 nV.
    The library counts what goes through that copy: the host-pointer entry points and the resident-vector path move
    exactly 12 N nV bytes per vector set, once, whatever the width of the host array."""
    e = hostct.lambert_edges()
    N, V, v0, nV, F = 6 * 256, 40, 13, 11, 256
    vecs = synth.synth_vectors(N, V, seed=90)
    b0, c0 = ctx.counter('h2d_vector_bytes'), ctx.counter('vector_uploads')
    Ct, dCt = ctx.ct_palmer(vecs, 6, F, v0=v0, nV=nV)
    assert ctx.counter('h2d_vector_bytes') - b0 == 12 * N * nV and ctx.counter('vector_uploads') - c0 == 1
    full, dfull = ctx.ct_palmer(vecs, 6, F)
    assert np.array_equal(Ct, full[:, v0:v0 + nV]) and np.array_equal(dCt, dfull[:, v0:v0 + nV])
    b1 = ctx.counter('h2d_vector_bytes')
    hist, vsum, outer = ctx.rotate_hist(vecs, np.array(synth.Q_EXT), e[0], e[1], v0=v0, nV=nV, block_len=F)
    assert ctx.counter('h2d_vector_bytes') - b1 == 12 * N * nV
    # resident vectors: two files with tails, ONE pass over PCIe, then C(t) and the vector distribution from the same planes
    a, b = synth.synth_vectors(3 * F + 17, V, seed=91), synth.synth_vectors(2 * F + 200, V, seed=92)
    b2, c2 = ctx.counter('h2d_vector_bytes'), ctx.counter('vector_uploads')
    rv = ctx.vectors(nV)
    rv.append(a[:3 * F], v0=v0)
    rv.append(b[:2 * F], v0=v0)
    assert rv.frames == 5 * F
    Ct2, dCt2 = rv.ct(5, F)
    h2, s2, o2 = rv.hist(np.array(synth.Q_EXT), e[0], e[1], block_len=F)
    assert ctx.counter('h2d_vector_bytes') - b2 == 12 * 5 * F * nV and ctx.counter('vector_uploads') - c2 == 2
    cat = np.ascontiguousarray(np.concatenate([a[:3 * F], b[:2 * F]]))
    Ct3, dCt3 = ctx.ct_palmer(cat, 5, F, v0=v0, nV=nV)
    h3, s3, o3 = ctx.rotate_hist(cat, np.array(synth.Q_EXT), e[0], e[1], v0=v0, nV=nV, block_len=F)
    assert np.array_equal(Ct2, Ct3) and np.array_equal(dCt2, dCt3) and np.array_equal(h2, h3) and np.array_equal(s2, s3) and np.array_equal(o2, o3)
    assert np.array_equal(rv.download(F, 7), cat[F:F + 7, v0:v0 + nV])
    rv.truncate(4 * F)
    Ct4, _ = rv.ct(4, F)
    assert np.array_equal(Ct4, ctx.ct_palmer(cat[:4 * F], 4, F, v0=v0, nV=nV)[0])
    rv.close()


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


def test_resident_vector_appends_on_different_streams_chain(ctx):
    """sr_vectors_append_* on stream A, then on stream B, the consumer on stream C (sr_set_stream in between): every append first
    waits for whatever touched the object last, so the second append's frames land behind the first one's, a growth copy of the
    full object runs behind the frames it copies, and the consumer sees all of them.  Page-locked sources take the direct copy
    (no staging pass) and must give the same object."""
    import ctypes
    import torch
    F, V = 256, 24
    a, b, c = (synth.synth_vectors(n, V, seed=sd) for n, sd in ((3 * F, 61), (2 * F, 62), (F, 63)))
    whole = np.ascontiguousarray(np.concatenate([a, b, c]))
    want = ctx.ct_palmer(whole, 6, F)
    sA, sB, sC = (torch.cuda.Stream() for _ in range(3))
    big = torch.zeros((4096, 4096), device='cuda')
    rv = ctx.vectors(V, capacity=3 * F)                     # the third append has to grow the object
    with torch.cuda.stream(sA):
        for _ in range(4):
            big = big @ big                                 # stream A is busy: its copies start late
    ctx.set_stream(sA.cuda_stream)
    rv.append(a)
    ctx.set_stream(sB.cuda_stream)
    rv.append(b)
    # a page-locked source: the direct asynchronous copy
    nbytes = c.nbytes
    addr = ctx.host_alloc(nbytes)
    pin = np.frombuffer((ctypes.c_char * nbytes).from_address(addr), dtype=np.float32).reshape(c.shape)
    np.copyto(pin, c)
    ctx.set_stream(sA.cuda_stream)
    rv.append_pinned(addr, c.shape[0], V)
    assert rv.frames == 6 * F
    ctx.set_stream(sC.cuda_stream)
    got = rv.ct(6, F)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(rv.download(0, 6 * F), whole)
    ctx.set_stream(0)
    torch.cuda.synchronize()
    del pin
    ctx.host_free(addr)
    rv.close()
