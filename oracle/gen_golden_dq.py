#!/usr/bin/env python3
"""
gen_golden_dq.py -- fixtures for SURVEY.md section 8(f)-2 (global rotational diffusion, calculate-dq-distribution.py),
made with the REAL reference functions imported from /root/reference:

    obtain_self_dq (:102-109), average_LegendreP1quat (:111-112), average_LegendreP1quat_chunk (:128-135),
    conduct_exponential_fit (:199-208), format_header (:222-275), print_model_fits_gen (:280-339),
    transforms3d_supplement.rotate_vector_simd (rotation of every sample into a given frame),
    general_scripts.print_xylist (the -aniso_q.dat writer).

The script's main flow sits under `if __name__ == '__main__'` and needs the absent third-party package transforms3d
(qops.nearly_equivalent inside average_anisotropic_tensor, quat_frame_transform_min); the two lines that cannot be
called are restated here and marked: the outer-product mean `np.mean(np.einsum('ij,ik->ijk', vq, vq), axis=0)` (:126) and
the chunk boundaries (:137-143).  The PAF frame quaternion comes from spinrelax_amd.quaternions (restated transforms3d
algorithm; its parity is pinned by properties only) and is stored as an INPUT of the fixture.

Run in the build container only:   python oracle/gen_golden_dq.py
TEST INFRASTRUCTURE ONLY.
"""
import contextlib
import importlib.util
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader                                   # noqa: E402
from spinrelax_amd import synth                     # noqa: E402
from spinrelax_amd import quaternions as myq        # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
ref = ref_loader.load()
spec = importlib.util.spec_from_file_location('ref_dq', os.path.join(ref_loader.REF, 'calculate-dq-distribution.py'))
rdq = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rdq)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def tensor(vq):
    return np.mean(np.einsum('ij,ik->ijk', vq, vq), axis=0)                     # :126


def chunks(ndat, nchunk):
    nblock = int(np.ceil(1.0 * ndat / nchunk))                                    # :137
    return [(nblock * i, min(ndat, nblock * (i + 1))) for i in range(nchunk)]


def run_case(tag, q32, dt_ps, min_int, max_int, skip_int, num_chunk):
    lags = list(range(min_int, max_int + 1, skip_int))
    nl = len(lags)
    q = q32                                           # float32 in, like data[1:5].T of the PLUMED reader
    iso = np.zeros(nl)
    moi = np.zeros((nl, 3, 3))
    moiR = np.zeros((nl, 3, 3))
    ch_iso = np.zeros((num_chunk, nl))
    ch_moi = np.zeros((num_chunk, nl, 3, 3))
    ch_moiR = np.zeros((num_chunk, nl, 3, 3))
    q_frame = None
    for k, d in enumerate(lags):
        vq = rdq.obtain_self_dq(q, d)[..., 1:4]
        assert vq.dtype == np.float64
        nd = vq.shape[0]
        iso[k] = rdq.average_LegendreP1quat(nd, vq)
        moi[k] = tensor(vq)
        if q_frame is None:
            eigval, eigvec = np.linalg.eigh(moi[k])
            q_frame = myq.quat_frame_transform_min(eigvec.T)
        vr = ref.qs.rotate_vector_simd(vq, q_frame, axis=-1)                     # average_anisotropic_tensor :124-125
        moiR[k] = tensor(vr)
        ch_iso[:, k] = rdq.average_LegendreP1quat_chunk(nd, vq, num_chunk)
        for c, (a, b) in enumerate(chunks(nd, num_chunk)):
            ch_moi[c, k] = tensor(vq[a:b])
            ch_moiR[c, k] = tensor(vr[a:b])
    dtlist = np.array(lags) * dt_ps
    aniso2 = np.stack([1 - 2 * moiR[:, i, i] for i in range(3)])
    ch_aniso2 = np.stack([np.stack([1 - 2 * ch_moiR[c, :, i, i] for i in range(3)]) for c in range(num_chunk)])
    # The isotropic list is 1 - (2/3) sum_i |v_i|^2 (see oracle/sr_oracle.py:average_LegendreP1quat): for any realistic
    # trajectory length it is far below -0.5 and the reference's own guess, log((y1+0.5)/(y0+0.5)), raises
    # "math domain error" -- recorded as such (iso_fit_error = 1, no -iso.dat fixture).
    try:
        tau_iso = quiet(rdq.conduct_exponential_fit, dtlist, iso, 1.5, -0.5)
        ch_tau_iso = [quiet(rdq.conduct_exponential_fit, dtlist, ch_iso[c], 1.5, -0.5) for c in range(num_chunk)]
        iso_err = 0
    except ValueError as exc:
        print('reference iso fit raised:', exc)
        tau_iso, ch_tau_iso, iso_err = np.nan, [np.nan] * num_chunk, 1
    taus = np.array([quiet(rdq.conduct_exponential_fit, dtlist, aniso2[i], 0.5, 0.5) for i in range(3)])
    ch_taus = np.array([[quiet(rdq.conduct_exponential_fit, dtlist, ch_aniso2[c][i], 0.5, 0.5) for i in range(3)]
                        for c in range(num_chunk)])
    # the reference's own writers
    models = rdq.anisotropic_decay_noc(dtlist, taus.reshape((3, 1)))
    hdr = rdq.format_header('aniso_err', taus, ch_taus)
    hdr.append(rdq.format_header_quat(q_frame))
    pl = [np.concatenate((aniso2, models))]
    for c in range(num_chunk):
        pl.append(np.concatenate((ch_aniso2[c], rdq.anisotropic_decay_noc(dtlist, ch_taus[c].reshape((3, 1))))))
    quiet(rdq.print_model_fits_gen, os.path.join(GOLD, '%s-aniso2.dat' % tag), 3, hdr, dtlist, pl)
    if not iso_err:
        model = rdq.isotropic_decay(dtlist, tau_iso)
        pl = [[iso, model]]
        for c in range(num_chunk):
            pl.append([ch_iso[c], rdq.isotropic_decay(dtlist, ch_tau_iso[c])])
        quiet(rdq.print_model_fits_gen, os.path.join(GOLD, '%s-iso.dat' % tag), 3, rdq.format_header('iso_err', tau_iso, ch_tau_iso),
              dtlist, pl)
    # no-chunk variants of the headers / files
    hdr = rdq.format_header('aniso', taus)
    hdr.append(rdq.format_header_quat(q_frame))
    quiet(rdq.print_model_fits_gen, os.path.join(GOLD, '%s-aniso2_nochunk.dat' % tag), 2, hdr, dtlist, np.concatenate((aniso2, models)))
    path = os.path.join(GOLD, '%s_dq.npz' % tag)
    np.savez_compressed(path, q32=q32 if q32.shape[0] <= 2000 else np.zeros((0, 4), np.float32),
                        q_sha=__import__('hashlib').sha256(np.ascontiguousarray(q32).tobytes()).hexdigest(),
                        nframes=q32.shape[0], dt_ps=dt_ps, lags=np.array(lags), num_chunk=num_chunk, q_frame=np.array(q_frame),
                        iso=iso, moi=moi, moiR=moiR, chunk_iso=ch_iso, chunk_moi=ch_moi, chunk_moiR=ch_moiR,
                        tau_iso=tau_iso, chunk_tau_iso=np.array(ch_tau_iso), iso_fit_error=iso_err, taus=taus, chunk_taus=ch_taus)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; tau_iso %.6g, taus %s, iso[:3] %s' % (tau_iso, taus, iso[:3]))
    return path


def main():
    # (a) 20 000-frame anisotropic random walk from spinrelax_amd.synth (regenerated bit for bit by the tests)
    q = synth.synth_orientation(20000, 11)
    run_case('dqA', q, synth.DT_PS, 10, 500, 10, 4)
    # (b) the 1 000-frame PLUMED file of the de-tumbling fixture, read by the reference's reader
    import plumedcolvario as ref_pl
    names, data = quiet(ref_pl.read_from_plumedprint, os.path.join(GOLD, 'cfg1_colvar-qorient'))
    run_case('dqB', np.ascontiguousarray(data[1:5].T), float(data[0, 1] - data[0, 0]), 1, 40, 1, 3)
    # (c) float64 quaternions (what the reference's gmx-rotmat route holds: rotmatrix_to_quaternion keeps float64,
    #     calculate-dq-distribution.py:406-423, 482-497): the synthetic walk nudged off the float32 grid and re-normalised in
    #     float64; only the per-lag reductions are stored (obtain_self_dq / tensor / average_LegendreP1quat[_chunk])
    q64 = synth.synth_orientation(5000, 12).astype(np.float64)
    q64 += 1e-9 * np.sin(np.arange(q64.size, dtype=np.float64)).reshape(q64.shape)
    q64 /= np.sqrt((q64 * q64).sum(axis=1))[:, None]
    assert not np.array_equal(q64, q64.astype(np.float32).astype(np.float64))
    lags = list(range(1, 40, 3))
    nch = 3
    iso = np.zeros(len(lags))
    moi = np.zeros((len(lags), 3, 3))
    ch_iso = np.zeros((nch, len(lags)))
    ch_moi = np.zeros((nch, len(lags), 3, 3))
    for k, d in enumerate(lags):
        vq = rdq.obtain_self_dq(q64, d)[..., 1:4]
        nd = vq.shape[0]
        iso[k] = rdq.average_LegendreP1quat(nd, vq)
        moi[k] = tensor(vq)
        ch_iso[:, k] = rdq.average_LegendreP1quat_chunk(nd, vq, nch)
        for c, (a, b) in enumerate(chunks(nd, nch)):
            ch_moi[c, k] = tensor(vq[a:b])
    np.savez_compressed(os.path.join(GOLD, 'dqC_dq.npz'), q64=q64, lags=np.array(lags), num_chunk=nch, iso=iso, moi=moi,
                        chunk_iso=ch_iso, chunk_moi=ch_moi)
    print('wrote dqC_dq.npz (float64 quaternions)')
    mf = os.path.join(GOLD, 'MANIFEST.json')
    man = json.load(open(mf))
    man['dqC_dq.npz'] = 'reference reductions of calculate-dq-distribution.py on FLOAT64 quaternions (oracle/gen_golden_dq.py, case c)'
    for tag in ('dqA', 'dqB'):
        man['%s_dq.npz' % tag] = 'reference reductions of calculate-dq-distribution.py (oracle/gen_golden_dq.py)'
    with open(mf, 'w') as fp:
        json.dump(man, fp, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
