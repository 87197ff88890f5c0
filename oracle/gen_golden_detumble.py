#!/usr/bin/env python3
"""
gen_golden_detumble.py -- fixture for SURVEY.md section 8(f)-1 (per-frame de-tumbling from colvar-qorient), made by
running the REAL reference functions (imported from /root/reference through oracle/ref_loader.py):

    plumedcolvario.read_from_plumedprint      on a synthetic PLUMED PRINT file (float32 fields, like PLUMED writes)
    transforms3d_supplement.rotate_vector_simd with one quaternion per frame, bond by bond: v (N, 3), q (N, 4) --
                                              the only N-D form its broadcasting supports (q (N, 1, 4) against
                                              v (N, V, 3) fails inside decompose_quat's reshape)
    calculate-Ct-from-traj.py: reformat_vecs_by_tau + calculate_Ct_Palmer on the de-tumbled vectors

Run in the build container only:   python oracle/gen_golden_detumble.py
Writes tests/golden/cfg1_colvar-qorient (data file), tests/golden/cfg1_detumble.npz and updates MANIFEST.json.
"""
import contextlib
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

GOLD = os.path.join(ROOT, 'tests', 'golden')
ref = ref_loader.load()
import plumedcolvario as ref_pl                     # noqa: E402  (the reference's module; ref_loader put it on sys.path)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def tumbling(nframes, seed):
    """A smooth random orientation trajectory: product of small random rotations (unit quaternions, w x y z)."""
    rng = np.random.default_rng(seed)
    q = np.empty((nframes, 4))
    cur = np.array([0.312824, 0.361795, -0.802215, -0.357347])
    cur /= np.linalg.norm(cur)
    for n in range(nframes):
        q[n] = cur
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        th = 0.08 * rng.standard_normal()
        d = np.concatenate(([np.cos(th / 2)], np.sin(th / 2) * ax))
        w0, v0 = cur[0], cur[1:]
        w1, v1 = d[0], d[1:]
        cur = np.concatenate(([w0 * w1 - v0 @ v1], w0 * v1 + w1 * v0 + np.cross(v0, v1)))
        cur /= np.linalg.norm(cur)
    return q


def main():
    s = synth.config_shapes(1)
    body = synth.synth_config(1)                                  # (1000, 32, 3) float32, "true" internal motion
    q_true = tumbling(s['frames'], 20240611)
    fn = os.path.join(GOLD, 'cfg1_colvar-qorient')
    with open(fn, 'w') as fp:
        fp.write('#! FIELDS time q.w q.x q.y q.z rest0.bias\n')
        for n in range(s['frames']):
            fp.write(' %f %f %f %f %f %f\n' % (n * s['dt'], q_true[n, 0], q_true[n, 1], q_true[n, 2], q_true[n, 3], 0.0))
    names, data = quiet(ref_pl.read_from_plumedprint, fn)
    assert data.dtype == np.float32 and data.shape == (6, s['frames'])
    q32 = np.ascontiguousarray(data[1:5].T)                       # what a consumer of the file sees
    # lab-frame vectors: the molecule tumbles with q(t) (file precision), stored as float32 like MDTraj coordinates
    q64 = q32.astype(np.float64)
    lab = np.stack([ref.qs.rotate_vector_simd(body[:, v, :], q64) for v in range(body.shape[1])], axis=1).astype(np.float32)
    qinv = q32.astype(np.float64)
    qinv[:, 1:] *= -1.0
    back64 = np.stack([ref.qs.rotate_vector_simd(lab[:, v, :], qinv) for v in range(lab.shape[1])], axis=1)   # float64
    assert back64.dtype == np.float64
    back32 = back64.astype(np.float32)
    v4 = quiet(ref.calcCt.reformat_vecs_by_tau, [back32], s['dt'], s['tau_memory'])
    Ct64, dCt64 = quiet(ref.calcCt.calculate_Ct_Palmer, v4.astype(np.float64))
    path = os.path.join(GOLD, 'cfg1_detumble.npz')
    np.savez_compressed(path, field_names=np.array(names), colvar=np.asarray(data), q32=q32, lab=lab, body64=back64[:, :8],
                        Ct64=Ct64, dCt64=dCt64, max_dev_from_true_body=np.max(np.abs(back64 - body)))
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; de-tumbled vs original body vectors: max |diff| %.2e'
          % np.max(np.abs(back64 - body)))
    mf = os.path.join(GOLD, 'MANIFEST.json')
    man = json.load(open(mf))
    man['cfg1_detumble.npz'] = dict(field_names=[6], colvar=[6, s['frames']], q32=[s['frames'], 4], lab=list(lab.shape),
                                    body64=list(back64[:, :8].shape), Ct64=list(Ct64.shape), dCt64=list(dCt64.shape))
    man['cfg1_colvar-qorient'] = 'synthetic PLUMED PRINT file read by the reference reader for cfg1_detumble.npz'
    with open(mf, 'w') as fp:
        json.dump(man, fp, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
