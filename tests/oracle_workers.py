"""Oracle work farmed out to plain child processes (numpy + the oracle only: no torch, no GPU):
    python tests/oracle_workers.py fit <in.npz> <out.pkl> <i0> <i1>
runs the oracle's model-order search (optimised_curve_fitting through scipy, fitting_Ct_functions.py:278-345) for residues
i0 .. i1-1 of the arrays t (L,), y (n, L), dy (n, L) in <in.npz>.  Test infrastructure."""
import os
import pickle
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))


def fit_slice(t, Y, dY):
    import sr_oracle as o
    out = []
    for y, dy in zip(Y, dY):
        best, _ = o.optimised_curve_fitting(t, y, dy)
        out.append(None if best is None else (int(best['nParams']), float(best['S2']), [float(v) for v in best['C']],
                                              [float(v) for v in best['tau']], float(best['chiSq'])))
    return out


def fit_all(t, y, dy, tmpdir, nproc=16):
    """the same for every row of y, on nproc child processes; returns the list of per-residue tuples"""
    n = y.shape[0]
    fin = os.path.join(tmpdir, 'oracle_fit_in.npz')
    np.savez(fin, t=t, y=y, dy=dy)
    step = (n + nproc - 1) // nproc
    procs = []
    for k, i0 in enumerate(range(0, n, step)):
        fout = os.path.join(tmpdir, 'oracle_fit_out_%d.pkl' % k)
        procs.append((fout, subprocess.Popen([sys.executable, os.path.abspath(__file__), 'fit', fin, fout, str(i0), str(min(n, i0 + step))],
                                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    out = []
    for fout, p in procs:
        _, err = p.communicate(timeout=1200)
        assert p.returncode == 0, err.decode()[-2000:]
        with open(fout, 'rb') as fp:
            out.extend(pickle.load(fp))
    return out


if __name__ == '__main__':
    if sys.argv[1] == 'fit':
        z = np.load(sys.argv[2])
        i0, i1 = int(sys.argv[4]), int(sys.argv[5])
        with open(sys.argv[3], 'wb') as fp:
            pickle.dump(fit_slice(z['t'], z['y'][i0:i1], z['dy'][i0:i1]), fp)
