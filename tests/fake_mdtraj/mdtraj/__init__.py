"""
A stand-in for the parts of MDTraj that scripts/calculate-Ct-from-traj.py::load_mdtraj touches, so that the loader EXECUTES
in an image without MDTraj (tests/test_gpu_frontend.py puts this directory on PYTHONPATH): load / iterload of "trajectory"
and "topology" files that are .npz archives written by the test (xyz, atom resSeq numbers, and the index lists a selection
string resolves to).  It reads files and resolves selections -- what MDTraj does for the script -- and nothing else; the
vectors, centring and superposition are the script's (GPU) work.  Test infrastructure, not product.
"""
import numpy as np


class _Residue:
    def __init__(self, resSeq):
        self.resSeq = int(resSeq)


class _Atom:
    def __init__(self, resSeq):
        self.residue = _Residue(resSeq)


class Topology:
    def __init__(self, resseq, selections):
        self._resseq = resseq
        self._sel = selections

    def select(self, text):
        if text == 'all':
            return np.arange(len(self._resseq))
        if text not in self._sel:
            raise ValueError('fake mdtraj: unknown selection %r' % text)
        return np.array(self._sel[text])

    def atom(self, k):
        return _Atom(self._resseq[k])


class Trajectory:
    def __init__(self, xyz, top, timestep):
        self.xyz = xyz
        self.topology = top
        self.timestep = timestep


def _open(fn):
    z = np.load(fn, allow_pickle=True)
    sel = {str(k)[4:]: z[k] for k in z.files if str(k).startswith('sel:')}
    return z, Topology(z['resseq'], sel)


def load(fn, top=None):
    z, t = _open(fn)
    return Trajectory(np.asarray(z['xyz'], dtype=np.float32), t, float(z['dt']))


def iterload(fn, chunk=100, top=None):
    z, t = _open(fn)
    xyz = np.asarray(z['xyz'], dtype=np.float32)
    for f0 in range(0, xyz.shape[0], chunk):
        yield Trajectory(xyz[f0:f0 + chunk], t, float(z['dt']))
