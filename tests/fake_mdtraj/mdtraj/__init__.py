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
    """xyz + float32 frame times, as MDTraj holds them for an .xtc file.  `timestep` is MDTraj's property: time[1] - time[0]
    of THIS object's frames, and it raises for a single frame -- so a caller that asks every chunk of an iterload for it
    fails on a one-frame last chunk, and sees last-bit differences between chunks for a time step like 0.1 ps."""
    def __init__(self, xyz, top, time):
        self.xyz = xyz
        self.topology = top
        self.time = np.asarray(time, dtype=np.float32)

    @property
    def n_frames(self):
        return self.xyz.shape[0]

    @property
    def timestep(self):
        if self.n_frames <= 1:
            raise ValueError("Cannot calculate timestep if trajectory has one frame.")
        return self.time[1] - self.time[0]


def _times(f0, f1, dt):
    return (np.arange(f0, f1, dtype=np.float64) * float(dt)).astype(np.float32)


def _open(fn):
    z = np.load(fn, allow_pickle=True)
    sel = {str(k)[4:]: z[k] for k in z.files if str(k).startswith('sel:')}
    return z, Topology(z['resseq'], sel)


def load(fn, top=None):
    z, t = _open(fn)
    xyz = np.asarray(z['xyz'], dtype=np.float32)
    return Trajectory(xyz, t, _times(0, xyz.shape[0], z['dt']))


def iterload(fn, chunk=100, top=None):
    z, t = _open(fn)
    xyz = np.asarray(z['xyz'], dtype=np.float32)
    for f0 in range(0, xyz.shape[0], chunk):
        yield Trajectory(xyz[f0:f0 + chunk], t, _times(f0, min(f0 + chunk, xyz.shape[0]), z['dt']))
