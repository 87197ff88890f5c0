"""
Host-side mirror of the spectral-density / relaxation part of the reference (spectral_densities.py and
the _obtain_* loops of calculate-relaxations-from-Ct.py).  Class and function names, argument meaning,
unit handling and output dtypes follow the reference; the per-residue / per-bin / per-field arithmetic
runs in ONE batched GPU launch (sr_jomega_relax_f64) instead of a Python loop over residues.

What lives here is host bookkeeping only: nuclear constants, angular frequencies, the D tensor
conversions, packing of the fitted-C(t) parameters, and the float32 datablock layout the writers expect.
"""
import sys

import numpy as np

from . import hip
from . import dist as srdist
from . import _hostmath as _nph

# spectral_densities.py:50-67, :39-48
GAMMA = {'1H': 267.513e6, '13C': 67.262e6, '15N': -27.116e6, '17O': -36.264e6, '19F': 251.662e6, '31P': 108.291e6}
CSA_DEFAULT = {'15N': -170e-6, '13C': -130e-6}


def _ctx(ctx):
    return ctx if ctx is not None else hip.default_context()


def _return_time_fact(tu):
    """spectral_densities.py:1831-1844."""
    table = {'ps': 1.0e-12, 'ns': 1.0e-9, 'us': 1.0e-6, 'ms': 1.0e-3, 's': 1.0e-0}
    if tu not in table:
        print("= = ERROR in object definition: invalid time unit definition!", file=sys.stderr)
        return None
    return table[tu]


class gyromag:
    """spectral_densities.py:23-79 (gamma in rad s^-1 T^-1, CSA default per isotope)."""

    def __init__(self, isotope, csa=None):
        self.isotope = isotope
        self.gamma = GAMMA[isotope]
        self.csa = CSA_DEFAULT.get(isotope, 0.0) if csa is None else csa

    def set_csa(self, csa, i=None):
        if i is None:
            self.csa = csa
        else:
            self.csa[i] = csa

    def get_csa(self, i=None):
        if i is None or np.ndim(self.csa) == 0:
            return self.csa
        return self.csa[i]


class diffusionModel:
    """The subset of spectral_densities.py:1450-1558 the hot path uses: 'direct_transform',
    'rigid_sphere_D' (D = Diso), 'rigid_sphere_T', 'rigid_symmtop_D' (D = [Dpar, Dperp]),
    'rigid_symmtop_Dref' (Diso, aniso)."""

    def __init__(self, model, timeUnit, *args):
        self.timeUnit = timeUnit
        self.time_fact = _return_time_fact(timeUnit)
        if model == 'direct_transform':
            self.name = 'direct_transform'
            self.D = np.nan
        elif model == 'rigid_sphere_T':
            self.name = 'rigid_sphere'
            self.D = 1.0 / (6.0 * float(args[0]))
        elif model == 'rigid_sphere_D':
            self.name = 'rigid_sphere'
            self.D = float(args[0])
        elif model == 'rigid_symmtop_Dref':
            self.name = 'rigid_symmtop'
            Dperp = 3.0 * args[0] / (2.0 + args[1])
            self.D = np.array([args[1] * Dperp, Dperp])
        elif model == 'rigid_symmtop_D':
            self.name = 'rigid_symmtop'
            self.D = np.array([args[0], args[1]], dtype=float)
        else:
            print("= = ERROR: rotational diffusion model %s is not supported by the GPU path." % model, file=sys.stderr)
            sys.exit(1)

    def set_time_unit(self, tu):
        old = self.time_fact
        self.time_fact = _return_time_fact(tu)
        self.timeUnit = tu
        self.D = self.D * (self.time_fact / old)

    def change_Diso(self, Diso):
        """spectral_densities.py:1529-1542."""
        if self.name == 'rigid_sphere':
            self.D = Diso
        elif self.name == 'rigid_symmtop':
            tmp = self.D[0] / self.D[1]
            Dperp = 3.0 * Diso / (2.0 + tmp)
            self.D = np.array([tmp * Dperp, Dperp])


class relaxationModel:
    """Old API object of calculate-relaxations-from-Ct.py (spectral_densities.py:1560-1811): nuclei,
    field, time unit, the five angular frequencies [0, wX, wH-wX, wH, wH+wX] and the diffusion model."""
    iOmX = 1
    iOmH = 3

    def __init__(self, bondType, B_0):
        self.timeUnit = 'ns'
        self.time_fact = _return_time_fact(self.timeUnit)
        self.distUnit = 'nm'
        self.dist_fact = 1.0e-9
        self.bondType = bondType
        self.B_0 = B_0
        if bondType == 'NH':
            self.gH = gyromag('1H')
            self.gX = gyromag('15N')
        elif bondType == 'CH':
            self.gH = gyromag('1H')
            self.gX = gyromag('13C')
        else:
            print("= = ERROR in relaxationModel: wrong bondType definition! = = %s" % bondType, file=sys.stderr)
            sys.exit(1)
        self.rXH = 1.02e-1
        self.set_rotdif_model('rigid_sphere_T', 1.0)
        self.set_freq_relaxation()

    def set_B0(self, B_0):
        self.B_0 = B_0

    def set_time_unit(self, tu):
        """spectral_densities.py:1619-1626: omega is rescaled (not recomputed), gammas stay in s^-1."""
        old = self.time_fact
        self.time_fact = _return_time_fact(tu)
        self.timeUnit = tu
        self.omega *= self.time_fact / old
        self.rotdifModel.set_time_unit(tu)

    def set_freq_relaxation(self):
        """spectral_densities.py:1630-1645."""
        self.num_omega = 5
        self.omega = np.zeros(5)
        self.omega[3] = -1.0 * self.gH.gamma * self.B_0 * self.time_fact
        self.omega[1] = -1.0 * self.gX.gamma * self.B_0 * self.time_fact
        self.omega[2] = self.omega[3] - self.omega[1]
        self.omega[4] = self.omega[3] + self.omega[1]

    def print_frequencies(self):
        print("# Order of frequencies for %s - %s relaxation:" % (self.gX.isotope, self.gH.isotope))
        print("# 0  iOmX    iOmH-iOmX   iOmH    iOmH+iOmX")
        print(self.omega)

    def set_rotdif_model(self, model, *args):
        self.rotdifModel = diffusionModel(model, self.timeUnit, *args)

    def get_f_DD(self):
        """spectral_densities.py:1696."""
        return 0.10 * 1.1121216813552401e-82 * self.gH.gamma ** 2.0 * self.gX.gamma ** 2.0 * (self.rXH * self.dist_fact) ** -6.0

    def get_f_CSA(self, CSAvalue=None):
        """spectral_densities.py:1698-1701."""
        csa = self.gX.csa if CSAvalue is None else CSAvalue
        return 2.0 / 15.0 * csa ** 2.0 * (self.gX.gamma * self.B_0) ** 2

    def calculate_rho_from_relaxation(self, rvec):
        """spectral_densities.py:1788-1800 (Ghose, Fushman & Cowburn 2001, eq. 4), host scalar maths."""
        R1, R2, NOE = rvec[0], rvec[1], rvec[2]
        HF = -0.2 * (self.gX.gamma / self.gH.gamma) * (1 - NOE) * R1
        R1p = R1 - 7.0 * (0.921 / 0.87) ** 2.0 * HF
        R2p = R2 - 6.5 * (0.955 / 0.87) ** 2.0 * HF
        return 4.0 / 3.0 * R1p / (2.0 * R2p - R1p)


# ---------------------------------------------------------------------------------------------------
# npufunc drop-in (Jomega/Jomega.c) and the J(omega) helpers built on it
# ---------------------------------------------------------------------------------------------------
class _JomegaUfunc:
    """Stand-in for `npufunc.Jomega` (Jomega/Jomega.c:135-156): callable with numpy broadcasting and
    `.outer`, evaluated on the GPU (sr_jomega_f64)."""
    types = ['ee->e', 'ff->f', 'dd->d', 'gg->g']

    def __call__(self, x, y, ctx=None):
        return _ctx(ctx).jomega(x, y)

    def outer(self, x, y, ctx=None):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        return _ctx(ctx).jomega(x.reshape(x.shape + (1,) * y.ndim), y)


Jomega = _JomegaUfunc()


def _do_Jsum(om, A_J, D_J, ctx=None):
    """spectral_densities.py:1961-1972."""
    return np.einsum('...j,jk', A_J, Jomega.outer(D_J, om, ctx=ctx))


def D_coefficients_symmtop(D):
    """spectral_densities.py:1874-1884."""
    return _nph.D_coefficients_symmtop(D)


def A_coefficients_symmtop(v, bProlate=True):
    """spectral_densities.py:1886-1906."""
    return _nph.A_coefficients_symmtop(v, bProlate)


# ---------------------------------------------------------------------------------------------------
# batched R1/R2/NOE/rho (calculate-relaxations-from-Ct.py:82-191)
# ---------------------------------------------------------------------------------------------------
def _pack(S2, consts, taus):
    n = len(S2)
    K = np.array([len(c) for c in consts], dtype=np.int32)
    Kmax = max(int(K.max()) if n else 1, 1)
    C = np.zeros((n, Kmax))
    T = np.ones((n, Kmax))
    for i in range(n):
        C[i, :K[i]] = consts[i]
        T[i, :K[i]] = taus[i]
    return np.asarray(S2, dtype=float), C, T, K


def _model_args(RObj):
    name = RObj.rotdifModel.name
    if name == 'direct_transform':
        return 0, None
    if name == 'rigid_sphere':
        return 1, [RObj.rotdifModel.D]
    if name == 'rigid_symmtop':
        return 2, [RObj.rotdifModel.D[0], RObj.rotdifModel.D[1]]
    print("= = ERROR: Unknown rotdifModel in the relaxation object used in calculations!", file=sys.stderr)
    return None, None


def _run(RObj, nSites, S2, consts, taus, vecXH, weights, CSAvaluesArray, want_J, ctx, weights_dev_ptr=None):
    model, D = _model_args(RObj)
    if model is None:
        return None, None, None
    S2a, C, T, K = _pack(S2, consts, taus)
    if CSAvaluesArray is None:
        CSAvaluesArray = np.repeat(RObj.gX.csa, nSites)
    fcsa = RObj.get_f_CSA(np.asarray(CSAvaluesArray, dtype=float))
    kw = {}
    dist = False
    if model == 2:
        vecXH = np.asarray(vecXH, dtype=float)
        if vecXH.ndim > 2:
            dist = True
            same = all(np.array_equal(vecXH[0], vecXH[i]) for i in range(1, min(len(vecXH), 4)))
            if not same or not np.array_equal(vecXH[0], vecXH[-1]):
                raise ValueError('GPU path expects the same bin-centre vectors for every residue (histogram input)')
            kw = dict(binvecs=vecXH[0], weights=weights, weights_dev_ptr=weights_dev_ptr)
        else:
            kw = dict(resvecs=vecXH)
    out, J = srdist.relax(_ctx(ctx), model, D, RObj.omega, RObj.get_f_DD(), fcsa[None, :], RObj.time_fact,
                             RObj.gH.gamma / RObj.gX.gamma, S2a, C, T, K, noe_mode=0, want_J=want_J, **kw)
    return out[0], (J[0] if J is not None else None), dist


def _obtain_R1R2NOErho(RObj, nSites, S2, consts, taus, vecXH, weights=None, CSAvaluesArray=None, ctx=None,
                       weights_dev_ptr=None):
    """calculate-relaxations-from-Ct.py:125-191.  Returns the float32 datablock (4, nSites) or, for a
    vector distribution, (4, nSites, 2) with [weighted mean, weighted sigma]."""
    out, _, dist = _run(RObj, nSites, S2, consts, taus, vecXH, weights, CSAvaluesArray, False, ctx, weights_dev_ptr)
    if out is None:
        return []
    if dist:
        return np.transpose(out, (1, 0, 2)).astype(np.float32)
    return out[:, :, 0].T.astype(np.float32)


def _obtain_Jomega(RObj, nSites, S2, consts, taus, vecXH, weights=None, ctx=None):
    """calculate-relaxations-from-Ct.py:82-122: float32 (5, nSites) or (5, nSites, 2)."""
    _, J, dist = _run(RObj, nSites, S2, consts, taus, vecXH, weights, None, True, ctx)
    if J is None:
        return []
    if dist:
        return np.transpose(J, (1, 0, 2)).astype(np.float32)
    return J[:, :, 0].T.astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# vector-distribution files
# ---------------------------------------------------------------------------------------------------
def convert_LambertCylindricalHist_to_vecs(hist, edges):
    """spectral_densities.py:2334-2350 / calculate-relaxations-from-Ct.py:405-421: bin-centre unit vectors
    (phi-major) and the histogram counts as weights.  Returns (binvecs (B,3), weights (nRes,B)); the
    reference repeats binvecs per residue -- the kernels take the shared copy."""
    binVecs = _nph.lambert_bin_vectors(edges)
    nResidues = hist.shape[0]
    return binVecs, np.reshape(hist, (nResidues, binVecs.shape[0]))


def read_vector_distribution_from_file(fileName):
    """calculate-relaxations-from-Ct.py:424-454 for the numpy formats written by calculate-Ct-from-traj.py.
    Returns resIDs, vecs, weights; for histograms vecs is (nRes, B, 3) (a broadcast view) like the reference."""
    if not fileName.endswith('.npz'):
        print("= = = ERROR: only the numpy (.npz) vector-distribution formats are supported by the GPU path!", file=sys.stderr)
        sys.exit(1)
    obj = np.load(fileName, allow_pickle=True)
    resIDs = obj['names']
    weights = None
    if obj['bHistogram']:
        if obj['dataType'] != 'LambertCylindrical':
            print("= = = Histogram projection not supported! %s" % obj['dataType'], file=sys.stderr)
            sys.exit(1)
        binVecs, weights = convert_LambertCylindricalHist_to_vecs(obj['data'], obj['edges'])
        vecs = np.broadcast_to(binVecs[np.newaxis, ...], (weights.shape[0],) + binVecs.shape)
    else:
        if obj['dataType'] != 'PhiTheta':
            print("= = = Numpy binary datatype not supported! %s" % obj['dataType'], file=sys.stderr)
            sys.exit(1)
        vecs = _nph.rtp_to_xyz_unit(obj['data'])
    if weights is not None:
        print("    ...converted input phi_theta data to vecXH / weights, whose shapes are:", vecs.shape, weights.shape)
    else:
        print("    ...converted input phi_theta data to vecXH, whose shape is:", vecs.shape)
    return resIDs, vecs, weights


# the new class API (angularFrequencies, globalRotationalDiffusion_*, spinRelaxationR1/R2/NOE,
# spinRelaxationExperiments) lives in spin_relaxation.py; re-exported here under the reference's module name
from .spin_relaxation import (angularFrequencies, gyromagMultiCSA, globalRotationalDiffusion_Base,      # noqa: E402,F401
                              globalRotationalDiffusion_Isotropic, globalRotationalDiffusion_Axisymmetric,
                              spinRelaxationBase, spinRelaxationR1, spinRelaxationR2, spinRelaxationNOE,
                              spinRelaxationExperiments)
