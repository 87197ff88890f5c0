"""
Host-side mirror of the reference's NEW class API (spectral_densities.py:81-124, 136-246, 253-603, 607-1447):
angularFrequencies, globalRotationalDiffusion_{Isotropic,Axisymmetric}, spinRelaxationR1/R2/NOE and the
multi-experiment container spinRelaxationExperiments with its global (Diso, Daniso, zeta, CSA) and
residue-specific (rsCSA) optimisation loops, as driven by calculate-relaxations-multi-field.py.

GPU mapping.  `eval_all()` evaluates EVERY experiment (field x type) for EVERY residue and histogram bin in one
batched launch (sr_jomega_relax_f64, noe_mode = 1: NOE from the vector-averaged R1, spectral_densities.py:881-892).
The rsCSA step does not re-evaluate J(omega) per objective call like the reference
(optimisation_loop_rsCSA_inner_function, :1430-1447): the CSA enters R1/R2 only through f_CSA ~ csa^2, so one
launch returns, per (experiment, residue), the weighted means / (co)variances of the CSA-free and CSA-proportional
parts, and the objective for any csa is a closed form of 12 numbers -- algebraically identical, not bitwise
(SURVEY.md section 7).  The optimisers themselves are the reference's: scipy.optimize.fmin_powell, called the same way.
"""
import sys

import numpy as np


def fmin_powell(*args, **kwargs):
    """scipy.optimize.fmin_powell, imported when an optimisation is really run: importing scipy.optimize costs 0.2 s, a third of
    the wall time of a plain calculate-relaxations-from-Ct.py run that never optimises anything"""
    from scipy.optimize import fmin_powell as f
    return f(*args, **kwargs)


from . import hip
from . import dist as srdist
from . import _hostmath as hm
from . import general_scripts as gs
from . import fitting_Ct_functions as fitCt
from .spectral_densities import GAMMA, CSA_DEFAULT, _return_time_fact, gyromag, convert_LambertCylindricalHist_to_vecs


def _ctx(ctx):
    return ctx if ctx is not None else hip.default_context()


def _BAIL(functionName, message):
    print("= = ERROR in function %s : %s" % (functionName, message), file=sys.stderr)
    sys.exit(1)


class gyromagMultiCSA(gyromag):
    """spectral_densities.py:81-124: one CSA per residue."""

    def __init__(self, isotope, n, csa=None):
        self.num = n
        self.isotope = isotope
        self.gamma = GAMMA[isotope]
        if csa is None:
            self.csa = np.repeat(CSA_DEFAULT.get(isotope, 0.0), n)
        else:
            self.set_csa(csa)

    def set_csa(self, csa, ind=None):
        if ind is None:
            if self.num != len(csa):
                print("= = ERROR: attempting to set CSA array in gyromagMultiCSA, but the lengths to not match!")
                sys.exit(1)
            self.csa = np.array(csa, dtype=float)
        else:
            self.csa[ind] = csa

    def get_csa(self, ind=None):
        return self.csa if ind is None else self.csa[ind]


class angularFrequencies:
    """spectral_densities.py:136-246."""

    def __init__(self, nucleiA='15N', nucleiB='1H', fieldStrength=600, fieldUnit='MHz', timeUnit='ps'):
        self.timeUnit = timeUnit
        self.time_fact = _return_time_fact(timeUnit)
        self.distUnit = 'nm'
        self.dist_fact = 1.0e-9
        self.gA = gyromag(nucleiA)
        self.gB = gyromag(nucleiB)
        self.rAB = 1.02e-1
        self.set_magnetic_field(fieldStrength, fieldUnit)
        self.nOmega = 5
        self.omega = np.zeros(5)
        self.omega[1] = -1.0 * self.gA.gamma * self.B0 * self.time_fact
        self.omega[3] = -1.0 * self.gB.gamma * self.B0 * self.time_fact
        self.omega[2] = self.omega[3] - self.omega[1]
        self.omega[4] = self.omega[3] + self.omega[1]

    def set_magnetic_field(self, inp, unit):
        if unit == 'Hz':
            self.B0 = 2.0 * np.pi * inp / 267.513e6
        elif unit == 'MHz':
            self.B0 = 2.0 * np.pi * inp / 267.513
        elif unit == 'T':
            self.B0 = inp
        else:
            _BAIL("set_magnetic_field", "incorrect field units given ( %s )" % unit)

    def get_magnetic_field(self, unit='T'):
        if unit == 'T':
            return self.B0
        if unit == 'MHz':
            return self.B0 * 267.513 / (2.0 * np.pi)
        if unit == 'Hz':
            return self.B0 * 267.513e6 / (2.0 * np.pi)
        _BAIL("get_magnetic_field", "incorrect field units given ( %s )" % unit)

    def get_factor_DD(self):
        """spectral_densities.py:239 (gamma_A^2 * gamma_B^2 order: 1 ulp from the old API's)."""
        return 0.10 * 1.1121216813552401e-82 * self.gA.gamma ** 2.0 * self.gB.gamma ** 2.0 * (self.rAB * self.dist_fact) ** -6.0

    def get_factor_CSA(self, i=None):
        return 2.0 / 15.0 * self.gA.get_csa(i) ** 2.0 * (self.gA.gamma * self.B0) ** 2

    def csa_prefactor(self):
        """f_CSA = csa^2 * csa_prefactor()."""
        return 2.0 / 15.0 * (self.gA.gamma * self.B0) ** 2

    def initialise_CSA_array(self, numCSAs, CSAvalues=None):
        self.gA = gyromagMultiCSA(self.gA.isotope, numCSAs, CSAvalues)

    def get_nuclei_names(self):
        return [self.gA.isotope, self.gB.isotope]


class globalRotationalDiffusion_Base:
    def __init__(self):
        self.name = 'base'
        self.D = None
        self.bVecs = False
        self.axisAvg = None
        self.vecNames = None
        self.binvecs = None        # (B, 3) bin-centre vectors shared by every residue
        self.vecWeights = None     # (nRes, B)

    def import_frame_vectors_npz(self, fileName):
        """spectral_densities.py:279-306 for the Lambert histogram files of calculate-Ct-from-traj.py."""
        obj = np.load(fileName, allow_pickle=True)
        if not obj['bHistogram'] or obj['dataType'] != 'LambertCylindrical':
            print("= = = Only Lambert-cylindrical histogram distributions are supported by the GPU path! %s" % obj['dataType'],
                  file=sys.stderr)
            sys.exit(1)
        self.binvecs, self.vecWeights = convert_LambertCylindricalHist_to_vecs(obj['data'], obj['edges'])
        self.vecNames = obj['names']
        self.bVecs = True
        self.axisAvg = 0

    def import_frame_vectors(self, fileName):
        if fileName.endswith('.npz'):
            return self.import_frame_vectors_npz(fileName)
        print("= = = ERROR: only .npz vector distributions are supported.", file=sys.stderr)
        sys.exit(1)

    def get_names(self):
        return [str(x) for x in self.vecNames]


class globalRotationalDiffusion_Isotropic(globalRotationalDiffusion_Base):
    """spectral_densities.py:388-461."""

    def __init__(self, D=None, tau=None):
        globalRotationalDiffusion_Base.__init__(self)
        self.name = 'isotropic'
        self.D = D if D is not None else 1.0 / (6.0 * tau)

    def get_Diso(self):
        return self.D

    def set_Diso(self, Diso):
        self.D = Diso

    def get_Daniso(self):
        return 1.0

    def set_Daniso(self, Daniso):
        return

    def kernel_model(self):
        return 1, [self.D]


class globalRotationalDiffusion_Axisymmetric(globalRotationalDiffusion_Base):
    """spectral_densities.py:463-603; stored as (Diso, Daniso); bConvert: D given as (Dpar, Dperp)."""

    def __init__(self, D=None, bConvert=False, tau=None, aniso=None):
        globalRotationalDiffusion_Base.__init__(self)
        self.name = 'axisymmetric'
        if D is not None:
            if bConvert:
                self.D = np.array([(2.0 * D[1] + D[0]) / 3.0, D[0] / D[1]], dtype=float)
            else:
                self.D = np.array(D, dtype=float)
        else:
            self.D = np.array([1.0 / (6.0 * tau), aniso], dtype=float)
        self.bProlate = bool(self.D[1] > 1)

    def set_Diso(self, Diso):
        self.D[0] = Diso

    def set_Daniso(self, Daniso):
        self.D[1] = Daniso

    def get_Diso(self):
        return self.D[0]

    def get_Daniso(self):
        return self.D[1]

    def transform_D(self):
        tmp = 3.0 * self.D[0] / (2.0 + self.D[1])
        return self.D[1] * tmp, tmp

    def kernel_model(self):
        Dpar, Dperp = self.transform_D()
        return 2, [Dpar, Dperp]


class spinRelaxationBase:
    """One experiment (type at one field): holds the last computed values / errors (spectral_densities.py:607-818)."""
    column = None

    def __init__(self, name, timeUnit='ps', angFreq=None, globalRotDif=None, localCtModels=None):
        self.name = name
        self.values = None
        self.errors = None
        self.timeUnit = timeUnit
        self.time_fact = _return_time_fact(timeUnit)
        self.angFreq = angFreq
        self.globalRotDif = globalRotDif
        self.localCtModels = localCtModels
        if globalRotDif is not None and localCtModels is not None:
            n = localCtModels.nModels
            self.values = np.zeros(n)
            if globalRotDif.axisAvg is not None:
                self.errors = np.zeros(n)

    def get_magnetic_field(self):
        return self.angFreq.get_magnetic_field()

    def get_num(self):
        return self.localCtModels.nModels

    def get_name(self):
        return self.name

    def get_description(self):
        return "%s Experiment at %sT over %i vectors" % (self.get_name(), self.get_magnetic_field(), self.get_num())

    def eval(self, ind=None, ctx=None):
        """Evaluate this experiment alone (the container's eval_all batches all experiments in one launch)."""
        tmp = spinRelaxationExperiments(self.globalRotDif, self.localCtModels)
        tmp.spinrelax = [self]
        tmp.numExpts = 1
        tmp.eval_all(ind=ind, ctx=ctx)
        return self.values if ind is None else self.values[ind]

    def get_suffix_from_conditions(self):
        return '_%s%s' % (self.angFreq.gA.isotope, self.angFreq.gB.isotope) + \
               '_%iMHz' % (round(self.angFreq.get_magnetic_field(unit='MHz'))) + '_%s' % (self.name)

    def print_metadata(self, style='stdout', fp=sys.stdout):
        if style == 'stdout':
            print("# %s" % self.get_description(), file=fp)
        elif style == 'xmgrace':
            print('# Type %s' % self.name, file=fp)
            print('# NucleiA %s' % (self.angFreq.gA.isotope), file=fp)
            print('# NucleiB %s' % (self.angFreq.gB.isotope), file=fp)
            print('# Frequency %g %s' % (self.angFreq.get_magnetic_field(unit='MHz'), 'MHz'), file=fp)

    def print_values(self, style='stdout', fp=sys.stdout):
        names = self.localCtModels.get_names()
        if style == 'xmgrace':
            print("@type xy" if self.errors is None else "@type xydy", file=fp)
        if self.errors is None:
            for x, y in zip(names, self.values):
                print("%s %g" % (x, y), file=fp)
        else:
            for x, y, dy in zip(names, self.values, self.errors):
                print("%s %g %g" % (x, y, dy), file=fp)
        print("&" if style == 'xmgrace' else '', file=fp)

    def calc_chisq(self, Target, dTarget=None, indices=None):
        """spectral_densities.py:803-818."""
        v, e = self.values, self.errors
        if indices is not None:
            v = v[indices]
            if e is not None:
                e = e[indices]
        if e is not None and dTarget is not None:
            return np.mean(np.square(v - Target) / (np.square(dTarget) + np.square(e)))
        if e is None:
            return np.mean(np.square(v - Target) / np.square(dTarget))
        return np.mean(np.square(v - Target) / np.square(e))


class spinRelaxationR1(spinRelaxationBase):
    column = 0


class spinRelaxationR2(spinRelaxationBase):
    column = 1


class spinRelaxationNOE(spinRelaxationBase):
    column = 2


_TYPES = {'R1': spinRelaxationR1, 'R2': spinRelaxationR2, 'NOE': spinRelaxationNOE}


class spinRelaxationExperiments:
    """spectral_densities.py:909-1447."""
    listAllowedOptimisationVariables = ['Diso', 'Daniso', 'CSA', 'zeta', 'rsCSA']
    dictStepSizes = {'Diso': 1e-5, 'Daniso': 0.1, 'zeta': 0.1, 'CSA': 1e-5, 'rsCSA': 1e-5}
    dictExportScaling = {'Diso': 1.0, 'Daniso': 1.0, 'zeta': 1.0, 'CSA': 1e6, 'rsCSA': 1e6}
    dictExportUnits = {'Diso': 'ps^-1', 'Daniso': 'a.u.', 'zeta': 'a.u.', 'CSA': 'ppm', 'rsCSA': 'ppm'}

    def __init__(self, globalRotDif=None, localCtModels=None, ctx=None):
        self.numExpts = 0
        self.spinrelax = []
        self.data = []
        self.globalRotDif = globalRotDif
        self.localCtModels = localCtModels
        self.mapModelNames = []
        self.mapExptCoverage = []
        self.bOptInitialised = False
        self.listUpdateVariables = []
        self.listStepSizes = []
        self.bDoLocalOpt = False
        self.bOptCompleted = False
        self.chisq = None
        self.ctx = ctx
        self.nObjectiveCalls = 0

    # ---- input ----
    def add_experiment(self, fileName, bIgnoreErrors=False):
        """spectral_densities.py:935-1010: '# Type|NucleiA|NucleiB|Frequency|FrequencyUnit' header, 'name value [error]' rows."""
        strType = nucleiA = nucleiB = freq = None
        freqUnit = 'MHz'
        names, values, errors = [], [], []
        for line in open(fileName, 'r'):
            l = line.split()
            if len(l) == 0:
                continue
            if line[0] == '#' or line[0] == '@':
                if len(l) > 2:
                    if l[1] == 'Type':
                        strType = l[2]
                    elif l[1] == 'NucleiA':
                        nucleiA = l[2]
                    elif l[1] == 'NucleiB':
                        nucleiB = l[2]
                    elif l[1] == 'Frequency':
                        freq = float(l[2])
                    elif l[1] == 'FrequencyUnit':
                        freqUnit = l[2]
                continue
            if len(l) == 1 or len(l) > 3:
                print("ERROR in spinRelaxationExperiments.add_experiment(): data line does not obey expected conventions of 2 or 3 "
                      "space-separated values!", l, file=sys.stderr)
                sys.exit()
            names.append(l[0])
            values.append(float(l[1]))
            errors.append(float(l[2]) if len(l) > 2 else None)
        if nucleiB is None and strType in ('R1', 'R2'):
            nucleiB = '1H'
        if strType is None or nucleiA is None or nucleiB is None or freq is None or strType not in _TYPES:
            print("ERROR in spinRelaxationExperiments.add_experiment(): not all metadata has been read! "
                  "Require: Type, NucleiA, NucleiB, Frequency", file=sys.stderr)
            sys.exit(1)
        nMissing = sum(x is None for x in errors)
        if nMissing == len(errors):
            errors = None
        elif nMissing > 0:
            print("ERROR in spinRelaxationExperiments.add_experiment(): either all entries must have uncertainties or none!", file=sys.stderr)
            sys.exit(1)
        else:
            errors = np.array(errors, dtype=float)
        wObj = angularFrequencies(nucleiA=nucleiA, nucleiB=nucleiB, fieldStrength=freq, fieldUnit=freqUnit)
        obj = _TYPES[strType](strType, angFreq=wObj, globalRotDif=self.globalRotDif, localCtModels=self.localCtModels)
        self.numExpts += 1
        self.spinrelax.append(obj)
        self.data.append(dict(names=np.array(names), y=np.array(values, dtype=float), dy=errors))

    def map_experiment_peaknames_to_models(self):
        """spectral_densities.py:1051-1091."""
        if self.localCtModels is None:
            print("ERROR in spinRelaxationExperiments.map_peak_names: need a local C(t) model to map experimental peak names!", file=sys.stderr)
            sys.exit(1)
        namesCt = self.localCtModels.get_names()
        if self.globalRotDif is not None and self.globalRotDif.bVecs:
            namesRotdif = self.globalRotDif.get_names()
            if len(namesCt) != len(namesRotdif) or any(a != b for a, b in zip(namesCt, namesRotdif)):
                print("ERROR in spinRelaxationExperiments.map_peak_names: local C(t) model and global rotational-diffusion model "
                      "do not have matching peak names!", file=sys.stderr)
                sys.exit(1)
        print("    ....mapping peak sets (residue IDs) between simulated localCtModels and expeirmental datasets")
        self.mapModelNames = []
        for i in range(self.numExpts):
            tmp = []
            for x in self.data[i]['names']:
                t = np.where(namesCt == x)[0]
                if len(t) > 0:
                    tmp.append(t[0])
            self.mapModelNames.append(tmp)
        self.mapExptCoverage = []
        for i in range(self.localCtModels.nModels):
            l = []
            for exptID in range(self.numExpts):
                ret = np.where(self.data[exptID]['names'] == namesCt[i])[0]
                if len(ret) > 0:
                    l.append((exptID, ret[0]))
            self.mapExptCoverage.append(l)

    def report_maps(self):
        print("Number of simulation residues covered by each experiment:", ''.join(' %i' % len(x) for x in self.mapModelNames))
        print("Number of Experiments covering each simulation residue:", ''.join(' %i' % len(x) for x in self.mapExptCoverage))

    def initialise_CSA_array(self, namesCSA, CSAValues):
        """spectral_densities.py:1103-1135 (identical name lists only)."""
        namesCSA = [str(x) for x in namesCSA]
        namesTest = [str(x) for x in self.localCtModels.get_names()]
        if namesCSA != namesTest:
            _BAIL("initialise_CSA_array", "The given CSA array must list the same residues as the fitted C(t) models.")
        for sp in self.spinrelax:
            sp.angFreq.initialise_CSA_array(len(CSAValues), CSAValues)

    # ---- global parameters ----
    def set_global_Diso(self, Diso):
        self.globalRotDif.set_Diso(Diso)

    def get_global_Diso(self):
        return self.globalRotDif.get_Diso()

    def set_global_Daniso(self, Daniso):
        self.globalRotDif.set_Daniso(Daniso)

    def get_global_Daniso(self):
        return self.globalRotDif.get_Daniso()

    def set_global_zeta(self, zeta):
        self.localCtModels.set_zeta(zeta)

    def get_global_zeta(self):
        return self.localCtModels.get_zeta()

    def set_all_csa(self, csa, ind=None):
        for sp in self.spinrelax:
            sp.angFreq.gA.set_csa(csa, ind)

    def get_first_csa(self, ind=None):
        return self.spinrelax[0].angFreq.gA.get_csa(ind)

    def _getter(self, name):
        return {'Diso': self.get_global_Diso, 'Daniso': self.get_global_Daniso, 'zeta': self.get_global_zeta,
                'CSA': self.get_first_csa}[name]

    def _setter(self, name):
        return {'Diso': self.set_global_Diso, 'Daniso': self.set_global_Daniso, 'zeta': self.set_global_zeta,
                'CSA': self.set_all_csa}[name]

    # ---- evaluation: ONE batched GPU launch for all experiments ----
    def _kernel_inputs(self):
        n = self.localCtModels.nModels
        E = len(self.spinrelax)
        S2, C, tau, K = self.localCtModels.get_params_as_arrays()
        zeta = self.localCtModels.get_zeta()
        om = np.array([sp.angFreq.omega for sp in self.spinrelax])
        fdd = np.array([sp.angFreq.get_factor_DD() for sp in self.spinrelax])
        fcsa = np.empty((E, n))
        for e, sp in enumerate(self.spinrelax):
            fcsa[e] = sp.angFreq.get_factor_CSA()
        tf = np.array([sp.time_fact for sp in self.spinrelax])
        gr = np.array([sp.angFreq.gB.gamma / sp.angFreq.gA.gamma for sp in self.spinrelax])
        model, D = self.globalRotDif.kernel_model()
        kw = {}
        if model == 2:
            if not self.globalRotDif.bVecs:
                _BAIL("eval_all", "the axisymmetric model needs a vector distribution (--distfn)")
            kw = dict(binvecs=self.globalRotDif.binvecs, weights=self.globalRotDif.vecWeights)
        return (model, D, om, fdd, fcsa, tf, gr, zeta * S2, zeta * C, tau, K), kw

    def eval_all(self, ind=None, bVerbose=False, ctx=None):
        """spectral_densities.py:1145-1157 for every experiment at once.  `ind`: update only that residue's entries."""
        args, kw = self._kernel_inputs()
        out, _ = srdist.relax(_ctx(ctx or self.ctx), *args, noe_mode=1, **kw)
        for e, sp in enumerate(self.spinrelax):
            v = out[e, :, sp.column, 0]
            err = out[e, :, sp.column, 1] if self.globalRotDif.axisAvg is not None else None
            if ind is None:
                sp.values = v.copy()
                sp.errors = None if err is None else err.copy()
            else:
                sp.values[ind] = v[ind]
                if err is not None:
                    sp.errors[ind] = err[ind]

    def get_all_values(self, ind=None):
        out = []
        for sp in self.spinrelax:
            v = sp.values if ind is None else sp.values[ind]
            e = None if sp.errors is None else (sp.errors if ind is None else sp.errors[ind])
            out.append([v, e] if e is not None else [v])
        return out

    def calc_chisq(self):
        """spectral_densities.py:1409-1413."""
        chisq = 0.0
        for i, sp in enumerate(self.spinrelax):
            chisq += sp.calc_chisq(self.data[i]['y'], self.data[i]['dy'], self.mapModelNames[i])
        return chisq / self.numExpts

    # ---- optimisation ----
    def parse_optimisation_params(self, listOpts):
        """spectral_densities.py:1269-1300."""
        self.bOptInitialised = False
        self.listUpdateVariables = []
        self.listStepSizes = []
        if 'CSA' in listOpts and 'rsCSA' in listOpts:
            _BAIL("parse_optimisation_params", "Cannot run both global CSA as well as residue-specific CSA optimisation!")
        for o in listOpts:
            if o not in spinRelaxationExperiments.listAllowedOptimisationVariables:
                _BAIL("parse_optimisation_params", "Optimisation variable %s not found in list!\\nPossibilities are: %s "
                      % (o, spinRelaxationExperiments.listAllowedOptimisationVariables))
            if o == 'rsCSA':
                csa = self.get_first_csa()
                if not type(csa) is np.ndarray:
                    print("    ... NOTE: CSA values have not been preset but residue-specific CSA optimisation is being performed. "
                          "Reinitialising CSA variables as being residue-specific.")
                    self.initialise_CSA_array(self.localCtModels.get_names(), np.repeat(csa, self.localCtModels.nModels))
                self.bDoLocalOpt = True
                continue
            self.listStepSizes.append(spinRelaxationExperiments.dictStepSizes[o])
            self.listUpdateVariables.append(o)
        self.bOptInitialised = True

    def optimisation_loop_get_globals(self):
        return [self._getter(o)() for o in self.listUpdateVariables]

    def optimisation_loop_set_globals(self, vNew):
        for o, v in zip(self.listUpdateVariables, vNew):
            self._setter(o)(v)

    def optimisation_loop_do_global_step(self):
        """spectral_densities.py:1360-1369; every objective call is one batched GPU evaluation."""
        def objective(params, *a):
            self.optimisation_loop_set_globals(np.atleast_1d(params))
            self.eval_all()
            self.nObjectiveCalls += 1
            return self.calc_chisq()
        direc = np.diag(np.array(self.listStepSizes, dtype=float))
        fminOut = fmin_powell(objective, x0=self.optimisation_loop_get_globals(), direc=direc, full_output=True, disp=False)
        print("= = = Optimisation complete over variables: %s" % self.listUpdateVariables)
        self.chisq = fminOut[1]

    def rscsa_statistics(self, ctx=None):
        """One launch: per (experiment, residue) the 12 sufficient statistics of the CSA dependence."""
        args, kw = self._kernel_inputs()
        _, _, stats = srdist.relax(_ctx(ctx or self.ctx), *args, noe_mode=1, want_stats=True, **kw)
        return stats

    def rscsa_closed_form(self, stats, e, i, csa):
        """(value, error) of experiment e for residue i at CSA `csa` from the statistics (see module docstring)."""
        sp = self.spinrelax[e]
        st = stats[e, i]
        f = csa * csa * sp.angFreq.csa_prefactor()
        has_err = self.globalRotDif.axisAvg is not None
        if sp.column == 0:
            v = st[0] + f * st[1]
            var = st[2] + 2.0 * f * st[3] + f * f * st[4]
        elif sp.column == 1:
            v = st[5] + f * st[6]
            var = st[7] + 2.0 * f * st[8] + f * f * st[9]
        else:
            R1 = st[0] + f * st[1]
            c = sp.time_fact * (sp.angFreq.gB.gamma / sp.angFreq.gA.gamma) / R1 * sp.angFreq.get_factor_DD()
            v = 1.0 + c * st[10]
            var = c * c * st[11]
        return v, (np.sqrt(max(var, 0.0)) if has_err else None)

    def rscsa_search_inputs(self):
        """Dense (E, n) target / uncertainty / coverage arrays and the per-experiment constants of the closed forms."""
        n = self.localCtModels.nModels
        E = len(self.spinrelax)
        y, dy, cover = np.zeros((E, n)), np.zeros((E, n)), np.zeros((E, n), dtype=np.uint8)
        for i, cov in enumerate(self.mapExptCoverage):
            for exptID, peakID in cov:
                cover[exptID, i] = 1
                y[exptID, i] = self.data[exptID]['y'][peakID]
                if self.data[exptID]['dy'] is not None:
                    dy[exptID, i] = self.data[exptID]['dy'][peakID]
        col = np.array([sp.column for sp in self.spinrelax], dtype=np.int32)
        pref = np.array([sp.angFreq.csa_prefactor() for sp in self.spinrelax])
        cnoe = np.array([sp.time_fact * (sp.angFreq.gB.gamma / sp.angFreq.gA.gamma) for sp in self.spinrelax])
        fdd = np.array([sp.angFreq.get_factor_DD() for sp in self.spinrelax])
        return col, pref, cnoe, fdd, y, dy, cover

    def optimisation_loop_do_local_step_host(self, stats, i0, nloc):
        """The host form of the search (scipy's fmin_powell called per residue, as in the reference): kept as the
        checker of the device search in tests/ -- the product path below never calls it."""
        for i in range(i0, i0 + nloc):
            cover = self.mapExptCoverage[i]
            if len(cover) == 0:
                continue

            def objective(params, *a):
                csa = float(np.atleast_1d(params)[0])
                self.set_all_csa(csa, ind=i)
                chisq = 0.0
                for exptID, peakID in cover:
                    v, dv = self.rscsa_closed_form(stats, exptID, i, csa)
                    sp = self.spinrelax[exptID]
                    sp.values[i] = v
                    if dv is not None:
                        sp.errors[i] = dv
                    else:
                        dv = 0.0
                    target = self.data[exptID]
                    dt = target['dy'][peakID] if target['dy'] is not None else 0.0
                    w = dv ** 2 + dt ** 2
                    if w == 0:
                        w = 1
                    chisq += (v - target['y'][peakID]) ** 2 / w
                self.nObjectiveCalls += 1
                return chisq / len(cover)
            fmin_powell(objective, x0=self.get_first_csa(ind=i), direc=[spinRelaxationExperiments.dictStepSizes['rsCSA']],
                        full_output=False, disp=False)

    def optimisation_loop_do_local_step(self, ctx=None):
        """spectral_densities.py:1371-1382 + 1430-1447: per-residue 1-D Powell on the rsCSA objective, every residue's
        search in ONE launch (sr_rscsa_search_f64: a thread per residue walks scipy's Powell / Brent iteration over the
        closed forms of the 12 statistics).  Like the reference, the CSA kept for a residue is the one of the LAST
        objective evaluation (the reference ignores fmin_powell's return value and relies on the side effect of
        set_all_csa inside the objective), and the values / errors of the covered experiments are the ones of that
        evaluation."""
        stats = self.rscsa_statistics()
        n = self.localCtModels.nModels
        i0, nloc = srdist.my_range(n)            # several ranks: each searches its residues, results gathered below
        sl = slice(i0, i0 + nloc)
        col, pref, cnoe, fdd, y, dy, cover = self.rscsa_search_inputs()
        has_err = self.globalRotDif.axisAvg is not None
        csa_all = np.array(self.get_first_csa(), dtype=float)
        if nloc > 0:
            csa, vals, errs, _, nfev = _ctx(ctx or self.ctx).rscsa_search(
                stats[:, sl], col, pref, cnoe, fdd, y[:, sl], dy[:, sl], cover[:, sl], has_err, csa_all[sl],
                spinRelaxationExperiments.dictStepSizes['rsCSA'])
            self.nObjectiveCalls += int(nfev.sum())
            csa_all[sl] = csa
            for e, sp in enumerate(self.spinrelax):
                m = np.nonzero(cover[e, sl])[0]
                sp.values[i0 + m] = vals[e, m]
                if has_err:
                    sp.errors[i0 + m] = errs[e, m]
        if srdist.world() > 1:
            csa_all = srdist.gather_rows(csa_all[sl], n)
            for sp in self.spinrelax:
                sp.values = srdist.gather_rows(np.asarray(sp.values)[sl], n)
                if sp.errors is not None:
                    sp.errors = srdist.gather_rows(np.asarray(sp.errors)[sl], n)
        for sp in self.spinrelax:
            sp.angFreq.gA.set_csa(csa_all)

    def perform_optimisation(self, maxCycles=10, tol=1e-6):
        """spectral_densities.py:1302-1358."""
        if not self.bOptInitialised:
            _BAIL("perform_fit", "You must first run parse_optimisation_params to tell the script what to optimise.")
        bDoGlobalOpt = len(self.listUpdateVariables) > 0
        if bDoGlobalOpt and not self.bDoLocalOpt:
            self.optimisation_loop_do_global_step()
            self.bOptCompleted = True
            return self.chisq
        if self.bDoLocalOpt and not type(self.get_first_csa()) is np.ndarray:
            _BAIL("perform_optimisation", "CSA values are not an array for local optimisation!")
        if self.bDoLocalOpt and not bDoGlobalOpt:
            self.eval_all()
            self.optimisation_loop_do_local_step()
            self.bOptCompleted = True
            self.chisq = self.calc_chisq()
            return self.chisq
        if bDoGlobalOpt and self.bDoLocalOpt:
            bFirst = True
            for n in range(maxCycles):
                paramPrev = self.optimisation_loop_get_globals()
                self.optimisation_loop_do_global_step()
                paramNow = self.optimisation_loop_get_globals()
                if not bFirst and np.allclose(paramPrev, paramNow, rtol=tol):
                    self.bOptCompleted = True
                    break
                csaPrev = np.copy(self.get_first_csa())
                self.optimisation_loop_do_local_step()
                csaNow = self.get_first_csa()
                if not bFirst and np.allclose(csaPrev, csaNow, rtol=tol):
                    self.chisq = self.calc_chisq()
                    self.bOptCompleted = True
                    break
                bFirst = False
            return self.chisq
        _BAIL("perform_optimisation", "neither global or local optimisation have been successfully specified!")

    # ---- output ----
    def print_parameters(self, style='stdout', fp=sys.stdout):
        """spectral_densities.py:1224-1242."""
        for x in spinRelaxationExperiments.listAllowedOptimisationVariables:
            if x == 'rsCSA':
                continue
            v = self._getter(x)()
            s1 = 'Optimised' if x in self.listUpdateVariables else 'Fixed'
            if x == 'CSA' and type(v) is np.ndarray:
                v = np.mean(v)
                s1 = 'OptimisedMean' if (self.bOptCompleted and self.bDoLocalOpt) else 'FixedMean'
            print('# %s %s: %g %s' % (s1, x, v * spinRelaxationExperiments.dictExportScaling[x],
                                      spinRelaxationExperiments.dictExportUnits[x]), file=fp)
        if self.bOptCompleted:
            print('# Optimised chi: %g a.u.' % np.sqrt(self.chisq), file=fp)

    def print_experiment_data(self, ind, style='stdout', fp=sys.stdout):
        d = self.data[ind]
        if style == 'xmgrace':
            print('@type xy' if d['dy'] is None else '@type xydy', file=fp)
        if d['dy'] is None:
            for x, y in zip(d['names'], d['y']):
                print("%s %g" % (x, y), file=fp)
        else:
            for x, y, dy in zip(d['names'], d['y'], d['dy']):
                print("%s %g %g" % (x, y, dy), file=fp)
        print('&' if style == 'xmgrace' else '', file=fp)

    def export_xvg(self, filePrefix, bIncludeExpt=False):
        """spectral_densities.py:1178-1194: one <prefix>_<15N1H>_<MHz>MHz_<Type>.xvg per experiment."""
        for i, sp in enumerate(self.spinrelax):
            with open('%s%s.xvg' % (filePrefix, sp.get_suffix_from_conditions()), 'w') as fp:
                sp.print_metadata('xmgrace', fp)
                self.print_parameters('xmgrace', fp)
                print('', file=fp)
                print('@target s0', file=fp)
                sp.print_values('xmgrace', fp)
                if bIncludeExpt:
                    print('@target s1', file=fp)
                    self.print_experiment_data(ind=i, style='xmgrace', fp=fp)
