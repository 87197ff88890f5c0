"""
Host-side mirror of the reference's fitting_Ct_functions.py: the containers for fitted
autocorrelation models (same class / method / attribute names, same `_fittedCt.dat` text format) and
the model-order search of optimised_curve_fitting.  Every least-squares solve runs on the GPU
(sr_expfit_lm_f64: batched trust-region-reflective fit, one wave per residue); this module holds the
host logic around it: initial guesses, bounds, quality flags and the accept/reject sequence, all
restated in the reference's operation order including its quirks.

Reference: fitting_Ct_functions.py:12-481.
"""
import sys
from collections import OrderedDict

import numpy as np

from . import hip
from . import dist as srdist

GREEK = np.array(['a', 'b', 'g', 'd', 'e', 'z', 'h'])      # fitting_Ct_functions.py:136


def _ctx(ctx):
    return ctx if ctx is not None else hip.default_context()


# ---------------------------------------------------------------------------------------------------
# pure host logic (unit-tested without a GPU)
# ---------------------------------------------------------------------------------------------------
def split_nparams(nParams):
    """set_nParams, fitting_Ct_functions.py:376-382 -> (nComps, bS2Fast)."""
    return int(nParams / 2), (nParams % 2 == 1)


def initial_guess(DeltaT, Decay, nParams, nSample=10):
    """initialise_for_fit_advanced, fitting_Ct_functions.py:359-374.  Returns the p0 list in the order
    get_params_as_list produces ([C..., tau..., (S2)]) plus the (C list, S2) the later quality check uses."""
    K, free = split_nparams(nParams)
    tau = np.logspace(np.log10(np.mean(DeltaT[1:] - DeltaT[:-1])), np.log10(DeltaT[-1] * 2.0), K + 2)[1:-1]
    avgBeg = np.mean(Decay[:nSample])
    avgEnd = np.mean(Decay[-nSample:])
    C = [np.fabs(avgBeg - avgEnd) / K] * K
    S2 = avgEnd if free else 1.0 - np.mean(C)
    p0 = list(C) + list(tau) + ([S2] if free else [])
    return np.array(p0, dtype=float), C, S2


def quality_flags(ok, popt, dParam, C0, S2_0, bS2Fast):
    """The three flags of conduct_curve_fitting (fitting_Ct_functions.py:320-338).  Quirk kept: the
    "sum > 1" test runs before the optimum is stored, i.e. on the INITIAL guess (C0, S2_0), with S2
    overwritten by 1 - sum(C0) when it is not a free parameter (:330-331)."""
    if not ok:
        return [False, True, True]
    q = [True, True, True]
    with np.errstate(invalid='ignore'):
        if np.any(dParam > popt):
            q[1] = False
    S2 = S2_0 if bS2Fast else 1.0 - np.sum(C0)
    if S2 + np.sum(C0) > 1.0:
        q[2] = False
    return q


class OrderSearch:
    """State machine of optimised_curve_fitting (fitting_Ct_functions.py:278-304) for ONE residue, fed one
    model order at a time so that many residues can advance in lock-step over batched GPU fits."""

    def __init__(self, chiSqThreshold=0.5):
        self.thr = chiSqThreshold
        self.first = True
        self.best = None          # dict of the accepted fit
        self.last = None          # last fit attempted (what the reference leaves in `self` on total failure)
        self.done = False

    def feed(self, fit):
        """fit: dict(chiSq, quality, ...).  Returns True while further orders should be tried."""
        if self.done:
            return False
        self.last = fit
        if self.first:
            if np.all(fit['quality']):
                self.best = fit
                self.first = False
            return True
        if not np.all(fit['quality']):
            self.done = True
            return False
        if fit['chiSq'] >= self.best['chiSq'] * self.thr:
            self.done = True
            return False
        self.best = fit
        return True

    def result(self):
        return self.best


# ---------------------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------------------
class autoCorrelationModel:
    """One residue's C(t) = S2 + sum_k C_k exp(-t/tau_k); see fitting_Ct_functions.py:128-416."""
    dictGreek = GREEK

    def __init__(self, name='Fit', listC=[], listTau=[], S2=None, bS2Fast=False, bSort=True):
        self.name = name
        self.tau = np.array(listTau, dtype=float)
        self.C = np.array(listC, dtype=float)
        self.bS2Fast = bS2Fast
        self.S2 = S2
        self.nComps = len(self.C)
        self.nParams = len(self.C) + len(self.tau) + (1 if bS2Fast else 0)
        self.bHasFit = False
        self.zeta = 1.0
        if bS2Fast and self.S2 is None:
            print("= = = ERROR: S2 must be given in fitPatam initialisation is bS2Fast is set to True!")
            sys.exit(1)
        if self.S2 is None:
            self.S2 = 1.0 - np.sum(self.C)
        self.check_consistency()
        if self.nComps > 1 and bSort:
            self.sort_components()

    def check_consistency(self):
        if self.nComps < 1:
            return
        if len(self.C) != len(self.tau):
            print("= = = ERROR: transient components in fitParam initialisation do not have matching number of parameters!")
            sys.exit(1)
        if not self.bS2Fast:
            if not np.all(np.isclose(self.S2 + np.sum(self.C), 1.0, rtol=1e-6)):
                print("= = = ERROR: Contribution of components in fitParam initialisation do not sum sufficeintly close to 1.00!")
                sys.exit(1)

    def set_nParams(self, n):
        self.nParams = n
        self.nComps, self.bS2Fast = split_nparams(n)

    def calc_S2Fast(self):
        return 1.0 - self.S2 - np.sum(self.C) if self.bS2Fast else 0.0

    def sort_components(self):
        inds = np.argsort(self.tau)
        self.tau = self.tau[inds]
        self.C = self.C[inds]
        if self.bHasFit:
            self.dtau = self.dtau[inds]
            self.dC = self.dC[inds]

    def set_zeta(self, zeta):
        self.zeta = zeta

    def get_zeta(self):
        return self.zeta

    def get_params_as_list(self):
        return list(self.C) + list(self.tau) + ([self.S2] if self.bS2Fast else [])

    def get_uncertainties_as_list(self):
        return list(self.dC) + list(self.dtau) + ([self.dS2] if self.bS2Fast else [])

    def eval(self, DeltaT):
        """fitting_Ct_functions.py:266-270 (host evaluation for the exported model curve)."""
        return self.zeta * (self.S2 + np.sum(self.C[:, np.newaxis] * np.exp(-1.0 * DeltaT[np.newaxis, :] / self.tau[:, np.newaxis]), axis=0))

    def calc_chiSq(self, DeltaT, Decay, dDecay=None):
        """fitting_Ct_functions.py:272-276: sigma, not sigma^2, in the denominator."""
        if dDecay is None:
            return np.mean(np.square(self.eval(DeltaT) - Decay))
        return np.mean(np.square(self.eval(DeltaT) - Decay) / dDecay)

    def _load_fit(self, fit):
        """Install a fit dict produced by fit_batch (already component-sorted)."""
        self.set_nParams(fit['nParams'])
        self.C = np.array(fit['C'], dtype=float)
        self.tau = np.array(fit['tau'], dtype=float)
        self.S2 = fit['S2']
        self.dC = np.array(fit['dC'], dtype=float)
        self.dtau = np.array(fit['dtau'], dtype=float)
        self.dS2 = fit['dS2']
        self.chiSq = fit['chiSq']
        self.bHasFit = True

    def conduct_curve_fitting(self, DeltaT, Decay, dDecay=None, bReInitialise=False, fp=sys.stdout, ctx=None):
        """fitting_Ct_functions.py:306-345 for the current nParams (GPU solve, batch of one)."""
        fit = fit_batch(np.asarray(DeltaT)[None], np.asarray(Decay)[None], None if dDecay is None else np.asarray(dDecay)[None],
                        self.nParams, ctx=ctx)[0]
        if not fit['ok']:
            print("= = = WARNING, curve fitting of %s with %i params failed!" % (self.name, self.nParams), file=fp)
            return np.inf, fit['quality']
        if not fit['quality'][1]:
            print("= = = WARNING, curve fitting of %s with %i params indicates overfitting." % (self.name, self.nParams), file=fp)
        if not fit['quality'][2]:
            print("= = = WARNING, curve fitting of %s with %i params returns sum>1." % (self.name, self.nParams), file=fp)
        self._load_fit(fit)
        return self.chiSq, fit['quality']

    def optimised_curve_fitting(self, DeltaT, Decay, dDecay=None, listDoG=[2, 3, 5, 7, 9], chiSqThreshold=0.5, fp=sys.stdout,
                                ctx=None):
        """fitting_Ct_functions.py:278-304 for one residue (batch of one on the GPU)."""
        res = optimised_curve_fitting_batch([self.name], np.asarray(DeltaT)[None], np.asarray(Decay)[None],
                                            None if dDecay is None else np.asarray(dDecay)[None], listDoG, chiSqThreshold,
                                            fp=fp, ctx=ctx)[0]
        if res is not None:
            self._load_fit(res)
            return self.chiSq
        return np.inf

    def report(self, style='stdout', fp=sys.stdout):
        """fitting_Ct_functions.py:224-264; the 'xmgrace' style is the header block of _fittedCt.dat."""
        g = autoCorrelationModel.dictGreek
        if style == 'stdout':
            print("Name: %s" % self.name, file=fp)
            if self.bHasFit:
                print('  chi-Square: %g ' % self.chiSq, file=fp)
            if self.bS2Fast:
                print("  S2_fast: %g" % self.calc_S2Fast(), file=fp)
            for i in range(self.nComps):
                if self.bHasFit:
                    print("  component %s, const.: %g +- %g" % (g[i], self.C[i], self.dC[i]), file=fp)
                    print("  component %s, tau: %g +- %g" % (g[i], self.tau[i], self.dtau[i]), file=fp)
                else:
                    print("  component %s, const.: %g " % (g[i], self.C[i]), file=fp)
                    print("  component %s, tau: %g " % (g[i], self.tau[i]), file=fp)
            if self.bHasFit:
                print("  S2_0: %g +- %g" % (self.S2, self.dS2), file=fp)
            else:
                print("  S2_0: %g" % self.S2, file=fp)
        elif style == 'xmgrace':
            print('# Residue: %s ' % self.name, file=fp)
            if self.bHasFit:
                print('# Chi-Square: %g ' % self.chiSq, file=fp)
                if self.bS2Fast:
                    print('# Param S2_fast: %g +- 0.0' % self.calc_S2Fast(), file=fp)
                    print('# Param S2_0: %g +- %g' % (self.S2, self.dS2), file=fp)
                else:
                    print('# Param S2_0: %g +- 0.0' % self.S2, file=fp)
                for i in range(self.nComps):
                    print('# Param C_%s: %g +- %g' % (g[i], self.C[i], self.dC[i]), file=fp)
                    print('# Param tau_%s: %g +- %g' % (g[i], self.tau[i], self.dtau[i]), file=fp)
            else:
                if self.bS2Fast:
                    print('# Param S2_fast: %g' % self.calc_S2Fast(), file=fp)
                print('# Param S2_0: %g' % self.S2, file=fp)
                for i in range(self.nComps):
                    print('# Param C_%s: %g' % (g[i], self.C[i]), file=fp)
                    print('# Param tau_%s: %g' % (g[i], self.tau[i]), file=fp)
        else:
            print("= = = ERROR: fitParam.report() does not recognise the style argument! Choices are: stdout, xmgrace",
                  file=sys.stderr)


class autoCorrelations:
    """Set of models plus their target curves; see fitting_Ct_functions.py:12-126."""

    def __init__(self):
        self.nModels = 0
        self.model = OrderedDict()
        self.nTargets = 0
        self.DeltaT = OrderedDict()
        self.Decay = OrderedDict()
        self.dDecay = OrderedDict()

    def get_names(self):
        return np.array([k for k in self.model.keys()])

    def get_params_as_list(self):
        keys = self.model.keys()
        return ([self.model[k].S2 for k in keys], [self.model[k].C for k in keys], [self.model[k].tau for k in keys],
                [self.model[k].calc_S2Fast() for k in keys])

    def get_params_as_arrays(self, Kmax=None):
        """Padded arrays for the batched GPU kernels: S2 (n,), C (n,Kmax), tau (n,Kmax), nComps (n,)."""
        n = self.nModels
        K = np.array([m.nComps for m in self.model.values()], dtype=np.int32)
        Kmax = max(int(K.max()) if n else 1, 1) if Kmax is None else Kmax
        C = np.zeros((n, Kmax))
        tau = np.ones((n, Kmax))
        S2 = np.zeros(n)
        for i, m in enumerate(self.model.values()):
            C[i, :K[i]] = m.C
            tau[i, :K[i]] = m.tau
            S2[i] = m.S2
        return S2, C, tau, K

    def set_zeta(self, zeta):
        for m in self.model.values():
            m.set_zeta(zeta)

    def get_zeta(self):
        for m in self.model.values():
            return m.get_zeta()

    def add_model(self, key, name=None, listC=[], listTau=[], S2=None, bS2Fast=False, bSort=True):
        self.model[key] = autoCorrelationModel(key if name is None else name, listC, listTau, S2, bS2Fast, bSort)
        self.nModels = len(self.model)
        return self.model[key]

    def get_nth_model(self, n):
        return self.model[self.get_names()[n]]

    def add_target(self, key, DeltaT, Decay, dDecay):
        self.DeltaT[key] = DeltaT
        self.Decay[key] = Decay
        self.dDecay[key] = dDecay
        self.nTargets = len(self.DeltaT)

    def import_target_array(self, keys, DeltaT, Decay, dDecay=None):
        for i, k in enumerate(keys):
            self.add_target(k, DeltaT[i], Decay[i], None if dDecay is None else dDecay[i])

    def report(self):
        print("Number of C(t) models loaded:", self.nModels)
        print("Number of targets loaded:", self.nTargets)

    def fit_all(self, listDoG=[2, 3, 5, 7, 9], chiSqThreshold=0.5, nc=-1, bUseSFast=True, fp=sys.stdout, ctx=None):
        """The per-residue loop of calculate-fitted-Ct.py:161-178 as ONE GPU launch: every residue runs through the
        model orders, quality checks and the accept / reject rule on the device (sr_expfit_order_search_f64)."""
        keys = list(self.DeltaT.keys())
        t = np.array([self.DeltaT[k] for k in keys], dtype=float)
        y = np.array([self.Decay[k] for k in keys], dtype=float)
        dy = None if self.dDecay[keys[0]] is None else np.array([self.dDecay[k] for k in keys], dtype=float)
        if nc == -1:
            results = optimised_curve_fitting_batch(keys, t, y, dy, listDoG, chiSqThreshold, fp=fp, ctx=ctx)
        else:
            nP = 2 * nc + 1 if bUseSFast else 2 * nc
            results = [f if f['ok'] else None for f in fit_batch(t, y, dy, nP, ctx=ctx)]
        for k, res in zip(keys, results):
            obj = self.add_model(k)
            if res is not None:
                obj._load_fit(res)
        return results

    def export(self, fileName, style='xmgrace'):
        """fitting_Ct_functions.py:107-126: header block, fitted curve, raw curve per residue."""
        from .general_scripts import _native_lib
        lib = _native_lib()
        keys = list(self.model.keys())
        curves = [self.model[k].eval(self.DeltaT[k]) for k in keys]
        # every "%8g %8g" row of the file in ONE threaded call of the library (same C formatting), cut at the block boundaries
        blocks = None
        if lib is not None and keys:
            A = np.concatenate([np.concatenate((np.asarray(self.DeltaT[k], dtype=np.float64)[:len(c)],) * 2) for k, c in zip(keys, curves)])
            B = np.concatenate([np.concatenate((np.asarray(c, dtype=np.float64), np.asarray(self.Decay[k], dtype=np.float64)[:len(c)]))
                                for k, c in zip(keys, curves)])
            if np.isfinite(A).all() and np.isfinite(B).all():
                import ctypes
                bounds = np.cumsum([0] + [len(c) for c in curves for _ in (0, 1)]).astype(np.int64)
                offs = np.empty(bounds.size, dtype=np.int64)
                buf = ctypes.create_string_buffer(32 * max(1, A.size))
                n = lib.sr_text_format_g8_pairs(A.ctypes.data, B.ctypes.data, A.size, buf, 32 * A.size, 0, bounds.ctypes.data, bounds.size,
                                                offs.ctypes.data)
                if n >= 0:
                    text = buf.raw[:n].decode('ascii')
                    blocks = [text[offs[i]:offs[i + 1]] for i in range(bounds.size - 1)]
        with open(fileName, 'w') as fp:
            s = 0
            for i, k in enumerate(keys):
                m = self.model[k]
                m.report(style='xmgrace', fp=fp)
                dt = self.DeltaT[k]
                Ct = self.Decay[k]
                ymodel = curves[i]
                print("@s%d legend \"Res %d\"" % (s, m.name), file=fp)
                if blocks is not None:
                    fp.write(blocks[2 * i])
                else:
                    for j in range(len(ymodel)):
                        print("%8g %8g" % (dt[j], ymodel[j]), file=fp)
                print('&', file=fp)
                if blocks is not None:
                    fp.write(blocks[2 * i + 1])
                else:
                    for j in range(len(ymodel)):
                        print("%8g %8g" % (dt[j], Ct[j]), file=fp)
                print('&', file=fp)
                s += 2


# ---------------------------------------------------------------------------------------------------
# GPU-backed batched fitting
# ---------------------------------------------------------------------------------------------------
def initial_guess_batch(t, y, nParams, nSample=10):
    """Vectorised initial_guess for (n, L) arrays: identical floating-point results to the per-residue
    function (same numpy reductions over the same contiguous runs).  Returns p0 (n, P), C0 (n, K), S2_0 (n,)."""
    n = y.shape[0]
    K, free = split_nparams(nParams)
    if n > 1 and np.all(t == t[0]):
        tau1 = np.logspace(np.log10(np.mean(t[0, 1:] - t[0, :-1])), np.log10(t[0, -1] * 2.0), K + 2)[1:-1]
        tau = np.broadcast_to(tau1, (n, K))
    else:
        tau = np.array([np.logspace(np.log10(np.mean(t[i, 1:] - t[i, :-1])), np.log10(t[i, -1] * 2.0), K + 2)[1:-1]
                        for i in range(n)]).reshape(n, K)
    avgBeg = np.mean(y[:, :nSample], axis=1)
    avgEnd = np.mean(y[:, -nSample:], axis=1)
    C0 = np.repeat((np.fabs(avgBeg - avgEnd) / K)[:, None], K, axis=1)
    S2_0 = avgEnd if free else 1.0 - np.mean(C0, axis=1)
    p0 = np.concatenate([C0, tau] + ([S2_0[:, None]] if free else []), axis=1)
    return np.ascontiguousarray(p0), C0, S2_0


def host_runner(t, y, dy, ctx=None):
    """Runner over host arrays: runner(nParams, p0, idx) -> popt, dP, chi, status for the residues idx."""
    def run(nParams, p0, idx):
        tmax = t[idx, -1]
        popt = np.empty((idx.size, nParams))
        dP = np.empty((idx.size, nParams))
        chi = np.empty(idx.size)
        status = np.empty(idx.size, dtype=np.int32)
        for tm in np.unique(tmax):                       # tau bound = 10 * t_max (fitting_Ct_functions.py:324)
            sel = np.flatnonzero(tmax == tm)
            sub = idx[sel]
            po, pc, ch, st, _ = _ctx(ctx).expfit(t[sub], y[sub], None if dy is None else dy[sub], p0[sel], tm * 10)
            popt[sel] = po
            with np.errstate(invalid='ignore'):
                dP[sel] = np.sqrt(np.diagonal(pc, axis1=1, axis2=2))
            chi[sel] = ch
            status[sel] = st
        return popt, dP, chi, status
    return run


def tau_guesses(t, listDoG):
    """The log-spaced tau part of initialise_for_fit_advanced (fitting_Ct_functions.py:361-362) for every order of
    listDoG, concatenated: (1, sum K) when every residue shares its time axis, else (n, sum K).  Same numpy calls as
    initial_guess, so the device search starts from bit-identical guesses."""
    t = np.atleast_2d(np.asarray(t, dtype=float))
    rows = t[:1] if (t.shape[0] == 1 or np.all(t == t[0])) else t
    out = []
    for ti in rows:
        row = []
        for nP in listDoG:
            K = int(nP / 2)
            row.append(np.logspace(np.log10(np.mean(ti[1:] - ti[:-1])), np.log10(ti[-1] * 2.0), K + 2)[1:-1])
        out.append(np.concatenate(row))
    return np.ascontiguousarray(out)


def order_search_device(t, y, dy, listDoG=(2, 3, 5, 7, 9), chiSqThreshold=0.5, ctx=None):
    """optimised_curve_fitting for all residues in ONE kernel launch (model orders, quality flags and the accept /
    reject rule evaluated on the GPU).  Returns the dict of Context.order_search."""
    t = np.atleast_2d(np.asarray(t, dtype=float))
    y = np.atleast_2d(np.asarray(y, dtype=float))
    if t.shape[0] == 1 and y.shape[0] > 1:
        t = np.ascontiguousarray(np.broadcast_to(t, y.shape))
    tau_max = t[0, -1] * 10                      # fitting_Ct_functions.py:324 (per-residue axes share their end point)
    if not np.all(t[:, -1] == t[0, -1]):
        raise ValueError('order_search_device: residues with different final times need separate calls (tau bound)')
    tg = tau_guesses(t, listDoG)
    n = y.shape[0]
    if srdist.world() == 1:
        return _ctx(ctx).order_search(t, y, dy, listDoG, tg, tau_max, chiSqThreshold)
    # several ranks (torchrun): residues are independent, every rank solves its contiguous range and all ranks receive
    # all results (no data-path collective, one gather of the results)
    i0, nloc = srdist.my_range(n)
    sl = slice(i0, i0 + nloc)
    if nloc > 0:
        loc = _ctx(ctx).order_search(t[sl], y[sl], None if dy is None else np.asarray(dy)[sl], listDoG,
                                     tg if np.ndim(tg) < 2 or tg.shape[0] == 1 else tg[sl], tau_max, chiSqThreshold)
    else:
        nO, Pmax = len(listDoG), max(listDoG)
        loc = dict(popt=np.empty((nO, 0, Pmax)), dP=np.empty((nO, 0, Pmax)), chisq=np.empty((nO, 0)),
                   status=np.empty((nO, 0), dtype=np.int32), nfev=np.empty((nO, 0), dtype=np.int32), best=np.empty(0, dtype=np.int32),
                   S2=np.empty(0), C=np.empty((0, Pmax // 2)), tau=np.empty((0, Pmax // 2)), chi=np.empty(0), K=np.empty(0, dtype=np.int32))
    out = {}
    for k, v in loc.items():
        if k == 'orders':
            out[k] = v
            continue
        out[k] = srdist.gather_rows(v, n, axis=1 if k in ('popt', 'dP', 'chisq', 'status', 'nfev') else 0)
    out.setdefault('orders', np.ascontiguousarray(listDoG, dtype=np.int32))
    return out


def order_search_device_results(t, y, dy, listDoG=(2, 3, 5, 7, 9), chiSqThreshold=0.5, ctx=None):
    """order_search_device, returned in the shape order_search_batch uses: (best (n,), list of per-order batch
    results with the initial guesses and the three quality flags re-derived on the host for the report)."""
    t = np.atleast_2d(np.asarray(t, dtype=float))
    y = np.atleast_2d(np.asarray(y, dtype=float))
    if t.shape[0] == 1 and y.shape[0] > 1:
        t = np.ascontiguousarray(np.broadcast_to(t, y.shape))
    dev = order_search_device(t, y, dy, listDoG, chiSqThreshold, ctx)
    per_order = []
    for j, nP in enumerate(listDoG):
        tried = dev['status'][j] != -100
        if not tried.any():
            break
        req = fit_request(t, y, nP, tried)
        idx = req['idx']
        per_order.append(fit_collect(req, dev['popt'][j][idx, :nP], dev['dP'][j][idx, :nP], dev['chisq'][j][idx],
                                     dev['status'][j][idx]))
    return dev['best'].astype(int), per_order


def fit_request(t, y, nParams, active=None):
    """First half of conduct_curve_fitting(bReInitialise=True) for a batch: the residues to solve and their
    initial guesses.  Returns dict(nParams, idx, p0, C0, S2_0)."""
    n = y.shape[0]
    idx = np.arange(n) if active is None else np.flatnonzero(active)
    if idx.size == 0:
        return dict(nParams=nParams, idx=idx, p0=np.empty((0, nParams)), C0=None, S2_0=None, n=n)
    p0, C0, S2_0 = initial_guess_batch(t[idx], y[idx], nParams)
    return dict(nParams=nParams, idx=idx, p0=p0, C0=C0, S2_0=S2_0, n=n)


def fit_collect(req, popt, dP, chi, status):
    """Second half: quality flags (fitting_Ct_functions.py:320-338; the sum>1 test runs on the initial guess --
    reference quirk) and the per-residue arrays over ALL n residues (rows of inactive residues undefined)."""
    nParams, idx, n = req['nParams'], req['idx'], req['n']
    K, free = split_nparams(nParams)
    res = dict(nParams=nParams, ok=np.zeros(n, dtype=bool), chiSq=np.full(n, np.inf), quality=np.zeros((n, 3), dtype=bool),
               popt=np.full((n, nParams), np.nan), dP=np.full((n, nParams), np.nan), p0=np.full((n, nParams), np.nan))
    if idx.size == 0:
        return res
    ok = status > 0
    with np.errstate(invalid='ignore'):
        q1 = ~np.any(dP > popt, axis=1)
    C0, S2_0 = req['C0'], req['S2_0']
    S2chk = S2_0 if free else 1.0 - np.sum(C0, axis=1)
    q2 = ~(S2chk + np.sum(C0, axis=1) > 1.0)
    res['p0'][idx] = req['p0']
    res['ok'][idx] = ok
    res['chiSq'][idx] = np.where(ok, chi, np.inf)
    res['quality'][idx] = np.stack([ok, np.where(ok, q1, True), np.where(ok, q2, True)], axis=1)
    res['popt'][idx] = popt
    res['dP'][idx] = dP
    return res


def fit_batch_arrays(t, y, nParams, runner, active=None):
    """conduct_curve_fitting(bReInitialise=True) for a batch through a synchronous runner."""
    req = fit_request(t, y, nParams, active)
    if req['idx'].size == 0:
        return fit_collect(req, None, None, None, None)
    popt, dP, chi, status = runner(nParams, req['p0'], req['idx'])
    return fit_collect(req, popt, dP, chi, status)


def _fit_dict(res, i):
    """Per-residue view of a batch result with components sorted by tau (sort_components, :203-209)."""
    nP = res['nParams']
    K, free = split_nparams(nP)
    f = dict(nParams=nP, ok=bool(res['ok'][i]), chiSq=float(res['chiSq'][i]), quality=list(res['quality'][i]), p0=res['p0'][i])
    if not f['ok']:
        return f
    popt, dP = res['popt'][i], res['dP'][i]
    C, tau = popt[:K], popt[K:2 * K]
    order = np.argsort(tau)
    f.update(popt=popt, dP=dP, C=C[order], tau=tau[order], dC=dP[:K][order], dtau=dP[K:2 * K][order],
             S2=(popt[-1] if free else 1.0 - np.sum(C)), dS2=(dP[-1] if free else 0.0))
    return f


def fit_batch(t, y, dy, nParams, active=None, ctx=None):
    """List-of-dicts form of fit_batch_arrays over host arrays (None for inactive residues)."""
    res = fit_batch_arrays(t, y, nParams, host_runner(t, y, dy, ctx), active)
    idx = np.arange(y.shape[0]) if active is None else np.flatnonzero(active)
    out = [None] * y.shape[0]
    for i in idx:
        out[i] = _fit_dict(res, i)
    return out


class OrderSearchBatch:
    """optimised_curve_fitting (fitting_Ct_functions.py:278-304) for all residues in lock-step, as a resumable
    state machine: `request()` hands out the next model order to solve (residues still searching + their
    initial guesses), `submit()` takes the solver's answer and applies the reference's accept / reject rules.
    Splitting the two lets a caller run the solve asynchronously (spinrelax_amd/pipeline.py overlaps the
    long tail of the last order with the next batch's C(t) kernel)."""

    def __init__(self, t, y, listDoG=(2, 3, 5, 7, 9), chiSqThreshold=0.5):
        self.t, self.y = t, y
        self.orders = tuple(listDoG)
        self.thr = chiSqThreshold
        n = y.shape[0]
        self.first = np.ones(n, dtype=bool)
        self.done = np.zeros(n, dtype=bool)
        self.best = np.full(n, -1, dtype=int)
        self.best_chi = np.full(n, np.inf)
        self.per_order = []
        self.j = 0
        self._req = None

    def request(self):
        """Next (order index, request dict) or None when the search is over."""
        if self.j >= len(self.orders):
            return None
        active = ~self.done
        if not active.any():
            return None
        self._active = active
        self._req = fit_request(self.t, self.y, self.orders[self.j], active)
        return self._req

    def submit(self, popt, dP, chi, status):
        res = fit_collect(self._req, popt, dP, chi, status)
        j, active = self.j, self._active
        self.per_order.append(res)
        allq = res['quality'].all(axis=1)
        chi_all = res['chiSq']
        was_first = self.first.copy()
        take_first = active & was_first & allq
        self.best[take_first] = j
        self.best_chi[take_first] = chi_all[take_first]
        self.first[take_first] = False
        later = active & ~was_first
        stop = later & (~allq | (chi_all >= self.best_chi * self.thr))
        self.done |= stop
        acc = later & ~stop
        self.best[acc] = j
        self.best_chi[acc] = chi_all[acc]
        self.j += 1
        return res

    def selected_arrays(self, Kmax=None):
        """S2 (n,), C (n,Kmax), tau (n,Kmax) sorted by tau, nComps (n,), chiSq (n,) of the selected models
        (vectorised; residues without a satisfactory fit have nComps = 0 and chiSq = nan)."""
        n = self.y.shape[0]
        Kmax = max(self.orders) // 2 if Kmax is None else Kmax
        S2 = np.zeros(n)
        C = np.zeros((n, Kmax))
        tau = np.ones((n, Kmax))
        K = np.zeros(n, dtype=np.int32)
        chi = np.full(n, np.nan)
        for j, res in enumerate(self.per_order):
            sel = np.flatnonzero(self.best == j)
            if sel.size == 0:
                continue
            k, free = split_nparams(res['nParams'])
            popt = res['popt'][sel]
            Cj, tj = popt[:, :k], popt[:, k:2 * k]
            order = np.argsort(tj, axis=1)
            C[sel, :k] = np.take_along_axis(Cj, order, axis=1)
            tau[sel, :k] = np.take_along_axis(tj, order, axis=1)
            S2[sel] = popt[:, -1] if free else 1.0 - np.sum(Cj, axis=1)
            K[sel] = k
            chi[sel] = res['chiSq'][sel]
        return S2, C, tau, K, chi


def order_search_batch(t, y, runner, listDoG=(2, 3, 5, 7, 9), chiSqThreshold=0.5):
    """Synchronous driver of OrderSearchBatch.  Returns (best_order_index (n,) with -1 = never satisfied, list of
    per-order batch results)."""
    search = OrderSearchBatch(t, y, listDoG, chiSqThreshold)
    while True:
        req = search.request()
        if req is None:
            break
        popt, dP, chi, status = runner(req['nParams'], req['p0'], req['idx'])
        search.submit(popt, dP, chi, status)
    return search.best, search.per_order


def optimised_curve_fitting_batch(names, t, y, dy, listDoG=(2, 3, 5, 7, 9), chiSqThreshold=0.5, fp=sys.stdout, ctx=None,
                                  return_trials=False, runner=None):
    """optimised_curve_fitting for every residue: list of selected fit dicts (None where no order was ever
    satisfactory).  Default: the one-launch search on the GPU (sr_expfit_order_search_f64); with a `runner` the
    host-driven state machine OrderSearchBatch is used instead (one solver call per model order; bit-identical
    results, tests/test_gpu_parity.py)."""
    tt = np.atleast_2d(np.asarray(t, dtype=float))
    if runner is None and len(listDoG) <= 8 and np.all(tt[:, -1] == tt[0, -1]):
        best, per_order = order_search_device_results(t, y, dy, listDoG, chiSqThreshold, ctx)
    else:
        # caller-supplied solver, more than 8 orders, or residues whose time axes end differently (one tau bound each)
        run = runner if runner is not None else host_runner(t, y, dy, ctx)
        best, per_order = order_search_batch(t, y, run, listDoG, chiSqThreshold)
    results = []
    for i in range(y.shape[0]):
        if fp is not None:
            for res in per_order:
                if np.isfinite(res['p0'][i, 0]):
                    print("    ...fit of %s with %i params yield chiSq of %g" % (names[i], res['nParams'], res['chiSq'][i]), file=fp)
        if best[i] < 0:
            if fp is not None:
                print("    ...ERROR: fit of %s has never generated a satisfactory outcome!" % names[i], file=fp)
            results.append(None)
        else:
            results.append(_fit_dict(per_order[best[i]], i))
    if return_trials:
        trials = [[_fit_dict(res, i) for res in per_order if np.isfinite(res['p0'][i, 0])] for i in range(y.shape[0])]
        return results, trials
    return results


# ---------------------------------------------------------------------------------------------------
# _fittedCt.dat reader
# ---------------------------------------------------------------------------------------------------
def read_fittedCt_parameters(fileName):
    """fitting_Ct_functions.py:432-481: header lines '# Residue: n', '# Param <name>: <v> +- <e>'; a
    section ends at the first non-comment line."""
    obj = autoCorrelations()
    index = None
    S2_slow = None
    S2_fast = None
    tmpC = OrderedDict()
    tmpTau = OrderedDict()
    inside = False
    with open(fileName) as fp:
        for line in fp:
            if line[:1] == "#":
                l = line.split()
                if l[1].startswith("Residue"):
                    if inside:
                        print("= = = ERROR in read_fittedCt_parameters: New parameter section detected when old parameter section is still being read! %s " % fileName, file=sys.stderr)
                        sys.exit(1)
                    inside = True
                    index = str(l[-1])
                elif l[1].startswith("Param"):
                    parName = l[2]
                    value = float(l[-3])
                    if parName.startswith("S2_0"):
                        S2_slow = value
                    elif parName.startswith("S2_fast"):
                        S2_fast = value
                    elif parName.startswith("C_"):
                        tmpC[str(index) + "-" + parName[2]] = value
                    elif parName.startswith("tau_"):
                        tmpTau[str(index) + "-" + parName[4]] = value
            elif inside:
                obj.add_model(index, S2=S2_slow, listC=[tmpC[k] for k in tmpC.keys()], listTau=[tmpTau[k] for k in tmpC.keys()],
                              bS2Fast=S2_fast is not None)
                inside = False
                tmpC = OrderedDict()
                tmpTau = OrderedDict()
                S2_fast = None
                S2_slow = None
                index = None
    return obj
