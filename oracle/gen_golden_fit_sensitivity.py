#!/usr/bin/env python3
"""
gen_golden_fit_sensitivity.py -- how far does the REFERENCE's own fit move when its input changes in the last bit?

TEST INFRASTRUCTURE (build container only):   make -C oracle ref && python oracle/gen_golden_fit_sensitivity.py

For every (residue, model order) trial of tests/golden/<tag>_fit.npz the real reference
(fitting_Ct_functions.autoCorrelationModel.conduct_curve_fitting -> scipy curve_fit / TRF, imported from /root/reference by
oracle/ref_loader.py) is run again on C(t) values moved by ONE ulp (all up, all down, alternating, then seeded random
+-1 ulp patterns: 64 runs per trial at L = 50, 32 at L = 512, 16 at L = 2048) -- the size of change a different exp() or
summation order makes inside any faithful re-execution of the algorithm.  Stored per trial: every chi^2 the reference
reached (`trial_chi_perturbed`), their range (`trial_chi_lo`, `trial_chi_hi`), the largest relative change
(`trial_chi_spread`) and whether the success flag flipped (`trial_ok_flip`).  tests/test_gpu_parity.py uses them as the
PER-TRIAL reason wherever the device fit is not within the tier tolerance of the unperturbed reference run: a trial whose
reference result moves by 1e-3 -- or jumps to another minimum 24 % higher in 2 of 64 runs, as the 7-parameter fit of
residue 26 of cfg1 does -- under a one-ulp change of its input has no 1e-6 answer to be compared with; the device must
then land inside the reference's own range.
Fixtures are data; no reference source text is stored.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader                                   # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
ref = ref_loader.load()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def perturbed(y, pattern, seed):
    up, dn = np.nextafter(y, np.inf), np.nextafter(y, -np.inf)
    if pattern == 0:
        return up
    if pattern == 1:
        return dn
    if pattern == 2:
        return np.where(np.arange(y.size) % 2 == 0, up, dn)
    rng = np.random.RandomState(seed + 7919 * pattern)
    return np.where(rng.rand(y.size) < 0.5, up, dn)


def main():
    for tag, NP in (('cfg1', 64), ('cfg2', 32), ('cfg3s', 16)):
        g = np.load(os.path.join(GOLD, '%s_fit.npz' % tag), allow_pickle=True)
        t, y, dy, orders = g['t'], g['y'], g['dy'], [int(n) for n in g['listDoG']]
        nres, nord = y.shape[0], len(orders)
        chis = np.full((nres, nord, NP + 1), np.nan)
        oks = np.zeros((nres, nord, NP + 1), dtype=bool)
        for i in range(nres):
            for j, nP in enumerate(orders):
                for k in range(NP + 1):
                    yy = y[i] if k == 0 else perturbed(y[i], k - 1, 1000 * i + j)
                    m = ref.fitCt.autoCorrelationModel(name=str(i))
                    m.set_nParams(nP)
                    chi, qual = quiet(m.conduct_curve_fitting, t[i], yy, dy[i], bReInitialise=True)
                    chis[i, j, k] = chi
                    oks[i, j, k] = bool(qual[0])
                # the unperturbed run must be the fixture's
                assert (np.isnan(chis[i, j, 0]) and np.isnan(g['trial_chi'][i, j])) or chis[i, j, 0] == g['trial_chi'][i, j] or \
                    not np.isfinite(chis[i, j, 0]), (tag, i, j, chis[i, j, 0], g['trial_chi'][i, j])
        with np.errstate(invalid='ignore', divide='ignore'):
            spread = np.nanmax(np.abs(chis[:, :, 1:] / chis[:, :, :1] - 1.0), axis=2)
        spread[~np.isfinite(spread)] = np.inf
        flip = (oks[:, :, 1:] != oks[:, :, :1]).any(axis=2)
        fin = np.where(np.isfinite(chis), chis, np.nan)
        with np.errstate(all='ignore'):
            lo, hi = np.nanmin(fin, axis=2), np.nanmax(fin, axis=2)
        np.savez_compressed(os.path.join(GOLD, '%s_fit_sens.npz' % tag), trial_chi_perturbed=chis, trial_ok_perturbed=oks,
                            trial_chi_spread=spread, trial_chi_lo=lo, trial_chi_hi=hi, trial_ok_flip=flip, listDoG=np.array(orders),
                            n_perturbations=NP)
        for j, nP in enumerate(orders):
            ok = oks[:, j, 0]
            sp = spread[ok, j]
            print('%s order %d: %d successful reference fits; own chi^2 spread under one-ulp input changes: median %.1e, max %.1e, '
                  '> 1e-6: %d, > 1e-4: %d; success flag flips: %d' % (tag, nP, ok.sum(), np.median(sp) if sp.size else 0, sp.max() if sp.size else 0,
                                                                        (sp > 1e-6).sum(), (sp > 1e-4).sum(), flip[:, j].sum()))


if __name__ == '__main__':
    main()
