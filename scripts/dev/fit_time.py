#!/usr/bin/env python3
"""Developer diagnostic: per-iteration latency of the device TRF fit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spinrelax_amd.hip import Context
ctx = Context(0)
g = np.load(os.path.join(ROOT, 'tests/golden/cfg3s_fit.npz'))
t, y, dy = g['t'], g['y'], g['dy']
for j, nP in enumerate(g['listDoG']):
    for an in (False, True):
        p0 = g['trial_p0'][:, j, :nP]
        ctx.expfit(t, y, dy, p0, t[0, -1] * 10, analytic_jac=an)
        t0 = time.perf_counter()
        for _ in range(3):
            popt, pcov, chi, status, nfev = ctx.expfit(t, y, dy, p0, t[0, -1] * 10, analytic_jac=an)
        dt = (time.perf_counter() - t0) / 3
        print('nP %d analytic %d: wall %.3f ms, max nfev %d -> %.1f us per nfev; nfev %s' % (nP, an, dt * 1e3, nfev.max(), dt * 1e6 / nfev.max(), list(nfev)), flush=True)
