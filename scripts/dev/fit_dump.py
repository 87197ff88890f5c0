#!/usr/bin/env python3
"""Developer diagnostic: run the device fit on every golden trial and dump the results."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spinrelax_amd.hip import Context
ctx = Context(0)
out = {}
for tag in ('cfg1', 'cfg2', 'cfg3s'):
    g = np.load(os.path.join(ROOT, 'tests/golden/%s_fit.npz' % tag))
    t, y, dy = g['t'], g['y'], g['dy']
    for j, nP in enumerate(g['listDoG']):
        for an in (0, 1):
            popt, pcov, chi, status, nfev = ctx.expfit(t, y, dy, g['trial_p0'][:, j, :nP], t[0, -1] * 10, analytic_jac=bool(an))
            k = '%s_%d_%d' % (tag, nP, an)
            out[k + '_popt'] = popt; out[k + '_chi'] = chi; out[k + '_status'] = status; out[k + '_nfev'] = nfev
            out[k + '_dP'] = np.sqrt(np.abs(np.diagonal(pcov, axis1=1, axis2=2)))
name = sys.argv[1] if len(sys.argv) > 1 else 'fit_dump'
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', name + '.npz'), **out)
print('ok')
