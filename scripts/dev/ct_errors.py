#!/usr/bin/env python3
"""Developer diagnostic: error of the GPU C(t) vs the golden float64 reference values."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
ctx = Context(0)
for tag, cfg, nvec in (('cfg1', 1, None), ('cfg2', 2, None), ('cfg3s', 3, 8)):
    g = np.load(os.path.join(ROOT, 'tests/golden/%s_ct.npz' % tag))
    s = synth.config_shapes(cfg)
    v = synth.synth_config(cfg, nvec=nvec)
    for mode in (0, 1):
        Ct, dCt = ctx.ct_palmer(v, s['R'], s['F'], mode=mode)
        eC = np.abs(Ct - g['Ct64']); eD = np.abs(dCt - g['dCt64'])
        print('%s mode %d: Ct abs %.2e rel %.2e | dCt abs %.2e rel %.2e (min dCt %.2e) | ref f32: Ct rel %.2e dCt rel %.2e' % (
            tag, mode, eC.max(), (eC / np.abs(g['Ct64'])).max(), eD.max(), (eD / g['dCt64']).max(), g['dCt64'].min(),
            (np.abs(g['Ct32'] - g['Ct64']) / np.abs(g['Ct64'])).max(), (np.abs(g['dCt32'] - g['dCt64']) / g['dCt64']).max()), flush=True)
