#!/usr/bin/env python3
"""Developer probe (library built with -DSR_FIT_DEV_NJEV: nfev carries the number of Jacobians in its high half): trial steps per
Jacobian of the model-order search on the cfg3 batch -- how often a trust-region step is REJECTED and the sub-problem solved again
with the same matrix."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402
from spinrelax_amd.pipeline import DevicePipeline    # noqa: E402

s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
p1 = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                    field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1, stream=torch.cuda.Stream(device=dev))
p1.step(vecs)
r = p1.slots[0].result
raw, stt = r['nfev'], r['status']
nf, nj = raw & 0xFFFF, raw >> 16
for j, nP in enumerate(p1.listDoG):
    m = stt[j] != -100
    if not m.any():
        continue
    f, jv = nf[j][m], nj[j][m]
    rej = f - jv                          # trial steps that did not end in a new Jacobian (rejected, or the terminating one)
    print('order %d: %4d fits, evaluations %6d, Jacobians %6d, trials per Jacobian %.2f | fits with > 1.5 trials per Jacobian: %d (their evaluations: %d)'
          % (nP, m.sum(), f.sum(), jv.sum(), f.sum() / max(1, jv.sum()), int((f > 1.5 * jv).sum()), int(f[f > 1.5 * jv].sum())))
    o = np.argsort(-f)[:6]
    print('        the longest: evaluations', f[o], 'Jacobians', jv[o])
tot_f, tot_j = nf[stt != -100].sum(), nj[stt != -100].sum()
print('all: evaluations %d, Jacobians %d, trials per Jacobian %.3f' % (tot_f, tot_j, tot_f / tot_j))
p1.close()
ctx.close()
