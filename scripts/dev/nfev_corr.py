"""Developer probe: do the evaluation counts of the lower orders predict which residues are expensive at order 9?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd.pipeline import DevicePipeline
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
nf, stt, chi = r['nfev'], r['status'], r['chisq']
print('orders', p1.listDoG)
tried9 = stt[4] != -100
idx = np.nonzero(tried9)[0]
o = idx[np.argsort(-nf[4][idx])]
print('residues that reach order 9: %d; their nfev at order 9 (desc):' % idx.size, nf[4][o][:20])
print('nfev at order 7 of those           :', nf[3][o][:20])
print('nfev at order 5 of those           :', nf[2][o][:20])
print('chisq at order 7 of those          :', np.round(chi[3][o][:20], 4))
for name, x in (('nfev7', nf[3][idx]), ('nfev5', nf[2][idx]), ('chi7', chi[3][idx]), ('nfev2+3+5+7', nf[:4, idx].sum(axis=0))):
    rk = np.argsort(-x)
    pos = {res: i for i, res in enumerate(idx[rk])}
    print('%-12s rank of the 5 most expensive order-9 residues when sorted by it:' % name, [pos[res] for res in o[:5]], 'of', idx.size)
tot = nf.sum(axis=0)
print('total nfev per residue: top', np.sort(tot)[::-1][:10], 'median', np.median(tot))
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out', 'nfev_cfg3.npz'), nfev=nf, status=stt)
# per-evaluation time by order: single-order solves of the whole batch, alone
import time
for j, nP in enumerate(p1.listDoG):
    pass
p1.close(); ctx.close()
