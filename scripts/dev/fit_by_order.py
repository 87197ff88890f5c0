"""Developer probe: what each model order costs when the chip is full of fits: merged launches (G batches, longest first)
with the order list cut after 2, 3, 5, 7, 9 parameters."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd.pipeline import DevicePipeline
from spinrelax_amd import fitting_Ct_functions as fitCt
G = 20
s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
p1 = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                    field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1, stream=torch.cuda.Stream(device=dev))
p1.step(vecs)
s0 = p1.slots[0]
r = s0.result
nf = r['nfev'].astype(float); nf[r['status'] == -100] = 0
print('evaluations per order', nf.sum(axis=1), 'fits per order', (r['status'] != -100).sum(axis=1))
cost = nf.sum(axis=0)
order = torch.from_numpy(np.argsort(-cost, kind='stable').copy()).to(dev)
f64 = dict(device=dev, dtype=torch.float64); i32 = dict(device=dev, dtype=torch.int32)
n = G * V
y, dy = s0.CtT[order].repeat_interleave(G, dim=0), s0.dCtT[order].repeat_interleave(G, dim=0)
t = hostt = None
prev = 0.0
full = p1.listDoG
for a in sys.argv[1:]:
    k, v = a.split('=')
    ctx.set_option(k, int(v))
for cut in range(1, (4 if len(sys.argv) > 1 else len(full) + 1)):
    lst = full[:cut]
    nO, Pmax = len(lst), max(lst)
    tg = torch.from_numpy(fitCt.tau_guesses(p1.t_host[0], lst)).to(dev)
    o = dict(popt=torch.empty((nO, n, Pmax), **f64), dP=torch.empty((nO, n, Pmax), **f64), chisq=torch.empty((nO, n), **f64),
             status=torch.empty((nO, n), **i32), nfev=torch.empty((nO, n), **i32), best=torch.empty((n,), **i32), S2=torch.empty((n,), **f64),
             C=torch.empty((n, max(1, Pmax // 2)), **f64), tau=torch.empty((n, max(1, Pmax // 2)), **f64), chi=torch.empty((n,), **f64), K=torch.empty((n,), **i32))
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.order_search_batched_dev(p1.t_dev.data_ptr(), 1, y.data_ptr(), dy.data_ptr(), n, s['L'], lst, tg.data_ptr(), 1, p1.tau_max, p1.chi_thr,
                                     o['popt'].data_ptr(), o['dP'].data_ptr(), o['chisq'].data_ptr(), o['status'].data_ptr(), o['nfev'].data_ptr(),
                                     o['best'].data_ptr(), o['S2'].data_ptr(), o['C'].data_ptr(), o['tau'].data_ptr(), o['chi'].data_ptr(), o['K'].data_ptr())
        ctx.sync(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ev = o['nfev'].cpu().numpy().astype(float); stt = o['status'].cpu().numpy(); ev[stt == -100] = 0
    tt = min(ts) / G
    print('orders %-16s %.3f ms per batch (+%.3f)  evaluations per batch by order %s' % (lst, tt, tt - prev, (ev.sum(axis=1) / G).astype(int)), flush=True)
    prev = tt
p1.close(); ctx.close()
