"""Developer probe for rocprofv3 counters: merged model-order search restricted to the first `cut` orders (default 1: two
parameters only), G batches, longest first.  usage: fit_order2_only.py [cut] [G]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd.pipeline import DevicePipeline
from spinrelax_amd import fitting_Ct_functions as fitCt
cut = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
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
f64 = dict(device=dev, dtype=torch.float64); i32 = dict(device=dev, dtype=torch.int32)
n = G * V
y, dy = s0.CtT.repeat_interleave(G, dim=0), s0.dCtT.repeat_interleave(G, dim=0)
lst = p1.listDoG[:cut]
nO, Pmax = len(lst), max(lst)
tg = torch.from_numpy(fitCt.tau_guesses(p1.t_host[0], lst)).to(dev)
o = dict(popt=torch.empty((nO, n, Pmax), **f64), dP=torch.empty((nO, n, Pmax), **f64), chisq=torch.empty((nO, n), **f64),
         status=torch.empty((nO, n), **i32), nfev=torch.empty((nO, n), **i32), best=torch.empty((n,), **i32), S2=torch.empty((n,), **f64),
         C=torch.empty((n, max(1, Pmax // 2)), **f64), tau=torch.empty((n, max(1, Pmax // 2)), **f64), chi=torch.empty((n,), **f64), K=torch.empty((n,), **i32))
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.order_search_batched_dev(p1.t_dev.data_ptr(), 1, y.data_ptr(), dy.data_ptr(), n, s['L'], lst, tg.data_ptr(), 1, p1.tau_max, p1.chi_thr,
                                 o['popt'].data_ptr(), o['dP'].data_ptr(), o['chisq'].data_ptr(), o['status'].data_ptr(), o['nfev'].data_ptr(),
                                 o['best'].data_ptr(), o['S2'].data_ptr(), o['C'].data_ptr(), o['tau'].data_ptr(), o['chi'].data_ptr(), o['K'].data_ptr())
    ctx.sync(); torch.cuda.synchronize()
    print('orders %s G=%d: %.3f ms per batch, %d evaluations per batch' % (lst, G, (time.perf_counter() - t0) * 1e3 / G, int(o['nfev'].sum().item()) // G), flush=True)
p1.close(); ctx.close()
