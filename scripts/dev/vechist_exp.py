#!/usr/bin/env python3
"""Developer experiment: time sr_rotate_hist_f32_dev alone for alternative builds of the library (SPINRELAX_HIP_LIB)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
s = synth.config_shapes(3); V = 512
pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
vecs = torch.from_numpy(pre).cuda()
N = s['N']; Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
hist = torch.empty((V, 72, 36), device='cuda', dtype=torch.float64)
vsum = torch.empty((V, 3), device='cuda', dtype=torch.float64)
outer = torch.empty((s['R'], V, 6), device='cuda', dtype=torch.float64)
edges = [np.linspace(-np.pi, np.pi, 73), np.linspace(-1, 1, 37)]
ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)
fn = lambda: ctx.rotate_hist_dev(soa.data_ptr(), Npad, N, V, synth.Q_EXT, edges[0], edges[1], hist.data_ptr(), vsum.data_ptr(), outer.data_ptr(), s['F'])
fn(); ctx.sync()
ts = []
for _ in range(7):
    ctx.timer_start(); fn(); ts.append(ctx.timer_stop_ms())
print('%%s  median %%.4f ms  min %%.4f  hist sum %%d' %% (os.environ.get('SPINRELAX_HIP_LIB', 'default'), float(np.median(ts)), min(ts), int(hist.sum().item())))
ctx.close()
''' % ROOT

for lib in sys.argv[1:]:
    env = dict(os.environ, SPINRELAX_HIP_LIB=os.path.abspath(lib))
    p = subprocess.run([sys.executable, '-c', CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    print('\n'.join(l for l in p.stdout.decode().splitlines() if 'median' in l or 'Error' in l), flush=True)
