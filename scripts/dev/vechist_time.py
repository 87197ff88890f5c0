#!/usr/bin/env python3
"""Developer timing of sr_rotate_hist_f32_dev alone on the cfg3 planes (library from SPINRELAX_HIP_LIB or the default)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402

s = synth.config_shapes(3)
V = 512
pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
vecs = torch.from_numpy(pre).cuda()
N = s['N']
Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
hist = torch.empty((V, 72, 36), device='cuda', dtype=torch.float64)
vsum = torch.empty((V, 3), device='cuda', dtype=torch.float64)
outer = torch.empty((s['R'], V, 6), device='cuda', dtype=torch.float64)
edges = [np.linspace(-np.pi, np.pi, 73), np.linspace(-1, 1, 37)]
ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)


def fn():
    ctx.rotate_hist_dev(soa.data_ptr(), Npad, N, V, synth.Q_EXT, edges[0], edges[1], hist.data_ptr(), vsum.data_ptr(), outer.data_ptr(), s['F'])


fn()
ctx.sync()
ts = []
for _ in range(int(os.environ.get('REPS', '7'))):
    ctx.timer_start()
    fn()
    ts.append(ctx.timer_stop_ms())
print('%s  median %.4f ms  min %.4f  hist sum %d' % (os.environ.get('SPINRELAX_HIP_LIB', 'default'), float(np.median(ts)), min(ts), int(hist.sum().item())))
ctx.close()
