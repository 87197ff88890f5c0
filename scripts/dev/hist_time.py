#!/usr/bin/env python3
"""Developer timing of the rotate + histogram kernel (k_vechist + finalize) alone on the cfg3 planes; prints a checksum of the
counts so that variant builds (SPINRELAX_HIP_LIB) can be told apart from wrong ones.  BLOCK=<frames> overrides the S2 block."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402

s = synth.config_shapes(3)
V = int(os.environ.get('NVEC', '512'))
pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
vecs = torch.from_numpy(pre).cuda()
N = s['frames']
Npad = (N + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
ctx.pack_soa_dev(vecs.data_ptr(), N, V, 0, V, soa.data_ptr(), Npad)
nphi, ncos = 72, 36
ephi = np.linspace(-np.pi, np.pi, nphi + 1)
ecos = np.linspace(-1.0, 1.0, ncos + 1)
hist = torch.zeros((V, nphi * ncos), device='cuda', dtype=torch.float64)
nB = N // s['F']
vecsum = torch.zeros((V, 3), device='cuda', dtype=torch.float64)
outer = torch.zeros((V, max(nB, 1), 6), device='cuda', dtype=torch.float64)
blk = int(os.environ.get('BLOCK', str(s['F'])))


def fn():
    ctx.rotate_hist_dev(soa.data_ptr(), Npad, N, V, synth.Q_EXT, ephi, ecos, hist.data_ptr(), vecsum.data_ptr(), outer.data_ptr(), blk)


fn()
ctx.sync()
ts = []
for _ in range(int(os.environ.get('REPS', '9'))):
    ctx.timer_start()
    fn()
    ts.append(ctx.timer_stop_ms())
h = hist.cpu().numpy()
print('hist  median %.4f ms  min %.4f  counts %d  checksum %.0f' % (float(np.median(ts)), min(ts), int(h.sum()), float((h * np.arange(h.shape[1])[None, :]).sum())))
ctx.close()
