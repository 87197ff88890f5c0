#!/usr/bin/env python3
"""Developer tool: phase times of k_ct_rfft32's transform loop from a library built with -DSR_CT32_STAMPS (s_memtime stamps summed
per wave, left behind lag L of every series' raw sums).  usage: SPINRELAX_HIP_LIB=_variants/lib_stamps.so ct32_stamps.py"""
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
ctx.set_option('ct_fft', 3)
vecs = torch.from_numpy(pre).cuda()
Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
Lp = ctx.psum_stride(s['F'])
psum = torch.zeros((V * s['R'] * Lp,), device='cuda', dtype=torch.float64)
ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)
for _ in range(3):
    ctx.ct_sums_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, psum.data_ptr())
ctx.sync()
p = psum.cpu().numpy().reshape(V * s['R'], Lp)
L = s['L']
d = p[:, L + 2:L + 2 + 64].reshape(-1, 4, 16)          # series, wave, value
names = ['step 1 (12-pt + twiddles + write)', 'barrier 1', 'step-2 reads arrive', 'step 2 (16-pt + twiddles + write)', 'step-3 reads arrive',
         'step 3 (16-pt)', 'row rewrite', 'barrier 2', 'spectrum (8 partner reads + arithmetic)', 'loads + next signal', 'barrier 3', 'loop total', 'to the end']
for w in range(4):
    print('wave %d (median over %d series, shader cycles summed over the %d passes of the loop + the back transform):' % (w, d.shape[0], 5))
    for i, n in enumerate(names):
        print('   %-44s %9.0f' % (n, np.median(d[:, w, i])))
ctx.close()
