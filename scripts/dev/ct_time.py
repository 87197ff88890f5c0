#!/usr/bin/env python3
"""Developer timing of the C(t) sums kernel alone on the cfg3 planes (CFG=2: cfg2's, NVEC of them); CT_FFT=0..4 selects the
formulation."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402

s = synth.config_shapes(int(os.environ.get('CFG', '3')))
V = int(os.environ.get('NVEC', '512'))
pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
ctx.set_option('ct_fft', int(os.environ.get('CT_FFT', '2')))
vecs = torch.from_numpy(pre).cuda()
Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
psum = torch.empty((V * s['R'] * ctx.psum_stride(s['F']),), device='cuda', dtype=torch.float64)
ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)


def fn():
    ctx.ct_sums_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, psum.data_ptr())


fn()
ctx.sync()
ts = []
for _ in range(int(os.environ.get('REPS', '5'))):
    ctx.timer_start()
    fn()
    ts.append(ctx.timer_stop_ms())
print('ct_fft=%s  median %.4f ms  min %.4f  checksum %.12g' % (os.environ.get('CT_FFT', '2'), float(np.median(ts)), min(ts), float(psum[:100000].nan_to_num().sum().item())))
ctx.close()
