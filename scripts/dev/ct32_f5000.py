#!/usr/bin/env python3
"""Developer timing of the M = 8192 variant of the C(t) kernel (F = 5000 and 5461 chunks of the cfg3 planes)."""
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
ctx.set_option('ct_fft', int(os.environ.get('CT_FFT', '3')))
vecs = torch.from_numpy(pre).cuda()
Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)
for F in (4096, 5000, 5461):
    R = s['frames'] // F
    psum = torch.empty((V * R * ctx.psum_stride(F),), device='cuda', dtype=torch.float64)
    ctx.ct_sums_dev(soa.data_ptr(), Npad, R, F, V, psum.data_ptr())
    ctx.sync()
    ts = []
    for _ in range(12):
        ctx.timer_start()
        ctx.ct_sums_dev(soa.data_ptr(), Npad, R, F, V, psum.data_ptr())
        ts.append(ctx.timer_stop_ms())
    print('F=%d R=%d  median %.4f ms  min %.4f   per frame %.3f ns' % (F, R, float(np.median(ts)), min(ts), min(ts) * 1e6 / (R * F * V)))
ctx.close()
