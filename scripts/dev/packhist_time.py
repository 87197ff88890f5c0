#!/usr/bin/env python3
"""Developer timing: k_pack_soa, k_vechist and the fused k_pack_hist on the cfg3 input; NHIST=<frames> limits the fused
kernel's histogram to the first frames (the rest of the input is then only packed: the memory pattern alone)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth, ct as hostct                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402

s = synth.config_shapes(3)
V = int(os.environ.get('NVEC', '512'))
pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
vecs = torch.from_numpy(pre).cuda()
Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
hist = torch.empty((V, 2592), device='cuda', dtype=torch.float64)
vsum = torch.empty((V, 3), device='cuda', dtype=torch.float64)
outer = torch.empty((s['R'], V, 6), device='cuda', dtype=torch.float64)
e = hostct.lambert_edges()
q = np.array(synth.Q_EXT)
nh = int(os.environ.get('NHIST', str(s['N'])))


def t(fn, reps=5):
    fn(); ctx.sync()
    ts = []
    for _ in range(reps):
        ctx.timer_start(); fn(); ts.append(ctx.timer_stop_ms())
    return float(np.median(ts))


print('pack      %.4f ms' % t(lambda: ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)))
print('hist      %.4f ms' % t(lambda: ctx.rotate_hist_dev(soa.data_ptr(), Npad, s['N'], V, q, e[0], e[1], hist.data_ptr(), vsum.data_ptr(), outer.data_ptr(), s['F'])))
print('fused     %.4f ms (N_hist %d)' % (t(lambda: ctx.pack_hist_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad, nh, q, e[0], e[1],
                                                                     hist.data_ptr(), vsum.data_ptr(), outer.data_ptr(), s['F'] if nh >= s['F'] else 0)), nh))
print('fused     %.4f ms (N_hist 4: pack only)' % t(lambda: ctx.pack_hist_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad, 4, q, e[0], e[1],
                                                                            hist.data_ptr(), vsum.data_ptr(), outer.data_ptr(), 0)))
ctx.close()
