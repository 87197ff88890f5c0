#!/usr/bin/env python3
"""Developer diagnostic: does an idle gap before the C(t) kernel change its duration (clock ramp)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
s = synth.config_shapes(3); V = 512
ctx = Context(0)
g = torch.Generator(device='cuda').manual_seed(1)
vecs = torch.randn((s['frames'], V, 3), device='cuda', generator=g, dtype=torch.float32); vecs = vecs / vecs.norm(dim=-1, keepdim=True)
Npad = (s['frames'] + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
Ct = torch.empty((s['L'], V), device='cuda', dtype=torch.float64); dCt = torch.empty_like(Ct)
ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)
run = lambda: ctx.ct_palmer_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, Ct.data_ptr(), dCt.data_ptr())
run(); ctx.sync()
for gap in (0.0, 0.005, 0.03, 0.2):
    ts = []
    for _ in range(6):
        time.sleep(gap)
        ctx.timer_start(); run(); ts.append(ctx.timer_stop_ms())
    print('idle gap %.3f s: ct ms' % gap, ' '.join('%.2f' % t for t in ts), flush=True)
# back-to-back pairs after a gap: is only the first one slow?
time.sleep(0.05)
ts = []
for _ in range(4):
    ctx.timer_start(); run(); ts.append(ctx.timer_stop_ms())
print('after 50 ms idle, 4 back to back:', ' '.join('%.2f' % t for t in ts))
