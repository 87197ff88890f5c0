#!/usr/bin/env python3
"""Developer probe: FFT formulation of kernel 1 against the direct kernel (float64 mode) and timing."""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402


def main():
    ctx = Context(0)
    for cfg, V in ((2, 32), (3, 16)):
        s = synth.config_shapes(cfg)
        vecs = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
        ref = ctx.ct_palmer(vecs, s['R'], s['F'], mode=1)
        ctx.set_option('ct_fft', 0)
        d = ctx.ct_palmer(vecs, s['R'], s['F'])
        ctx.set_option('ct_fft', 1)
        f = ctx.ct_palmer(vecs, s['R'], s['F'])
        for name, x in (('direct f32', d), ('fft', f)):
            print('cfg%d F=%d %-10s: max rel err C(t) %.2e, dC(t) %.2e' % (cfg, s['F'], name, np.max(np.abs(x[0] / ref[0] - 1)),
                                                                      np.max(np.abs(x[1] - ref[1]) / np.abs(ref[1]))), flush=True)
    # timing at full cfg3 size with resident planes
    s = synth.config_shapes(3)
    V = 512
    g = torch.Generator(device='cuda').manual_seed(1)
    Npad = (s['frames'] + 63) // 64 * 64
    soa = torch.randn((V, 3, Npad), device='cuda', generator=g, dtype=torch.float32)
    Ct = torch.empty((s['L'], V), device='cuda', dtype=torch.float64)
    dCt = torch.empty_like(Ct)
    for fft in (0, 1):
        ctx.set_option('ct_fft', fft)
        ts = []
        for _ in range(6):
            ctx.timer_start()
            ctx.ct_palmer_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, Ct.data_ptr(), dCt.data_ptr())
            ts.append(ctx.timer_stop_ms())
        print('cfg3 512 vectors, ct_fft=%d: %s ms' % (fft, ' '.join('%.2f' % t for t in ts)), flush=True)


if __name__ == '__main__':
    main()
