#!/usr/bin/env python3
"""Developer probe: how does a CU mask map onto the 8 XCDs?  Times the C(t) kernel under different masks."""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402


def words(bits):
    w = [0] * 8
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return w


def main():
    s = synth.config_shapes(3)
    V = 512
    ctx = Context(0)
    g = torch.Generator(device='cuda').manual_seed(1)
    Npad = (s['frames'] + 63) // 64 * 64
    soa = torch.randn((V, 3, Npad), device='cuda', generator=g, dtype=torch.float32)
    L = s['L']
    Ct = torch.empty((L, V), device='cuda', dtype=torch.float64)
    dCt = torch.empty((L, V), device='cuda', dtype=torch.float64)
    torch.cuda.synchronize()
    allb = set(range(256))
    masks = {
        'full': allb,
        'drop top16 (240..255)': allb - set(range(240, 256)),
        'drop 2 per 32-block': allb - {32 * k + j for k in range(8) for j in (30, 31)},
        'drop i%16==15': allb - {i for i in range(256) if i % 16 == 15},
        'only word0 (0..31)': set(range(32)),
        'only i%8==0': {i for i in range(256) if i % 8 == 0},
        'drop top32': allb - set(range(224, 256)),
    }
    for name, bits in masks.items():
        st = ctx.stream_create(words(sorted(bits)))
        ctx.set_stream(st)
        ts = []
        for _ in range(4):
            ctx.timer_start()
            ctx.ct_palmer_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, Ct.data_ptr(), dCt.data_ptr())
            ts.append(ctx.timer_stop_ms())
        ctx.set_stream(0)
        ctx.stream_destroy(st)
        print('%-26s CUs %3d  ct ms %s  -> x%.3f vs ideal x%.3f' % (name, len(bits), ' '.join('%.2f' % t for t in ts),
                                                                  min(ts) / 6.55, 256 / len(bits)), flush=True)


if __name__ == '__main__':
    main()
