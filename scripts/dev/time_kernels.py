#!/usr/bin/env python3
"""Developer timing of the individual kernels with inputs resident in HBM (not the judged bench)."""
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402


def main():
    real = os.environ.get('SR_REAL_DATA')
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    nvec = int(sys.argv[2]) if len(sys.argv) > 2 else None
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    s = synth.config_shapes(cfg)
    V = nvec or s['V']
    t0 = time.time()
    pre = synth.synth_vectors_parallel(s['frames'], V, s['seed']) if real else None
    ctx = Context(0)
    print(ctx.device_info(), flush=True)
    # cheap stand-in data for timing (real data only changes DVFS a little): random unit vectors per frame
    if real:
        vecs = torch.from_numpy(pre).cuda()
    else:
        g = torch.Generator(device='cuda').manual_seed(1)
        vecs = torch.randn((s['frames'], V, 3), device='cuda', generator=g, dtype=torch.float32)
        vecs = vecs / vecs.norm(dim=-1, keepdim=True)
    N = s['N']
    Npad = (s['frames'] + 63) // 64 * 64
    soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
    L = s['L']
    Ct = torch.empty((L, V), device='cuda', dtype=torch.float64)
    dCt = torch.empty((L, V), device='cuda', dtype=torch.float64)
    hist = torch.empty((V, 72, 36), device='cuda', dtype=torch.float64)
    vsum = torch.empty((V, 3), device='cuda', dtype=torch.float64)
    outer = torch.empty((s['R'], V, 6), device='cuda', dtype=torch.float64)
    torch.cuda.synchronize()
    print('setup %.1fs  cfg%d V=%d R=%d F=%d' % (time.time() - t0, cfg, V, s['R'], s['F']), flush=True)
    edges = [np.linspace(-np.pi, np.pi, 73), np.linspace(-1, 1, 37)]
    triples = synth.exact_triples(s['R'], s['F'], V)

    def timeit(name, fn, work=None, unit=''):
        fn()
        ctx.sync()
        ts = []
        for _ in range(reps):
            ctx.timer_start()
            fn()
            ts.append(ctx.timer_stop_ms())
        ms = float(np.median(ts))
        extra = ''
        if work:
            extra = '  %.3e %s' % (work / (ms * 1e-3), unit)
        print('%-14s median %.3f ms  min %.3f%s' % (name, ms, min(ts), extra), flush=True)
        return ms

    timeit('pack', lambda: ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad),
           2 * 12 * s['frames'] * V, 'B/s')
    ms = timeit('ct_palmer', lambda: ctx.ct_palmer_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, Ct.data_ptr(), dCt.data_ptr()),
                triples, 'triples/s')
    print('   -> %.1f TFLOP/s (8 flop/triple) = %.1f %% of 157.3' % (8 * triples / ms / 1e9, 8 * triples / ms / 1e9 / 157.3 * 100))
    timeit('rotate_hist', lambda: ctx.rotate_hist_dev(soa.data_ptr(), Npad, N, V, synth.Q_EXT, edges[0], edges[1],
                                                     hist.data_ptr(), vsum.data_ptr(), outer.data_ptr(), s['F']),
           12 * N * V, 'B/s')
    print('hist sum', float(hist.sum()), 'expected', N * V)


if __name__ == '__main__':
    main()
