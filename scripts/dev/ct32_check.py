#!/usr/bin/env python3
"""Developer check of k_ct_rfft32 (ct_fft = 3) against the float64 transform kernel (ct_fft = 2) on one GPU: values over the
shapes that matter (aligned / masked / odd / M = 8192 / non-unit vectors / unaligned chunk starts), then the timing on the
full cfg3 planes.  usage: ct32_check.py [notime]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402


def both(ctx, vecs, R, F, chunk_start=None):
    out = {}
    for k in (2, 3):
        ctx.set_option('ct_fft', k)
        out[k] = ctx.ct_palmer(vecs, R, F, chunk_start=chunk_start)
    ctx.set_option('ct_fft', 2)
    return out


def report(tag, o, R, F):
    (C2, D2), (C3, D3) = o[2], o[3]
    eC = np.max(np.abs(C3 / C2 - 1.0))
    eD = np.max(np.abs(D3 - D2)) if R > 1 else 0.0
    scale = np.maximum(1.0, np.max(np.abs(C2), axis=0))[None, :]            # tests/conftest.py:dct_close_f32_transform
    bar = 5e-8 / (np.sqrt(R) - 1.0) if R > 1 else 0.0
    ok = eC < 1e-7 and (R == 1 or np.all(np.abs(D3 - D2) <= np.maximum(1e-6 * np.abs(D2), scale * bar)))
    print('%-34s C(t) rel %.2e   dC(t) abs %.2e (bar %.2e)  %s' % (tag, eC, eD, bar, 'ok' if ok else 'FAIL'), flush=True)
    return ok


def main():
    ctx = Context(0)
    good = True
    s = synth.config_shapes(3)
    v3 = synth.synth_config(3, nvec=8)
    good &= report('cfg3 slice F=4096 R=24', both(ctx, v3, s['R'], s['F']), s['R'], s['F'])
    for F, R in ((4000, 3), (3001, 3), (2731, 4), (4096, 2), (4094, 2), (5000, 3), (5461, 2), (4097, 3), (4098, 2), (5333, 2)):
        vecs = synth.synth_vectors(R * F + 5, 3, seed=900 + F)
        good &= report('F=%d R=%d' % (F, R), both(ctx, vecs, R, F), R, F)
    # unaligned chunk starts (odd offsets): the dword load path
    F, R = 4096, 3
    vecs = synth.synth_vectors(R * F + 40, 2, seed=77)
    good &= report('F=4096 odd chunk starts', both(ctx, vecs, R, F, chunk_start=np.array([1, F + 3, 2 * F + 7])), R, F)
    # not unit vectors: scaled, and a zero vector among them (sixth signal)
    vecs = synth.synth_vectors(R * F, 3, seed=78).copy()
    vecs[:, 0] *= 1.25
    vecs[100, 1] = 0.0
    vecs[:, 2] *= np.float32(1.0 + 2e-6)
    good &= report('F=4096 non-unit vectors', both(ctx, vecs, R, F), R, F)
    F = 5000
    vecs = synth.synth_vectors(R * F, 2, seed=79).copy()
    vecs[:, 0] *= 0.5
    good &= report('F=5000 non-unit vectors', both(ctx, vecs, R, F), R, F)
    print('ALL OK' if good else 'SOME FAILED', flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'notime':
        ctx.close()
        return 0 if good else 1
    import torch
    V = 512
    pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
    vecs = torch.from_numpy(pre).cuda()
    Npad = (s['frames'] + 63) // 64 * 64
    soa = torch.empty((V, 3, Npad), device='cuda', dtype=torch.float32)
    psum = torch.empty((V * s['R'] * ctx.psum_stride(s['F']),), device='cuda', dtype=torch.float64)
    ctx.pack_soa_dev(vecs.data_ptr(), s['frames'], V, 0, V, soa.data_ptr(), Npad)
    for k in (2, 3, 2, 3):
        ctx.set_option('ct_fft', k)
        ctx.ct_sums_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, psum.data_ptr())
        ctx.sync()
        ts = []
        for _ in range(7):
            ctx.timer_start()
            ctx.ct_sums_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, psum.data_ptr())
            ts.append(ctx.timer_stop_ms())
        print('cfg3 512 vectors  ct_fft=%d  median %.4f ms  min %.4f' % (k, float(np.median(ts)), min(ts)), flush=True)
    ctx.close()
    return 0 if good else 1


if __name__ == '__main__':
    sys.exit(main())
