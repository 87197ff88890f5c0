#!/usr/bin/env python3
"""Developer probe: what a bandwidth-bound kernel running BESIDE the C(t) kernel costs it.  Stream A runs K C(t) launches back to
back; stream B meanwhile runs (a) nothing, (b) the pack kernel over and over, (c) a plain device-to-device copy of the same bytes
(torch's copy kernel), (d) the histogram kernel.  Reported: the duration of A's K launches, how many B operations finished inside
it, and the cost per B operation = (T - T_alone) / count -- to be compared with the operation's own duration alone."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
K = int(os.environ.get('K', '20'))
s = synth.config_shapes(3)
V = 512
pre = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(pre).to(dev)
N = s['frames']
Npad = (N + 63) // 64 * 64
soa = torch.empty((V, 3, Npad), device=dev, dtype=torch.float32)
soa2 = torch.empty_like(soa)
dst = torch.empty_like(vecs)
psum = torch.empty((V * s['R'] * ctx.psum_stride(s['F']),), device=dev, dtype=torch.float64)
hist = torch.zeros((V, 72 * 36), device=dev, dtype=torch.float64)
vecsum = torch.zeros((V, 3), device=dev, dtype=torch.float64)
nblk = N // s['F']
outer = torch.zeros((nblk, V, 6), device=dev, dtype=torch.float64)
edges = (np.linspace(-np.pi, np.pi, 73), np.linspace(-1.0, 1.0, 37))
q = np.asarray(synth.Q_EXT, dtype=np.float64)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ctx.set_stream(0)
ctx.pack_soa_dev(vecs.data_ptr(), N, V, 0, V, soa.data_ptr(), Npad)
torch.cuda.synchronize()


def ct():
    ctx.ct_sums_dev(soa.data_ptr(), Npad, s['R'], s['F'], V, psum.data_ptr())


def b_pack():
    ctx.pack_soa_dev(vecs.data_ptr(), N, V, 0, V, soa2.data_ptr(), Npad)


def b_copy():
    dst.copy_(vecs, non_blocking=True)


def b_hist():
    ctx.rotate_hist_dev(soa.data_ptr(), Npad, N, V, q, edges[0], edges[1], hist.data_ptr(), vecsum.data_ptr(), outer.data_ptr(), s['F'])


def alone(fn, stream, reps=20):
    ctx.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def beside(fn_b, nb):
    """K C(t) launches on stream A while stream B runs nb operations (enqueued first, so that they are there from the start)"""
    torch.cuda.synchronize()
    evs = []
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fn_b is not None:
        ctx.set_stream(sb.cuda_stream)
        with torch.cuda.stream(sb):
            for _ in range(nb):
                fn_b()
                e = torch.cuda.Event(enable_timing=True)
                e.record(sb)
                evs.append(e)
    ctx.set_stream(sa.cuda_stream)
    with torch.cuda.stream(sa):
        ea0.record(sa)
        for _ in range(K):
            ct()
        ea1.record(sa)
    torch.cuda.synchronize()
    T = ea0.elapsed_time(ea1)
    done = sum(1 for e in evs if ea0.elapsed_time(e) <= T)
    return T, done


for _ in range(3):
    beside(None, 0)
t_ct = alone(ct, sa)
print('alone: C(t) %.4f ms' % t_ct, flush=True)
for name, fn in (('pack', b_pack), ('copy 614 MB', b_copy), ('histogram', b_hist)):
    if os.environ.get('ONLY') and os.environ['ONLY'] not in name:
        continue
    t_b = alone(fn, sb)
    nb = int(1.3 * K * t_ct / t_b) + 4
    res = []
    for rep in range(3):
        T0, _ = beside(None, 0)
        T, done = beside(fn, nb)
        res.append((T0, T, done))
    T0 = float(np.median([r[0] for r in res]))
    T = float(np.median([r[1] for r in res]))
    done = float(np.median([r[2] for r in res]))
    print('%-12s alone %.4f ms | %d C(t) launches: %.2f ms alone, %.2f ms beside %d of them -> %.4f ms per operation (%.0f %% of its own time)'
          % (name, t_b, K, T0, T, done, (T - T0) / max(done, 1), 100 * (T - T0) / max(done, 1) / t_b), flush=True)
ctx.close()
