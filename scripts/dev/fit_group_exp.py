#!/usr/bin/env python3
"""Developer probe: the model-order search of G batches merged into ONE launch, in different residue orders:
sorted (longest first: what bench.py's saturated figure uses), natural (batch after batch, residues as they come: what a
pipeline can do without knowing the costs), interleaved (residue r of every batch next to each other).
usage: fit_group_exp.py"""
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402
from spinrelax_amd.pipeline import DevicePipeline    # noqa: E402


def main():
    reps = 2
    s = synth.config_shapes(3)
    V = 512
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
    ctx = Context(0)
    for a in sys.argv[1:]:
        k, v = a.split('=')
        ctx.set_option(k, int(v))
    dev = torch.device('cuda', 0)
    vecs = torch.from_numpy(vecs_host).to(dev)
    base = torch.cuda.Stream(device=dev)
    ctx.set_stream(base.cuda_stream)
    with torch.cuda.stream(base):
        p1 = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT, Diso=synth.DISO,
                            aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1, stream=base)
        p1.step(vecs)
        s0 = p1.slots[0]
        listDoG = p1.listDoG
        f64 = dict(device=dev, dtype=torch.float64)
        i32 = dict(device=dev, dtype=torch.int32)
        nO, Pmax, Kmax = len(listDoG), max(listDoG), max(listDoG) // 2
        cost = s0.result['nfev'].sum(axis=0)
        order = torch.from_numpy(np.argsort(-cost, kind='stable').copy()).to(dev)
        where = int(np.argmax(cost))
        print('slowest residue: index %d of %d, %d evaluations' % (where, V, int(cost.max())), flush=True)
        for K in (10, 20, 32):
            tK = p1.t_dev.repeat(K, 1)
            oK = dict(popt=torch.empty((nO, K * V, Pmax), **f64), dP=torch.empty((nO, K * V, Pmax), **f64), chisq=torch.empty((nO, K * V), **f64),
                      status=torch.empty((nO, K * V), **i32), nfev=torch.empty((nO, K * V), **i32), best=torch.empty((K * V,), **i32),
                      S2=torch.empty((K * V,), **f64), C=torch.empty((K * V, Kmax), **f64), tau=torch.empty((K * V, Kmax), **f64),
                      chi=torch.empty((K * V,), **f64), Kc=torch.empty((K * V,), **i32), work=torch.empty((K * V, s['L']), **f64))
            perm = torch.randperm(V, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
            permK = torch.randperm(K * V, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
            permK2 = torch.randperm(K * V, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            for name, (yK, dK) in (('sorted', (s0.CtT[order].repeat_interleave(K, dim=0), s0.dCtT[order].repeat_interleave(K, dim=0))),
                                   ('natural', (s0.CtT.repeat(K, 1), s0.dCtT.repeat(K, 1))),
                                   ('interleaved', (s0.CtT.repeat_interleave(K, dim=0), s0.dCtT.repeat_interleave(K, dim=0))),
                                   ('shuffled', (s0.CtT[perm].repeat(K, 1), s0.dCtT[perm].repeat(K, 1))),
                                   ('allshuffled', (s0.CtT.repeat(K, 1)[permK], s0.dCtT.repeat(K, 1)[permK])),
                                   ('allshuffled2', (s0.CtT.repeat(K, 1)[permK2], s0.dCtT.repeat(K, 1)[permK2]))):
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    ctx.order_search_dev(tK.data_ptr(), yK.data_ptr(), dK.data_ptr(), K * V, s['L'], listDoG, p1.tau_guess.data_ptr(), 1,
                                         p1.tau_max, p1.chi_thr, oK['popt'].data_ptr(), oK['dP'].data_ptr(), oK['chisq'].data_ptr(),
                                         oK['status'].data_ptr(), oK['nfev'].data_ptr(), oK['best'].data_ptr(), oK['S2'].data_ptr(),
                                         oK['C'].data_ptr(), oK['tau'].data_ptr(), oK['chi'].data_ptr(), oK['Kc'].data_ptr(),
                                         work_ptr=oK['work'].data_ptr())
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                print('G=%2d %-12s launch %.2f ms = %.3f ms per batch' % (K, name, min(ts), min(ts) / K), flush=True)
            del tK, oK
        p1.close()
    ctx.close()


if __name__ == '__main__':
    main()
