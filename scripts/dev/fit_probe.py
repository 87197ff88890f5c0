#!/usr/bin/env python3
"""Developer probe of the model-order search on the cfg3 batch: one batch alone (ms, evaluations by order, the slowest
residue) and the chip saturated (K batches' residues, longest first).  usage: fit_probe.py [K] [option=value ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402
from spinrelax_amd.pipeline import DevicePipeline    # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    s = synth.config_shapes(3)
    V = 512
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
    ctx = Context(0)
    for a in sys.argv[2:]:
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

        def timed(fn, reps=3):
            out = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(base)
                fn()
                b.record(base)
                b.synchronize()
                out.append(a.elapsed_time(b))
            return min(out)
        alone = timed(lambda: p1.stage_fit(s0))
        p1.step(vecs)
        nf = s0.result['nfev']
        import hashlib
        print('alone %.3f ms | evaluations by order %s | slowest residue %d evaluations (order-9 max %d) | table sha %s'
              % (alone, nf.sum(axis=1).tolist(), int(nf.sum(axis=0).max()), int(nf[-1].max()),
                 hashlib.sha256(np.ascontiguousarray(s0.result['relax']).tobytes()).hexdigest()[:12]), flush=True)
        listDoG = p1.listDoG
        f64 = dict(device=dev, dtype=torch.float64)
        i32 = dict(device=dev, dtype=torch.int32)
        nO, Pmax, Kmax = len(listDoG), max(listDoG), max(listDoG) // 2
        cost = nf.sum(axis=0)
        order = torch.from_numpy(np.argsort(-cost, kind='stable').copy()).to(dev)
        tK = p1.t_dev.repeat(K, 1)
        yK, dK = s0.CtT[order].repeat_interleave(K, dim=0), s0.dCtT[order].repeat_interleave(K, dim=0)
        oK = dict(popt=torch.empty((nO, K * V, Pmax), **f64), dP=torch.empty((nO, K * V, Pmax), **f64), chisq=torch.empty((nO, K * V), **f64),
                  status=torch.empty((nO, K * V), **i32), nfev=torch.empty((nO, K * V), **i32), best=torch.empty((K * V,), **i32),
                  S2=torch.empty((K * V,), **f64), C=torch.empty((K * V, Kmax), **f64), tau=torch.empty((K * V, Kmax), **f64),
                  chi=torch.empty((K * V,), **f64), Kc=torch.empty((K * V,), **i32), work=torch.empty((K * V, s['L']), **f64))
        torch.cuda.synchronize()

        def fitK():
            ctx.order_search_dev(tK.data_ptr(), yK.data_ptr(), dK.data_ptr(), K * V, s['L'], listDoG, p1.tau_guess.data_ptr(), 1,
                                 p1.tau_max, p1.chi_thr, oK['popt'].data_ptr(), oK['dP'].data_ptr(), oK['chisq'].data_ptr(),
                                 oK['status'].data_ptr(), oK['nfev'].data_ptr(), oK['best'].data_ptr(), oK['S2'].data_ptr(),
                                 oK['C'].data_ptr(), oK['tau'].data_ptr(), oK['chi'].data_ptr(), oK['Kc'].data_ptr(),
                                 work_ptr=oK['work'].data_ptr())
        print('saturated (K = %d, longest first): %.4f ms per batch' % (K, timed(fitK, reps=2) / K), flush=True)
        p1.close()
    ctx.close()


if __name__ == '__main__':
    main()
