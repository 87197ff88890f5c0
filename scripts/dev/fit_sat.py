#!/usr/bin/env python3
"""Developer probe: the model-order search with the chip saturated (K batches' residues in one launch, longest first) --
for rocprofv3 counters.  usage: fit_sat.py [K] [reps] [option=value ...]"""
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    s = synth.config_shapes(3)
    V = 512
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
    ctx = Context(0)
    for a in sys.argv[3:]:
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
        tK = p1.t_dev.repeat(K, 1)
        yK, dK = s0.CtT[order].repeat_interleave(K, dim=0), s0.dCtT[order].repeat_interleave(K, dim=0)
        oK = dict(popt=torch.empty((nO, K * V, Pmax), **f64), dP=torch.empty((nO, K * V, Pmax), **f64), chisq=torch.empty((nO, K * V), **f64),
                  status=torch.empty((nO, K * V), **i32), nfev=torch.empty((nO, K * V), **i32), best=torch.empty((K * V,), **i32),
                  S2=torch.empty((K * V,), **f64), C=torch.empty((K * V, Kmax), **f64), tau=torch.empty((K * V, Kmax), **f64),
                  chi=torch.empty((K * V,), **f64), Kc=torch.empty((K * V,), **i32), work=torch.empty((K * V, s['L']), **f64))
        torch.cuda.synchronize()
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.order_search_dev(tK.data_ptr(), yK.data_ptr(), dK.data_ptr(), K * V, s['L'], listDoG, p1.tau_guess.data_ptr(), 1,
                                 p1.tau_max, p1.chi_thr, oK['popt'].data_ptr(), oK['dP'].data_ptr(), oK['chisq'].data_ptr(),
                                 oK['status'].data_ptr(), oK['nfev'].data_ptr(), oK['best'].data_ptr(), oK['S2'].data_ptr(),
                                 oK['C'].data_ptr(), oK['tau'].data_ptr(), oK['chi'].data_ptr(), oK['Kc'].data_ptr(),
                                 work_ptr=oK['work'].data_ptr())
            torch.cuda.synchronize()
            print('K=%d: %.3f ms per batch' % (K, (time.perf_counter() - t0) * 1e3 / K), flush=True)
        p1.close()
    ctx.close()


if __name__ == '__main__':
    main()
