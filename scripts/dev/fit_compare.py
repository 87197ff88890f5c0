#!/usr/bin/env python3
"""Developer probe: one-launch model-order search vs the host-driven search (k_trf per order) on the full benchmark
batch (512 residues, L = 2048): are the two bit-identical for every residue and order?"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402
from spinrelax_amd.pipeline import DevicePipeline    # noqa: E402
from spinrelax_amd import fitting_Ct_functions as fitCt   # noqa: E402


def main():
    s = synth.config_shapes(3)
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
    ctx = Context(0)
    dev = torch.device('cuda', 0)
    vecs = torch.from_numpy(vecs_host).to(dev)
    base = torch.cuda.Stream(device=dev)
    ctx.set_stream(base.cuda_stream)
    with torch.cuda.stream(base):
        pipe = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT, Diso=synth.DISO,
                              aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1, stream=base)
        pipe.step(vecs)
        sl = pipe.slots[0]
        y = sl.CtT.cpu().numpy()
        dy = sl.dCtT.cpu().numpy()
        r = {k: v.copy() for k, v in sl.result.items()}
    ctx.set_stream(0)
    t = pipe.t_host
    search = fitCt.OrderSearchBatch(t, y, pipe.listDoG, 0.5)
    runner = fitCt.host_runner(t, y, dy, ctx=ctx)
    while True:
        req = search.request()
        if req is None:
            break
        search.submit(*runner(req['nParams'], req['p0'], req['idx']))
    print('best identical:', np.array_equal(r['best'], search.best))
    for j, res in enumerate(search.per_order):
        nP = res['nParams']
        tried = r['status'][j] != -100
        idx = np.flatnonzero(tried)
        same = np.all(r['popt'][j][idx, :nP] == res['popt'][idx], axis=1)
        print('order %d: %d residues, %d differ (max rel diff of chi %.2e)' % (
            nP, idx.size, int((~same).sum()), np.nanmax(np.abs(r['chisq'][j][idx] / res['chiSq'][idx] - 1))))


if __name__ == '__main__':
    main()
