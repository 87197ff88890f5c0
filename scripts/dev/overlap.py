#!/usr/bin/env python3
"""Developer probe: how long does a fit launch take while a C(t) launch fills the chip, for different stream set-ups?"""
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                      # noqa: E402
from spinrelax_amd.hip import Context                # noqa: E402
from spinrelax_amd.pipeline import DevicePipeline, _EdgeOnly    # noqa: E402
from spinrelax_amd import fitting_Ct_functions as fitCt        # noqa: E402


def words(bits):
    w = [0] * 8
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return w


def main():
    s = synth.config_shapes(3)
    V = 512
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
        head = sl.CtT[:, :10].cpu().numpy()
        tail = sl.CtT[:, -10:].cpu().numpy()
    torch.cuda.synchronize()
    reqs = {}
    for nP in (2, 3, 7):
        srch = fitCt.OrderSearchBatch(pipe.t_host, _EdgeOnly(head, tail, pipe.L), (nP,))
        reqs[nP] = srch.request()

    allb = list(range(256))
    A_opts = {'A plain': None, 'A mask224': words(allb[:224])}
    B_opts = {'B prio': 'prio', 'B mask-all': words(allb), 'B mask-resv32': words(allb[224:])}

    def mk(opt):
        if opt is None:
            return torch.cuda.Stream(device=dev)
        if opt == 'prio':
            return torch.cuda.Stream(device=dev, priority=-1)
        return torch.cuda.ExternalStream(ctx.stream_create(opt), device=dev)

    def launch_ct(A):
        ctx.set_stream(A.cuda_stream)
        pipe.stage_ct(sl)

    def launch_fit(B, nP):
        ctx.set_stream(B.cuda_stream)
        with torch.cuda.stream(B):
            pipe._launch_fit(sl, reqs[nP])

    for an, ao in A_opts.items():
        for bn, bo in B_opts.items():
            A, B = mk(ao), mk(bo)
            torch.cuda.synchronize()
            for nP in (2, 3, 7):
                res = []
                for mode in ('fit alone', 'ct alone', 'both'):
                    for rep in range(2):
                        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        if mode != 'fit alone':
                            e[0].record(A)
                            launch_ct(A)
                            e[1].record(A)
                            time.sleep(0.002)
                        if mode != 'ct alone':
                            e[2].record(B)
                            launch_fit(B, nP)
                            e[3].record(B)
                        torch.cuda.synchronize()
                        wall = (time.perf_counter() - t0) * 1e3
                    ct_ms = e[0].elapsed_time(e[1]) if mode != 'fit alone' else float('nan')
                    fit_ms = e[2].elapsed_time(e[3]) if mode != 'ct alone' else float('nan')
                    res.append('%s: ct %.2f fit %.2f wall %.2f' % (mode, ct_ms, fit_ms, wall))
                print('%-10s %-14s nP=%d | %s' % (an, bn, nP, ' | '.join(res)), flush=True)


if __name__ == '__main__':
    main()
