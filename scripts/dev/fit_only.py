#!/usr/bin/env python3
"""Developer probe: the model-order search kernel alone on the benchmark data (for rocprofv3 counters / timing)."""
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
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
        pipe = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT, Diso=synth.DISO,
                              aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1, stream=base)
        pipe.step(vecs)
        sl = pipe.slots[0]
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.stage_fit(sl)
            torch.cuda.synchronize()
            print('fit %.3f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
        pipe.stage_relax(sl)
        pipe.stage_download(sl)
        torch.cuda.synchronize()
        r = sl.host_results()
        tried = r['status'] != -100
        print('nfev per order', [int(r['nfev'][j][tried[j]].sum()) for j in range(5)], 'max', [int(r['nfev'][j].max()) for j in range(5)])


if __name__ == '__main__':
    main()
