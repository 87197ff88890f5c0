"""Developer probe: what the host side of GroupedPipeline.collect costs (copy out of the pinned mirrors + split per batch)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd.pipeline import GroupedPipeline
s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
pipe = GroupedPipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], group=32, q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                       field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, stream=torch.cuda.Stream(device=dev))
pipe.prime(vecs)
pipe.run(vecs, 20)
grp = pipe.groups[0] if pipe.groups[0].g == 20 else pipe.groups[1]
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); nd, ni = grp._sizes(20); a = grp.h_dres[:nd].copy(); b = grp.h_ires[:ni].copy(); t1 = time.perf_counter()
    r = grp.host_results(); t2 = time.perf_counter()
    grp.busy = True; grp.done = torch.cuda.Event(); grp.done.record(); pipe.collect(grp); t3 = time.perf_counter()
    print('mirror copy %.2f ms (%.1f MB), host_results %.2f ms, collect %.2f ms' % ((t1 - t0) * 1e3, (a.nbytes + b.nbytes) / 1e6, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
x = np.empty(nd); t0 = time.perf_counter(); y = x.copy(); print('pageable copy of the same size %.2f ms' % ((time.perf_counter() - t0) * 1e3))
pipe.close(); ctx.close()
