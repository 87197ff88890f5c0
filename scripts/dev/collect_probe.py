#!/usr/bin/env python3
"""Developer probe: what the host does behind the last kernel of a 20-batch grouped run (GroupedPipeline.collect): wait for
the group, copy the results out of the pinned mirrors, split them per batch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd import pipeline as P
s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
pipe = P.GroupedPipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], group=32, q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                         field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, stream=torch.cuda.Stream(device=dev))
pipe.prime(vecs)
marks = []
orig_collect = P.GroupedPipeline.collect
orig_host = P._Group.host_results
def host_results(self):
    t0 = time.perf_counter(); r = orig_host(self); marks.append(('host_results', (time.perf_counter() - t0) * 1e3)); return r
def collect(self, grp, on_finished=None):
    t0 = time.perf_counter(); grp.done.synchronize(); t1 = time.perf_counter()
    r = orig_collect(self, grp, on_finished)
    marks.append(('wait_for_group', (t1 - t0) * 1e3)); marks.append(('collect_total_after_wait', (time.perf_counter() - t1) * 1e3)); return r
P._Group.host_results = host_results
P.GroupedPipeline.collect = collect
for _ in range(30):
    pipe.run(vecs, 20); torch.cuda.synchronize()
for rep in range(6):
    marks.clear()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(20)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.run(vecs, 20, ev); t_run = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    fit_end = ev[0][0].elapsed_time(ev[0][5])
    print('region %.2f ms (run() returned at %.2f) | first C(t) start -> fit end %.2f | %s' % ((t1 - t0) * 1e3, (t_run - t0) * 1e3, fit_end,
          ', '.join('%s %.2f' % m for m in marks)), flush=True)
pipe.close(); ctx.close()
