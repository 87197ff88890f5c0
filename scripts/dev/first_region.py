"""Developer probe: why is the first timed region of bench.py slower than the next ones?  Sequence of runs of different
lengths, each timed between two synchronisations."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')
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
last = {}
def run(n, ev=False):
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(n)] if ev else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.run(vecs, n, events); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) * 1e3
    if ev:
        z = events[0][0]
        last['tl'] = 'first C(t) start..last C(t) end %.1f ms, fit %.1f -> %.1f, hist first/last end %.1f / %.1f' % (
            z.elapsed_time(events[-1][1]), z.elapsed_time(events[0][4]), z.elapsed_time(events[0][5]), z.elapsed_time(events[0][3]), z.elapsed_time(events[-1][3]))
    return t
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    run(20)
seq = [('20ev', 20, True), ('20ev', 20, True), ('5', 5, False), ('20ev', 20, True), ('20ev', 20, True), ('20ev', 20, True), ('5', 5, False), ('20ev', 20, True), ('20ev', 20, True),
       ('12', 12, False), ('20ev', 20, True), ('20ev', 20, True), ('32', 32, False), ('20ev', 20, True), ('20ev', 20, True)]
for name, n, ev in seq:
    if n == 0:
        time.sleep(0.02); print('sleep 20 ms'); continue
    print('%-5s %.2f ms  (%.3f per step)' % (name, run(n, ev), run.__defaults__ and 0 or 0), flush=True) if False else None
    t = run(n, ev)
    print('%-5s %.2f ms  %.3f per step   %s' % (name, t, t / n, last.pop('tl', '')), flush=True)
pipe.close(); ctx.close()
