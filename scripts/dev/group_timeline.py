"""Developer probe: timeline of one grouped run (GroupedPipeline): when every C(t) / histogram / merged fit launch starts
and ends relative to the first C(t) launch, and the host wall time of the run.  usage: group_timeline.py [K] [group] [overlap 0/1]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd.pipeline import GroupedPipeline

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ov = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
pipe = GroupedPipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], group=G, overlap=ov, q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                       field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, stream=torch.cuda.Stream(device=dev))
if len(sys.argv) > 4:
    pipe.sizes_override = [int(x) for x in sys.argv[4].split(',')]
pipe.gate_next = bool(os.environ.get('GATE'))
pipe.prime(vecs)
for _ in range(6):
    pipe.run(vecs, K)
torch.cuda.synchronize()
for rep in range(2):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(vecs, K, ev)
    t_host_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    z = ev[0][0]
    def T(e):
        try:
            return z.elapsed_time(e)
        except Exception:
            return float('nan')
    print('run %d: wall %.2f ms (host returned from run() after %.2f ms)' % (rep, wall * 1e3, t_host_enq * 1e3))
    continue_ = os.environ.get('BRIEF')
    if continue_:
        print('  fit  start/end:', ' '.join('%.1f/%.1f' % (T(e[4]), T(e[5])) for e in ev if T(e[4]) == T(e[4])), ' last C(t) end %.1f' % max(T(e[1]) for e in ev))
        continue
    print('  C(t) start:', ' '.join('%.1f' % T(e[0]) for e in ev))
    print('  C(t) end  :', ' '.join('%.1f' % T(e[1]) for e in ev))
    print('  hist start:', ' '.join('%.1f' % T(e[2]) for e in ev))
    print('  hist end  :', ' '.join('%.1f' % T(e[3]) for e in ev))
    print('  fit  start/end:', ' '.join('%.1f/%.1f' % (T(e[4]), T(e[5])) for e in ev if T(e[4]) == T(e[4])))
pipe.close()
ctx.close()
