"""Developer probe: what each bandwidth-bound stage costs the STEADY state of the grouped schedule.  Runs 120 steps with the stage in
place and with it left out (diagnostic only: the results of such a run are wrong or stale; the planes of an earlier pack stay valid
because the vectors do not change).   usage: stage_cost.py [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth
from spinrelax_amd.hip import Context
from spinrelax_amd.pipeline import GroupedPipeline

K = int(sys.argv[1]) if len(sys.argv) > 1 else 120
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
ctx = Context(0)
if os.environ.get('CT_WG_PER_CU'):
    ctx.set_option('ct_wg_per_cu', int(os.environ['CT_WG_PER_CU']))
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
pipe = GroupedPipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], group=32, q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                       field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, stream=torch.cuda.Stream(device=dev))
pipe.prime(vecs)
for _ in range(8):
    pipe.run(vecs, 32)                       # every plane buffer has been packed at least once
torch.cuda.synchronize()


def measure(label):
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run(vecs, K)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e3)
    print('%-34s %.3f %.3f %.3f ms per step (steady, %d steps)' % (label, ts[0], ts[1], ts[2], K), flush=True)
    return float(np.median(ts))


base = measure('all stages')
saved = {}
for name in ('stage_pack', 'stage_hist', 'stage_ct_finalize'):
    saved[name] = getattr(pipe, name)
    setattr(pipe, name, lambda *a, **k: None)
    v = measure('without ' + name)
    print('    -> %s costs the steady state %.3f ms per step' % (name, base - v))
    setattr(pipe, name, saved[name])
for name in ('stage_pack', 'stage_hist'):
    setattr(pipe, name, lambda *a, **k: None)
v = measure('without pack and histogram')
print('    -> both: %.3f ms per step' % (base - v))
for name in ('stage_pack', 'stage_hist'):
    setattr(pipe, name, saved[name])
measure('all stages (again)')
pipe.close()
ctx.close()
