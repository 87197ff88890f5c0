#!/usr/bin/env python3
"""Developer probe: how much of a grouped run is the HOST busy enqueueing (python + launch calls) as opposed to waiting for the
device in collect()?  Steady state is host-bound when the busy share approaches 100 %.  usage: host_busy.py [K] [group]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spinrelax_amd import synth                       # noqa: E402
from spinrelax_amd.hip import Context                 # noqa: E402
from spinrelax_amd.pipeline import GroupedPipeline    # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 120
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
s = synth.config_shapes(3)
V = 512
vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
pre = os.environ.get('PRE', '')          # what runs on the CPU before the GPU is initialised (bench.py's cpu_baseline leg does all three)
if pre:
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if 'omp' in pre:
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libsr_oracle.so'))
        R, F = s['R'], s['F']
        v4c = np.ascontiguousarray(vecs_host[:s['N'], :64].reshape(R, F, 64, 3), dtype=np.float32)
        Cc = np.empty((F // 2, 64), dtype=np.float32)
        dCc = np.empty_like(Cc)
        lib.sr_oracle_ct_palmer_f32_stream(v4c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(R), ctypes.c_int64(F), ctypes.c_int64(64),
                                           Cc.ctypes.data_as(ctypes.c_void_p), dCc.ctypes.data_as(ctypes.c_void_p))
    if 'scipy' in pre:
        from scipy.optimize import curve_fit
        x = np.linspace(0, 10, 2048)
        for _ in range(50):
            curve_fit(lambda t, a, b: a * np.exp(-t / b), x, np.exp(-x / 3.0), p0=(1.0, 1.0), bounds=(0, 10))
    if 'fft' in pre:
        np.fft.rfft(np.random.rand(16, 24, 8192), axis=-1)
    print('pre: %s done, threads now %d' % (pre, len(os.listdir('/proc/self/task'))), flush=True)
ctx = Context(0)
dev = torch.device('cuda', 0)
vecs = torch.from_numpy(vecs_host).to(dev)
pipe = GroupedPipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], group=G, q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                       field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, stream=torch.cuda.Stream(device=dev))
pipe.prime(vecs)
for _ in range(3):
    pipe.run(vecs, 40)
torch.cuda.synchronize()
wait = [0.0]
orig = pipe.collect


def collect(grp, on_finished=None):
    t0 = time.perf_counter()
    grp.done.synchronize()
    wait[0] += time.perf_counter() - t0
    return orig(grp, on_finished)


pipe.collect = collect
for rep in range(3):
    wait[0] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(vecs, K)
    t_run = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('K = %d: wall %.1f ms (%.3f per step); host inside run() %.1f ms, of which waiting for the device %.1f ms -> host busy %.3f ms per step'
          % (K, wall * 1e3, wall * 1e3 / K, t_run * 1e3, wait[0] * 1e3, (t_run - wait[0]) * 1e3 / K), flush=True)
pipe.close()
ctx.close()
