"""cProfile of the three drop-in scripts on the bench workload (cfg3): where the wall time of cli_wall_s goes.
usage (GPU box): python scripts/dev/cli_profile.py [outdir]"""
import os, sys, subprocess, tempfile, time, pstats, shutil, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spinrelax_amd import synth
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'cliprof')
os.makedirs(out, exist_ok=True)
s = synth.config_shapes(3)
vecs = synth.synth_vectors_parallel(s['frames'], s['V'], s['seed'])
tmp = tempfile.mkdtemp(prefix='sr_cli_')
fn = os.path.join(tmp, 'vecs.npy')
np.save(fn, vecs)
pref = os.path.join(tmp, 'rotdif')
scr = os.path.join(ROOT, 'scripts')
steps = [('ct', ['calculate-Ct-from-traj.py', '-s', 'reference.pdb', '-f', fn, '--dt', str(s['dt']), '--tau', str(s['tau_memory']), '-o', pref,
                 '--vecHist', '--binary', '--vecAvg', '--S2', '--Ct', '--vecRot', ' '.join('%.6f' % x for x in synth.Q_EXT)]),
         ('fit', ['calculate-fitted-Ct.py', '-f', pref + '_Ctint.dat', '-o', pref]),
         ('relax', ['calculate-relaxations-from-Ct.py', '-f', pref + '_fittedCt.dat', '-o', pref + '-600', '-F', '%ge6' % synth.FIELD_MHZ, '--tu', 'ps',
                    '--zeta', str(synth.ZETA), '--distfn', pref + '_vecHistogram.npz', '-D', '%g %g' % (synth.DISO, synth.DANI)])]
for rep in range(2):
    for name, cmd in steps:
        prof = os.path.join(out, name + '.prof')
        t0 = time.time()
        p = subprocess.run([sys.executable, '-m', 'cProfile', '-o', prof, os.path.join(scr, cmd[0])] + cmd[1:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        dt = time.time() - t0
        t0 = time.time()
        p2 = subprocess.run([sys.executable, os.path.join(scr, cmd[0])] + cmd[1:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        dt2 = time.time() - t0
        print('==== %s rep %d: %.2f s under cProfile, %.2f s plain (rc %d %d)' % (name, rep, dt, dt2, p.returncode, p2.returncode), flush=True)
        if rep == 1:
            sio = io.StringIO()
            pstats.Stats(prof, stream=sio).sort_stats('cumulative').print_stats(45)
            txt = sio.getvalue()
            open(os.path.join(out, name + '.txt'), 'w').write(txt)
            t0 = time.time()
            subprocess.run([sys.executable, '-X', 'importtime', '-c', 'import runpy'], stderr=subprocess.DEVNULL)
print({f: os.path.getsize(os.path.join(tmp, f)) for f in sorted(os.listdir(tmp))})
shutil.rmtree(tmp, ignore_errors=True)
