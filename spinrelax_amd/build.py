"""
Build libspinrelax_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build(); also
runnable as ``python -m spinrelax_amd.build``.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libspinrelax_hip.so')
SOURCES = ['sr_core.hip', 'sr_ct.hip', 'sr_vechist.hip', 'sr_fit.hip', 'sr_relax.hip', 'sr_dq.hip', 'sr_traj.hip', 'sr_vectors.hip', 'sr_textio.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']
# sr_ct.hip: the SLP vectoriser packs the FMAs of the C(t) inner loop into v_pk_fma_f32, whose operand pairs then
# need ~1 v_mov per FMA (rocprofv3: 4.8e9 VALU instructions for 2.5e9 FMAs); plain v_fma_f32 issues at the same rate.
EXTRA = {'sr_ct.hip': ['-fno-slp-vectorize'],
         # sr_fit.hip: explicit fma() only, see the note at the top of the file
         'sr_fit.hip': ['-ffp-contract=off']}
if os.environ.get('SR_FIT_DEV_FAST'):          # development: only the order-search variants the benchmark uses
    EXTRA['sr_fit.hip'] = EXTRA['sr_fit.hip'] + ['-DSR_FIT_DEV_FAST']


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, 'sr_internal.h'), os.path.join(HERE, '..', 'include', 'spinrelax_hip.h')]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.isfile(s):
            raise FileNotFoundError(s)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
