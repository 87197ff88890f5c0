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
SOURCES = ['sr_core.hip', 'sr_ct.hip', 'sr_ct32.hip', 'sr_vechist.hip', 'sr_fit.hip', 'sr_relax.hip', 'sr_dq.hip', 'sr_traj.hip', 'sr_vectors.hip', 'sr_textio.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']
# sr_ct.hip: the SLP vectoriser packs the FMAs of the C(t) inner loop into v_pk_fma_f32, whose operand pairs then
# need ~1 v_mov per FMA (rocprofv3: 4.8e9 VALU instructions for 2.5e9 FMAs); plain v_fma_f32 issues at the same rate.
EXTRA = {'sr_ct.hip': ['-fno-slp-vectorize'],
         # sr_ct32.hip: the same for the float32 transforms (packed complex arithmetic by the vectoriser costs a v_mov per
         # operand pair and 270 B of scratch at 128 VGPRs; without it 56 B)
         'sr_ct32.hip': ['-fno-slp-vectorize'],
         # sr_fit.hip: explicit fma() only, see the note at the top of the file
         'sr_fit.hip': ['-ffp-contract=off']}
if os.environ.get('SR_FIT_DEV_FAST'):          # development: only the order-search variants the benchmark uses
    EXTRA['sr_fit.hip'] = EXTRA['sr_fit.hip'] + ['-DSR_FIT_DEV_FAST']


def build_id():
    """sha256 (first 16 hex characters) over every source, header and compiler flag that goes into the library: what
    sr_build_id() of a library built by this module returns, and what scripts/make_profiles.py stores beside the PMC figures
    of a profiling run (bench.py compares the two)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + ['sr_internal.h']:
        with open(os.path.join(CSRC, name), 'rb') as fp:
            h.update(name.encode() + b'\0' + fp.read())
    with open(os.path.join(HERE, '..', 'include', 'spinrelax_hip.h'), 'rb') as fp:
        h.update(b'spinrelax_hip.h\0' + fp.read())
    h.update(repr((FLAGS, sorted(EXTRA.items()))).encode())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _same_flags(stamp, sig):
    try:
        with open(stamp) as fp:
            return fp.read() == sig
    except OSError:
        return False


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, 'sr_internal.h'), os.path.join(HERE, '..', 'include', 'spinrelax_hip.h')]
    objs = []
    bid = build_id()
    idfile = os.path.join(CSRC, '.build_id')
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.isfile(s):
            raise FileNotFoundError(s)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        extra = EXTRA.get(src, [])
        if src == 'sr_core.hip':                               # sr_core.o carries the id: its command line changes with it
            extra = extra + ['-DSR_BUILD_ID="%s"' % bid]
        cmd = [HIPCC] + FLAGS + extra + ['-c', s, '-o', o]
        # an object is rebuilt when its source or a header is newer, and when its COMMAND LINE differs from the one it was built
        # with (<object>.cmd): a flag change must not leave an object of the old flags under the new build id
        stamp = o + '.cmd'
        sig = ' '.join(FLAGS + extra + [src])                  # no absolute paths: the tree is copied to the GPU box as it is
        if force or _stale(o, [s] + hdrs) or not _same_flags(stamp, sig):
            jobs.append((src, cmd, stamp, sig))
        objs.append(o)

    def compile_one(job):
        src, cmd, stamp, sig = job
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, 'w') as fp:
            fp.write(sig)

    if jobs:
        # the sources compile side by side (sr_fit.hip alone takes minutes: the longest first); at most four at a time, each
        # hipcc is one process of ~2 GB
        from concurrent.futures import ThreadPoolExecutor
        jobs.sort(key=lambda j: 0 if j[0] == 'sr_fit.hip' else 1)
        with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, jobs))
    with open(idfile, 'w') as fp:
        fp.write(bid + '\n')
    if force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
