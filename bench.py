#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X SpinRelax hot path (BASELINE.json metric:
frame*vector*lag triples/s for the C(t) + R1/R2/NOE pipeline).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL backend)

A "step" is one pass of the whole hot path over one synthetic trajectory shard that is already
resident in HBM: pack -> C(t) -> rotation + spherical histogram -> multi-exponential fits with the
model-order search -> J(omega)/R1/R2/NOE/rho, and for N > 1 the RCCL all-gather of the per-shard results
(SURVEY.md section 8(e)).  Workload: BASELINE.json configs[2] (100 000 frames x 512 vectors, 2 048 lags,
axisymmetric D, q_ext rotation + vecHistogram) -- the configuration the north-star quotes its scaling on --
or, with `--workload cfg4`, configs[3] (100 000 x 2 048).  `--scaling strong` (default): the workload's vectors are
FIXED and sharded over the ranks as the product shards them (spinrelax_amd/dist.py:shard_range, contiguous vector ranges:
calculate-Ct-from-traj.py:225-228 is independent per vector), so N = 1 is the BENCH measurement and N = 8 of cfg3 leaves
64 vectors per GPU (cfg4: 256 per GPU = BASELINE configs[3]); `--scaling weak`: every rank owns its own 512 vectors
(rank r = vectors 512 r .. 512 r + 511 of one synthetic trajectory; linear by construction, there is no data-path
collective).  `value` = exact triples of all ranks / wall time of a step of the slowest rank (throughput); `latency_ms` = one
batch alone, start to results on the host.  Schedule (spinrelax_amd/pipeline.py:GroupedPipeline, `--group`, default 32): the
C(t) / histogram / chunk-statistics kernels of a group of steps back to back, then ONE merged model-order search and ONE
relaxation launch over the group's residues; a run of K steps is cut into groups of min(K, 32), every step's work is done
inside the timed region.  `--group 1` is the per-step schedule of rounds 1-2 (every step launches its own fits, `--depth`
steps in flight).

One JSON line is printed by rank 0 (contract in the task statement) with extra objects:
  kernels       one entry per kernel of a step, every `frac` a fraction of a bound THAT kernel can reach, from what it
                executes:  k_ct_rfft / k_ct_fft -- executed float64 flop / duration against the FP64 vector peak;  k_ct_palmer (the
                direct formulation, timed alone as a second line) -- 8 flop x exact triples against the FP32 vector peak;
                k_vechist, k_pack_soa -- algorithmic bytes / duration against the HBM peak;  k_order_search -- residues/s,
                evaluations/s and the executed float64 flop (PMC count of the committed profile: the data are
                deterministic) / duration against the FP64 vector peak.  Durations: `in_pipeline_ms` from HIP events placed
                around the launch on the stream it runs on, inside the timed region; `alone_ms` the same launch by itself
                on the whole chip after the timed region; `cu_ms_per_batch` = 256 CUs x the time a batch's worth of that
                kernel takes when the chip is saturated with it (the fit: 32 batches' residues, longest first, in one launch / 32).
  roofline      the `kernels` entry with the largest cu_ms_per_batch (the kernel that occupies most of the chip), in the
                contract's shape {bound, achieved, peak, unit, frac, traffic}; `direct_equivalent` keeps the reference
                formulation's 8 flop / 24 B per triple over the C(t) kernel's duration for comparison only.
  cpu_baseline  the reference's algorithm on this host (rank 0, N = 1) on a bounded sample of the SAME path the GPU value
                covers: per-lag float32 numpy einsum C(t) (calculate-Ct-from-traj.py:222-228), scipy curve_fit model-order
                search (fitting_Ct_functions.py:278-345), per-residue J(w)/R1/R2/NOE over the 2 592 bins
                (calculate-relaxations-from-Ct.py:158-176), restated in oracle/sr_oracle.py -- 1 thread like the reference;
                plus the CPU-FFT formulation (like-for-like with k_ct_fft) and the OpenMP C loop on all cores.
Peaks: FP32 vector 157.3 TFLOP/s and HBM 8 TB/s from /opt/skills/guides/MI355X_MICROARCH.md; FP64 vector 78.6 TFLOP/s from
the AMD Instinct MI355X data sheet (peak double-precision vector; = half the FP32 vector rate of the guide's table).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per stream the pipeline uses (two C(t) streams, auxiliary, chunk statistics, two group streams, the gather;
# `--group 1`: one per in-flight batch + the C(t) streams); the HIP runtime
# multiplexes streams onto 4 queues by default and streams that share a queue serialise (spinrelax_amd/pipeline.py).
# Not more than needed: two processes with 16 queues each on ONE GPU oversubscribe the hardware queues and the
# scheduler's time-slicing makes a step 20x slower (seen in the 2-rank rehearsal on one device).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')

PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: FP32 vector = FP32 matrix peak
PEAK_FP64_TFLOPS = 78.6           # AMD Instinct MI355X data sheet: peak FP64 vector (half the FP32 vector rate)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
N_CU = 256


def committed_profile():
    """Per-kernel figures of the committed rocprofv3 PMC passes (profiles/, scripts/make_profiles.py): HBM bytes per
    launch and, for the fit kernel, executed float64 flop per launch.  NOT measured in this run (PMC counters need
    rocprofv3); the inputs are deterministic, so the per-launch counts carry over."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_cfg3_hbm_counters.json')))
    if not files:
        return {}, None
    with open(files[-1]) as fp:
        d = json.load(fp)
    global PROFILE_BUILD_ID
    PROFILE_BUILD_ID = d.get('build_id')
    return d.get('kernels', {}), os.path.relpath(files[-1], ROOT)


PROFILE_BUILD_ID = None


def fp64_issue_probe():
    """scripts/dev/probe/fp64_rate.hip on MI355X (committed output profiles/r*_fp64_issue_rate.txt): float64 VALU wave-instructions
    one SIMD issues per microsecond as a function of the waves it holds -- 8 independent chains per wave, add / mul / fma alike.
    The nominal rate (peak 78.6 TFLOP/s = 1 024 SIMDs x 64 lanes x 2 flop / 1.667 ns) is 600 per us; a kernel whose registers allow
    two waves per SIMD cannot issue faster than the two-wave figure."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_fp64_issue_rate.txt')))
    if not files:
        return None
    acc = {}
    with open(files[-1]) as fp:
        for line in fp:
            m = re.match(r'(\d+) wave\(s\) per SIMD\s+(v_\w+_f64)\s+.*kernel ([\d.]+) ms', line)
            if m:
                acc.setdefault(int(m.group(1)), []).append(int(m.group(1)) * 20000 * 16 / (float(m.group(3)) * 1e3))
    if not acc:
        return None
    return {'source': os.path.relpath(files[-1], ROOT), 'nominal_instr_per_us_per_simd': 600.0,
            'instr_per_us_per_simd_by_waves': {str(k): float(np.mean(v)) for k, v in sorted(acc.items())}}


def profile_staleness():
    """Do the committed PMC figures belong to the kernels that ran?  The profile stores the build id (sha256 over the HIP
    sources, headers and compiler flags, spinrelax_amd/build.py:build_id) of the library it was collected with; the loaded
    library reports its own (sr_build_id)."""
    from spinrelax_amd import _lib
    try:
        mine = _lib.load().sr_build_id().decode()
    except Exception:
        mine = 'unknown'
    stale = PROFILE_BUILD_ID is None or PROFILE_BUILD_ID != mine
    return stale, mine


def prof_entry(prof, prefix):
    for k, v in prof.items():
        if k == prefix or k.startswith(prefix + '<'):
            return v
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed batches (the pipeline keeps --depth of them in flight; its fill and drain are inside the timed region)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--repeats', type=int, default=3, help='the timed region (exactly --steps steps between two synchronisations) is run this many times; value / ms_per_step are the MEDIAN, every sample is listed')
    ap.add_argument('--spinup-s', type=float, default=1.0, help='untimed seconds of the same pipeline before the warm-up steps: a fresh box needs about a second under load before the clocks settle (5 warm-up steps are 15 ms)')
    ap.add_argument('--steady-steps', type=int, default=120, help='steps of the extra (untimed for the headline) long run behind ms_per_step_steady; 0 = skip')
    ap.add_argument('--workload', type=str, default='cfg3', choices=['cfg2', 'cfg3', 'cfg4'])
    ap.add_argument('--scaling', type=str, default='strong', choices=['strong', 'weak'],
                    help='strong (default): the workload\'s vectors are fixed and sharded over the ranks (N = 1 is the BENCH run); weak: every rank owns the full vector count')
    ap.add_argument('--vectors', type=int, default=None, help='vectors of the workload (strong scaling: in total, sharded over the ranks; weak: per GPU); default: the config value')
    ap.add_argument('--reserve-cus', type=int, default=0, help='CUs kept free of the C(t)/histogram kernels for the latency-bound fit kernels (CU-masked stream; 0 = no partition)')
    ap.add_argument('--fits-on-reserved-only', type=int, default=1, help='1/0: confine the fit kernels to the reserved CUs (strict partition)')
    ap.add_argument('--aux-cus', type=int, default=0, help='CUs set aside for the pack / histogram stream (multiple of 8; compute kernels are masked off them)')
    ap.add_argument('--fit-priority', type=int, default=0, help='stream priority of the per-batch (fit) streams: -1 high, 0 normal (default: with equal priorities the dispatcher alternates between the C(t) grid and the fits; 3 %% better than high-priority fits)')
    ap.add_argument('--main-priority', type=int, default=0, help='stream priority of the main (C(t)) stream')
    ap.add_argument('--dev-no-events', action='store_true', help='DEVELOPMENT: no per-kernel HIP events inside the timed region (kernels entries lose their in-pipeline durations)')
    ap.add_argument('--plane-buffers', type=int, default=3, help='plane buffers the pack kernel cycles through (2 = the pack of batch k+1 waits for the C(t) launch of batch k-1)')
    ap.add_argument('--fit-waves', type=int, default=0, help='wavefronts per residue in the model-order search (0 = library default)')
    ap.add_argument('--fit-lds', type=int, default=-1, help='1/0: keep residue data in LDS during the fits (-1 = library default)')
    ap.add_argument('--dev-skip-fits', action='store_true', help='DEVELOPMENT ONLY (invalid as a benchmark): leave the fits and the relaxation kernel out, to see the floor the C(t) side alone sets')
    ap.add_argument('--hist-on-main', action='store_true', help='keep the histogram kernel in line with C(t) (only the pack runs beside it)')
    ap.add_argument('--ct-fft', type=int, default=-1, help='kernel 1: 4 = float32 transforms for every 1024 < F + L <= 8192, 3 = float32 real-input FFT (k_ct_rfft32) for 4096 < F + L <= 8192, 2 = float64 real-input FFT (k_ct_rfft), 1 = complex float64 FFT (k_ct_fft), 0 = direct (k_ct_palmer); -1 = library default (3)')
    ap.add_argument('--ct-wg-per-cu', type=int, default=0, help='DEVELOPMENT: cap of k_ct_rfft32 workgroups per CU (library option ct_wg_per_cu; 0 = as many as fit)')
    ap.add_argument('--ct-traceless', type=int, default=0, help='1: k_ct_rfft<12> with five transforms (traceless components; library option ct_traceless, default off)')
    ap.add_argument('--group', type=int, default=32, help='batches whose fits / relaxation run as ONE merged launch behind their C(t) kernels (GroupedPipeline; a run of K steps uses groups of min(K, group)); 1 = every batch launches its own fits (DevicePipeline, --depth of them in flight)')
    ap.add_argument('--pack-cus', type=int, default=128, help='grouped schedule: CUs the pack stream is confined to (0 = all; the compute streams always have the whole chip)')
    ap.add_argument('--psum-buffers', type=int, default=3, help='grouped schedule: raw-sum buffers the C(t) launches rotate through')
    ap.add_argument('--late-hist', type=int, default=0, help='grouped schedule: 1 = the histograms of a group run in the tail of its merged fit launch (released by the signal the launch\'s last workgroup writes; the planes of group + 3 batches stay alive), 0 = beside the C(t) kernels')
    ap.add_argument('--no-group-overlap', action='store_true', help='grouped schedule: the next group\'s C(t) kernels wait for the merged fit launch (strict phases)')
    ap.add_argument('--no-permute', action='store_true', help='grouped schedule: dispatch the merged launch in natural residue order')
    ap.add_argument('--dispatch', type=str, default='history', choices=['history', 'random'],
                    help='grouped schedule, order of the merged launch\'s residues: history = longest first by the evaluation counts of the last collected batch (a prediction; exact here because the benchmark repeats one shard -- the random-order figure is reported beside the headline), random = fixed pseudo-random order')
    ap.add_argument('--depth', type=int, default=5, help='batches in flight: the straggler tail of the last fit order of batch k overlaps batches k+1 .. k+depth-1 (1 = strictly serial steps)')
    ap.add_argument('--shard', type=str, default=None, help='r/N: time the vector range rank r of N would own under --scaling strong, alone on this GPU and without a process group (per-rank figures behind the scaling prediction of DESIGN.md section 7)')
    ap.add_argument('--no-h2d-stream', action='store_true', help='skip the upload-inclusive figure (value_with_h2d: every step\'s shard from pinned host memory)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cli-wall', action='store_true', help='skip the wall-clock run of the drop-in CLI chain (run-all.bash Step 3 + Step 4) on the same workload')
    ap.add_argument('--no-kernel-profile', action='store_true', help='skip the per-kernel alone / saturated timings after the timed region (the `kernels` entries then only carry in-pipeline durations)')
    ap.add_argument('--backend', type=str, default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)')
    ap.add_argument('--all-ranks-on-device0', action='store_true', help='rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--cpu-sample-vectors', type=int, default=6, help='vectors of the 1-thread reference-equivalent CPU sample (6: about 12 s of CPU work, run BEFORE the GPU is initialised)')
    ap.add_argument('--cpu-allcore-vectors', type=int, default=64, help='vectors of the all-core / CPU-FFT samples (>= 64: not cache-resident)')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the reference's algorithm for the whole path on a bounded sample
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(vecs_host, s, cfg, nv1, nvall):
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sr_oracle as o
    from spinrelax_amd import synth
    R, F, N = s['R'], s['F'], s['N']
    v4 = vecs_host[:N, :nv1].reshape(R, F, nv1, 3)
    triples = o.exact_triples(R, F, nv1)
    stages = {}
    t0 = time.time()
    Ct, dCt = o.calculate_Ct_Palmer(v4, dtype=np.float32)          # same numpy calls as the reference, float32, 1 thread
    stages['ct_s'] = time.time() - t0
    aniso = cfg >= 3
    hist = edges = None
    if aniso:
        t0 = time.time()
        rot = o.rotate_vector_simd(vecs_host[:N, :nv1], np.array(synth.Q_EXT))
        hist, edges = o.lambert_histogram(rot)
        del rot
        stages['rotate_hist_s'] = time.time() - t0
    t0 = time.time()
    t = o.calculate_dt(s['dt'], s['tau_memory'])
    fits = []
    for i in range(nv1):
        best, _ = o.optimised_curve_fitting(t, Ct[:, i].astype(np.float64), dCt[:, i].astype(np.float64))
        fits.append(best)
    stages['fit_s'] = time.time() - t0
    t0 = time.time()
    idx = [i for i, f in enumerate(fits) if f is not None]
    if idx:
        B0 = o.B0_from_Hz(synth.FIELD_MHZ * 1e6)
        S2 = [synth.ZETA * fits[i]['S2'] for i in idx]
        C = [synth.ZETA * np.asarray(fits[i]['C']) for i in idx]
        tau = [np.asarray(fits[i]['tau']) for i in idx]
        if aniso:
            Dpar, Dperp = o.symmtop_from_iso(synth.DISO, synth.DANI)
            bv, w = o.convert_LambertCylindricalHist_to_vecs(hist, edges)
            o.obtain_R1R2NOErho('rigid_symmtop', (Dpar, Dperp), B0, S2, C, tau, vecXH=[bv[i] for i in idx], weights=[w[i] for i in idx])
        else:
            o.obtain_R1R2NOErho('rigid_sphere', synth.DISO, B0, S2, C, tau)
    stages['relax_s'] = time.time() - t0
    total = sum(stages.values())
    out = dict(value=triples / total, unit='triples/s', cores=1, kind='port',
               sample='whole path (C(t) float32 per-lag einsum + %sscipy curve_fit order search + J(w)/R1/R2/NOE) on %d of the '
                      'vectors, all %d chunks x %d frames (%.3g exact triples), numpy/scipy as the reference calls them, 1 thread, %.1f s'
                      % ('rotation + histogram + ' if aniso else '', nv1, R, F, triples, total),
               stages_s={k: round(v, 3) for k, v in stages.items()},
               ct_only_triples_per_s=triples / stages['ct_s'])
    # like-for-like partner of k_ct_fft: the same Wiener-Khinchin formulation on the CPU (numpy FFT, 1 thread)
    nva = min(nvall, vecs_host.shape[1])
    v4a = vecs_host[:N, :nva].reshape(R, F, nva, 3)
    tra = o.exact_triples(R, F, nva)
    try:
        nf = min(nva, 16)
        t0 = time.time()
        o.calculate_Ct_fft(v4a[:, :, :nf])
        dtf = time.time() - t0
        out['cpu_fft_formulation'] = dict(value=o.exact_triples(R, F, nf) / dtf, unit='triples/s', cores=1,
                                          kind='port (float64 FFT autocorrelations, numpy.fft; C(t) stage only)',
                                          sample='%d vectors, %.2f s' % (nf, dtf))
    except Exception as exc:
        out['cpu_fft_formulation'] = dict(error=str(exc))
    # the reference's per-lag streaming algorithm as a multi-threaded C loop (oracle/ct_palmer_oracle.c), all host cores,
    # on a sample that does not fit the caches (>= 64 vectors: 75 MB)
    try:
        import ctypes
        import subprocess
        so = os.path.join(ROOT, 'oracle', 'libsr_oracle.so')
        if not os.path.isfile(so):
            subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libsr_oracle.so'],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        v4c = np.ascontiguousarray(v4a, dtype=np.float32)
        L = F // 2
        Cc = np.empty((L, nva), dtype=np.float32)
        dCc = np.empty_like(Cc)
        t0 = time.time()
        lib.sr_oracle_ct_palmer_f32_stream(v4c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(R), ctypes.c_int64(F),
                                           ctypes.c_int64(nva), Cc.ctypes.data_as(ctypes.c_void_p),
                                           dCc.ctypes.data_as(ctypes.c_void_p))
        dt2 = time.time() - t0
        out['all_cores'] = dict(value=tra / dt2, unit='triples/s', cores=os.cpu_count(),
                                kind='port (C + OpenMP, same per-lag streaming algorithm; C(t) stage only)',
                                sample='%d vectors (%.0f MB, not cache-resident), %.2f s' % (nva, v4c.nbytes / 1e6, dt2))
    except Exception as exc:                                 # the extra figure is optional
        out['all_cores'] = dict(error=str(exc))
    return out


def cli_wall(vecs_host, s, cfg, cpu):
    """Wall time of the PRODUCT: the drop-in scripts of run-all.bash Step 3 + Step 4 (run-all.bash:476-516) as subprocesses on
    the same synthetic trajectory, from a vector file on disk to the R1/R2/NOE tables -- process start-up, file reading,
    the one strided upload, the kernels, the reference's text writers and all.  Beside it the reference algorithm's time for
    the same chain, extrapolated from the cpu_baseline sample (1 thread, as the reference runs)."""
    import shutil
    import subprocess
    import tempfile
    from spinrelax_amd import synth
    tmp = tempfile.mkdtemp(prefix='sr_cli_')
    scr = os.path.join(ROOT, 'scripts')
    out = {}
    try:
        t0 = time.time()
        fn = os.path.join(tmp, 'vecs.npy')
        np.save(fn, vecs_host)
        out['write_input_s'] = time.time() - t0
        pref = os.path.join(tmp, 'rotdif')
        steps = [('calculate-Ct-from-traj', ['calculate-Ct-from-traj.py', '-s', 'reference.pdb', '-f', fn, '--dt', str(s['dt']), '--tau', str(s['tau_memory']),
                                             '-o', pref, '--vecHist', '--binary', '--vecAvg', '--S2', '--Ct'] +
                  (['--vecRot', ' '.join('%.6f' % x for x in synth.Q_EXT)] if cfg >= 3 else [])),
                 ('calculate-fitted-Ct', ['calculate-fitted-Ct.py', '-f', pref + '_Ctint.dat', '-o', pref]),
                 ('calculate-relaxations-from-Ct', ['calculate-relaxations-from-Ct.py', '-f', pref + '_fittedCt.dat', '-o', pref + '-600', '-F', '%ge6' % synth.FIELD_MHZ,
                                                    '--tu', 'ps', '--zeta', str(synth.ZETA)] +
                  (['--distfn', pref + '_vecHistogram.npz', '-D', '%g %g' % (synth.DISO, synth.DANI)] if cfg >= 3 else ['-D', '%g' % synth.DISO]))]
        total = 0.0
        for name, cmd in steps:
            t0 = time.time()
            p = subprocess.run([sys.executable, os.path.join(scr, cmd[0])] + cmd[1:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
            dt = time.time() - t0
            if p.returncode != 0:
                out['error'] = '%s failed: %s' % (name, p.stdout.decode()[-500:])
                return out
            out[name + '_s'] = round(dt, 3)
            total += dt
        out['total_s'] = round(total, 3)
        out['files'] = sorted(os.path.basename(f) for f in os.listdir(tmp) if f.startswith('rotdif'))
        out['note'] = ('three python processes (interpreter + library start-up each), input read from a %.0f MB .npy, ONE strided upload of the vectors, '
                       'outputs through the reference-format text writers' % (vecs_host.nbytes / 1e6))
        if cpu and cpu.get('value'):
            triples = synth.exact_triples(s['R'], s['F'], vecs_host.shape[1])
            out['cpu_reference_chain_s_extrapolated'] = round(triples / cpu['value'], 1)
            out['cpu_note'] = 'exact triples of the workload / the 1-thread whole-path rate of cpu_baseline (measured on a sample of the same vectors)'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


class PinnedFeed:
    """Feed of spinrelax_amd.pipeline.GroupedPipeline in which EVERY batch's shard arrives from page-locked host memory: the product's
    resident-vector path (ResidentVectors.append_pinned = sr_vectors_append_f32, csrc/sr_vectors.hip) on a copy stream, two device
    buffers -- batch k + 1 is in flight over PCIe while the kernels of batch k run.  The reference's unit of work starts at a
    trajectory on the host (calculate-Ct-from-traj.py:458-470); this is that stream's rate."""

    class _Src:
        def __init__(self, ptr, shape):
            self._ptr, self.shape = ptr, shape

        def data_ptr(self):
            return self._ptr

    def __init__(self, torch, ctx, dev, host_addr, frames, V, nbuf=2):
        self.torch, self.ctx, self.addr, self.frames, self.V, self.nbuf = torch, ctx, host_addr, frames, V, nbuf
        self.copy = torch.cuda.Stream(device=dev)
        self.rv = [ctx.vectors(V, capacity=frames) for _ in range(nbuf)]
        self.free = [None] * nbuf
        self.base = 0                    # batch number of the run's batch 0 (a run numbers its batches from 0)
        self.issued = -1
        self.last = -1

    def begin(self, nb):
        """a run of nb batches follows"""
        self.base = self.issued + 1
        self.last = self.base + nb - 1

    def _issue(self, n):
        b = n % self.nbuf
        with self.torch.cuda.stream(self.copy):
            if self.free[b] is not None:
                self.copy.wait_event(self.free[b])           # the pack that read this buffer last has run
            self.ctx.set_stream(self.copy.cuda_stream)
            self.rv[b].truncate(0)
            self.rv[b].append_pinned(self.addr, self.frames, self.V)
        self.issued = n

    def acquire(self, k, stream):
        n = self.base + k
        while self.issued < min(n + self.nbuf - 1, self.last):    # keep the copy engine one batch ahead
            self._issue(self.issued + 1)
        self.ctx.set_stream(stream.cuda_stream)
        ptr = self.rv[n % self.nbuf].device_ptr()                # makes `stream` wait for the batch's frames
        return PinnedFeed._Src(ptr, (self.frames, self.V, 3))

    def release(self, k, stream):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        self.free[(self.base + k) % self.nbuf] = ev

    def close(self):
        for r in self.rv:
            r.close()
        self.rv = []


def use_rfft_bench(args, s):
    return args.ct_fft in (-1, 2, 3) and 4096 < s['F'] + s['L'] <= 8192


def use_rfft32_bench(args, s):
    return (args.ct_fft in (-1, 3) and 4096 < s['F'] + s['L'] <= 8192) or (args.ct_fft == 4 and 1024 < s['F'] + s['L'] <= 8192)


def fft32_exec_flop(s, V):
    """float32 work per launch of k_ct_rfft32 by the textbook count (the committed PMC pass replaces it when it has the kernel):
    5 forward + 1 back complex transforms of H = M/2 points (5 H log2 H flop each), two twiddle passes (6 flop per point), the
    real-signal spectrum step (24 flop per frequency) for five signals, and per frame the signals (about 14 flop) twice (prologue
    means, epilogue e[j])."""
    need = s['F'] + s['L']
    M = 2048 if need <= 2048 else (4096 if need <= 4096 else (6144 if need <= 6144 else 8192))
    H = M // 2
    return M, s['R'] * V * (6 * (5 * H * np.log2(H) + 12 * H) + 5 * 24 * H + 60 * s['F'])


def fft_exec_flop(s, V, real_input=False, traceless=False):
    """executed float64 work per launch (formula; the committed PMC pass replaces it when it has the kernel).
    k_ct_fft: 4 complex M-point transforms (5 M log2 M flop each), the power spectra of 3 packed pairs (12 flop per
    frequency each) and the 6 products per frame, per (chunk, vector).  k_ct_rfft: 7 complex transforms of H = M/2 points,
    two twiddle passes (6 flop per point) per transform and the real-signal spectrum step (24 flop per frequency) for the
    six signals (five with the option ct_traceless of k_ct_rfft<12>: 6 transforms in all)."""
    need = s['F'] + s['L']
    M = 2048 if need <= 2048 else (4096 if need <= 4096 else (6144 if need <= 6144 else 8192))
    if real_input:
        H = M // 2
        nf = 5 if (traceless and M == 6144) else 6
        return M, s['R'] * V * ((nf + 1) * (5 * H * np.log2(H) + 12 * H) + nf * 24 * H + 6 * s['F'])
    return M, s['R'] * V * (4 * 5 * M * np.log2(M) + 3 * 12 * M + 6 * s['F'])


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from spinrelax_amd import synth
    from spinrelax_amd.hip import Context
    from spinrelax_amd.pipeline import DevicePipeline, GroupedPipeline

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))

    cfg = {'cfg2': 2, 'cfg3': 3, 'cfg4': 4}[args.workload]
    s = synth.config_shapes(cfg)
    aniso = synth.DANI if cfg >= 3 else None
    q = synth.Q_EXT if cfg >= 3 else None
    # ---- which vectors this rank owns ----
    strong = args.scaling == 'strong'
    if strong:
        Vtot = args.vectors or s['V']
        if Vtot % world:
            sys.exit('bench.py: --scaling strong needs the %d vectors of the workload to divide by the %d ranks (equal all-gather pieces)' % (Vtot, world))
        from spinrelax_amd.dist import shard_range
        v0, V = shard_range(Vtot, rank, world)
        if args.shard:
            sr, sn = (int(x) for x in args.shard.split('/'))
            v0, V = shard_range(Vtot, sr, sn)
            Vtot = V                       # the figures of this run are the shard's own
    else:
        V = args.vectors or s['V']
        Vtot, v0 = V * world, rank * V
    # ---- synthetic shard of this rank: generated on the host BEFORE the GPU is initialised (worker pool forks) ----
    t0 = time.time()
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'], v0=v0)
    gen_s = time.time() - t0
    # ---- CPU baseline (rank 0, N = 1), also before the GPU is touched: it needs nothing from the device, and run first it
    # does not sit between the GPU phases of the run ----
    cpu_res = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(vecs_host, s, cfg, min(args.cpu_sample_vectors, V), min(args.cpu_allcore_vectors, V))

    if args.all_ranks_on_device0:
        local = 0
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=args.backend)
    if args.gpus != world and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # The CPU baseline leaves a few hundred thousand Python objects behind (scipy, the oracle): a full collection of the cyclic
    # garbage collector then takes ~60 ms and, triggered by the event objects of a long run, lands inside it (steady 1.55-1.83
    # instead of 1.30 ms per step).  Collect now and move what exists to the permanent generation; the collector stays on.
    import gc
    gc.collect()
    gc.freeze()

    vecs = torch.from_numpy(vecs_host).to(dev)           # resident in HBM before the timed region
    ctx = Context(local)
    if args.fit_waves:
        ctx.set_option('fit_waves', args.fit_waves)
    if args.fit_lds >= 0:
        ctx.set_option('fit_lds', args.fit_lds)
    if args.ct_fft >= 0:
        ctx.set_option('ct_fft', args.ct_fft)
    if args.ct_traceless:
        ctx.set_option('ct_traceless', 1)
    if args.ct_wg_per_cu:
        ctx.set_option('ct_wg_per_cu', args.ct_wg_per_cu)
    triples = synth.exact_triples(s['R'], s['F'], V)
    pkw = dict(q_rot=q, Diso=synth.DISO, aniso=aniso, field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA)
    grouped = args.group > 1 and args.depth > 1 and not args.reserve_cus and not args.aux_cus and not args.hist_on_main
    if grouped:
        pipe = GroupedPipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], group=args.group, overlap=not args.no_group_overlap, psum_buffers=args.psum_buffers, late_hist=bool(args.late_hist), pack_cus=args.pack_cus,
                               stream=torch.cuda.Stream(device=dev, priority=args.main_priority),
                               **({} if args.late_hist else {'plane_buffers': args.plane_buffers}), **pkw)
        pipe.permute = not args.no_permute
        pipe.dispatch = args.dispatch
        pipe.dev_skip_fits = bool(args.dev_skip_fits)
    else:
        pipe = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], depth=args.depth,
                              stream=torch.cuda.Stream(device=dev, priority=args.main_priority), reserve_cus=args.reserve_cus, aux_cus=args.aux_cus,
                              fit_priority=args.fit_priority, plane_buffers=args.plane_buffers,
                              fits_on_reserved_only=bool(args.fits_on_reserved_only), hist_on_aux=not args.hist_on_main, **pkw)
        if args.dev_skip_fits:
            pipe.stage_fit = lambda s=None: None
            pipe.stage_relax = lambda s=None: None
    stream = pipe.main
    ctx.set_stream(stream.cuda_stream)

    with torch.cuda.stream(stream):

        # Result exchange: every rank ends up with C(t), dC(t), the histogram and the R1/R2/NOE
        # table of all vectors.  The all-gathers of a finished batch run on their own stream, beside the C(t) launches
        # the host has already queued for the following batches; the main stream only waits for them before it
        # reuses that batch's buffers (depth batches later).
        gstream = torch.cuda.Stream(device=dev) if world > 1 else None
        gbuf = {}

        def gather_part(kind, bv, ev):
            """grouped schedule: the bulky per-batch arrays travel as soon as they are final (C(t), dC(t) behind the batch's chunk
            statistics; the histogram behind its kernel), beside the rest of the group's work -- not in one piece at its end"""
            if world == 1:
                return
            with torch.cuda.stream(gstream):
                gstream.wait_event(ev)
                for name in (('Ct', 'dCt') if kind == 'ct' else ('hist',)):
                    tns = getattr(bv, name)
                    key = ('part', name, tuple(tns.shape))
                    if key not in gbuf:
                        gbuf[key] = [torch.empty_like(tns) for _ in range(world)]
                    dist.all_gather(gbuf[key], tns)

        def gather_results(slot):
            """queued right behind a batch (pipeline.run(on_enqueued=...)): waits for the batch on the device, never on the host"""
            if world == 1:
                return None
            with torch.cuda.stream(gstream):
                gstream.wait_event(slot.done)
                for name in (('relax',) if grouped else ('Ct', 'dCt', 'hist', 'relax')):   # grouped: the rest went through gather_part
                    tns = getattr(slot, name)
                    key = (id(slot), name, tuple(tns.shape))
                    if key not in gbuf:
                        gbuf[key] = [torch.empty_like(tns) for _ in range(world)]
                    dist.all_gather(gbuf[key], tns)
                ev = torch.cuda.Event()
                ev.record(gstream)
            return ev

        def run_batches(nb, events=None):
            if grouped:
                pipe.run(vecs, nb, events, on_enqueued=gather_results, on_part=gather_part)
            else:
                pipe.run(vecs, nb, events, on_enqueued=gather_results)
            torch.cuda.synchronize()

        pipe.prime(vecs)                 # set-up (code objects, first touch of every in-flight slot), not a warm-up step
        if world > 1:                    # set-up: communicator and gather buffers exist before anything is timed
            for sl in pipe.slots:
                gather_results(sl)
            torch.cuda.synchronize()
            dist.barrier()
        # clock spin-up (untimed, same work as the timed steps): run until --spinup-s seconds have passed
        spin_steps, t_spin = 0, time.perf_counter()
        spin_chunk = args.steps if grouped else 10     # grouped: the spin-up runs groups of the size the timed region uses
        while time.perf_counter() - t_spin < args.spinup_s:
            run_batches(spin_chunk)
            spin_steps += spin_chunk
        spin_s = time.perf_counter() - t_spin
        run_batches(args.warmup)
        nfev0 = pipe.nfev_total
        samples = []
        events = None
        for rep in range(max(1, args.repeats)):
            events = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(args.steps)]
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_batches(args.steps, None if args.dev_no_events else events)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t0)
        nfev_timed = pipe.nfev_total - nfev0
        # the same timed region with the prediction switched off (pseudo-random dispatch order of the merged launch), once: what a
        # stream of unrelated trajectories would see
        random_dispatch = None
        if grouped and args.dispatch == 'history' and not args.no_permute:
            pipe.dispatch = 'random'
            run_batches(args.steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_batches(args.steps)
            random_dispatch = (time.perf_counter() - t0) / args.steps
            pipe.dispatch = 'history'
            run_batches(args.steps)                   # (re-establishes the history for what follows)
        # steady state: one long run, fill and drain amortised (reported beside the headline, never as it)
        steady = None
        if args.steady_steps > 0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_batches(args.steady_steps)
            steady = (time.perf_counter() - t0) / args.steady_steps

    # ---- the same K steps with every step's shard coming from page-locked HOST memory (upload-inclusive stream) ----
    with_h2d = None
    if grouped and not args.no_h2d_stream:
        shard_bytes = int(vecs_host.nbytes)
        haddr = ctx.host_alloc(shard_bytes)
        import ctypes
        hview = np.frombuffer((ctypes.c_char * shard_bytes).from_address(haddr), dtype=np.float32).reshape(vecs_host.shape)
        np.copyto(hview, vecs_host)
        feed = PinnedFeed(torch, ctx, dev, haddr, s['frames'], V)
        with torch.cuda.stream(stream):
            for timed_run in (False, True):
                feed.begin(args.steps)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pipe.run(feed, args.steps, None, on_enqueued=gather_results, on_part=gather_part)
                torch.cuda.synchronize()
                if timed_run:
                    with_h2d = (time.perf_counter() - t0) / args.steps
        ctx.set_stream(stream.cuda_stream)
        feed.close()
        del hview
        ctx.host_free(haddr)

    if world > 1:                  # every repeat: the slowest rank's time
        tmax = torch.tensor(samples + [steady or 0.0, random_dispatch or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        samples = [float(x) for x in tmax[:-2].tolist()]
        steady = float(tmax[-2].item()) if steady else None
        random_dispatch = float(tmax[-1].item()) if random_dispatch else None
    elapsed = float(np.median(samples))
    if args.dev_no_events:
        ct_ms = hist_ms = fit_ms = float('nan')
    else:
        ct_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        def _pairs(i, j):
            out = []
            for e in events:
                try:
                    out.append(e[i].elapsed_time(e[j]))
                except (ValueError, RuntimeError):
                    pass
            return out
        hp = _pairs(2, 3) if q is not None else []
        hist_ms = float(np.mean(hp)) if hp else None
        fp = _pairs(4, 5)                # grouped schedule: one merged launch per group, recorded on its first batch's entry
        fit_ms = None if (args.dev_skip_fits or not fp) else float(np.mean(fp))
    # ---- one more (untimed) batch whose results are checksummed: C(t), dC(t), histogram and the R1/R2/NOE table of ALL the
    # workload's vectors (all-gathered for N > 1) -- identical for every rank count and schedule (tests compare them) ----
    last = {}

    def keep(b):
        last['Ct'], last['dCt'], last['hist'] = b.Ct.clone(), b.dCt.clone(), b.hist.clone()
        last['relax'] = torch.from_numpy(np.ascontiguousarray(b.result['relax'])).to(dev)
    with torch.cuda.stream(stream):
        pipe.run(vecs, 1, None, keep, None)
    torch.cuda.synchronize()
    checksums = None
    if last:
        import hashlib
        whole = {}
        for name, tns in last.items():
            tns = tns.contiguous()
            if world > 1:
                parts = [torch.empty_like(tns) for _ in range(world)]
                dist.all_gather(parts, tns)
                axis = {'Ct': 1, 'dCt': 1, 'hist': 0, 'relax': 1}[name]            # the vector / residue axis
                tns = torch.cat(parts, dim=axis)
            whole[name] = tns.cpu().numpy()
        if rank == 0:
            checksums = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] for k, v in whole.items()}
            checksums['shapes'] = {k: list(v.shape) for k, v in whole.items()}
    best = pipe.fit_best
    nfev_by_order = {str(k): int(np.sum(v)) for k, v in pipe.nfev_last.items()}
    nfits_by_order = {str(k): int(np.size(v)) for k, v in pipe.nfev_last.items()}
    nfev_step = nfev_timed / max(1, args.steps * max(1, args.repeats))
    depth_used, reserve_used = pipe.depth, pipe.reserve_cus
    pack_cus_used = getattr(pipe, 'pack_cus', 0) or 256
    group_used = min(args.group, args.steps) if grouped else 1
    late_hist_used, late_hist_note = bool(getattr(pipe, 'late_hist', False)), getattr(pipe, 'late_hist_note', None)
    listDoG = pipe.listDoG
    del events, gbuf
    pipe.close()

    # ---- after the timed region: one batch at a time on the whole chip -- latency, every kernel alone, the fit saturated ----
    alone, latency = {}, None
    if rank == 0 and not args.no_kernel_profile and not args.dev_skip_fits:
        p1 = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], depth=1, stream=torch.cuda.Stream(device=dev), **pkw)
        st1 = p1.main
        p1.step(vecs)
        lat = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            p1.step(vecs)
            lat.append((time.perf_counter() - t1) * 1e3)
        latency = dict(min=min(lat), mean=float(np.mean(lat)), note='one batch alone on the whole chip, vectors in HBM -> R1/R2/NOE table on the host')
        s0 = p1.slots[0]
        ctx.set_stream(st1.cuda_stream)

        def timed(fn, reps=3):
            out = []
            with torch.cuda.stream(st1):
                for _ in range(reps):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st1)
                    fn()
                    b.record(st1)
                    b.synchronize()
                    out.append(a.elapsed_time(b))
            return float(min(out))

        alone['pack'] = timed(lambda: p1.stage_pack(vecs))
        alone['ct'] = timed(lambda: p1.stage_ct(s0, finalize=False))
        alone['ct_finalize'] = timed(lambda: p1.stage_ct_finalize(s0))
        if q is not None:
            alone['hist'] = timed(lambda: p1.stage_hist(s0))
        alone['fit'] = timed(lambda: p1.stage_fit(s0), reps=2)
        alone['relax'] = timed(lambda: p1.stage_relax(s0))
        # the direct formulation of kernel 1 (the north-star's named kernel) as a second line
        if 1024 < s['F'] + s['L'] <= 8192 and args.ct_fft != 0:
            ctx.set_option('ct_fft', 0)
            alone['ct_direct'] = timed(lambda: p1.stage_ct(s0, finalize=False), reps=2)
            if args.ct_fft in (-1, 2, 3, 4) and (s['F'] + s['L'] > 4096 or args.ct_fft == 4):
                ctx.set_option('ct_fft', 1)                       # and the complex-FFT formulation it replaced, for reference
                alone['ct_complex_fft'] = timed(lambda: p1.stage_ct(s0, finalize=False), reps=2)
                if use_rfft32_bench(args, s):
                    ctx.set_option('ct_fft', 2)                   # the float64 transform kernel of this chunk length (rounds 2-4's production kernel)
                    alone['ct_rfft_f64'] = timed(lambda: p1.stage_ct(s0, finalize=False), reps=2)
            ctx.set_option('ct_fft', 3 if args.ct_fft < 0 else args.ct_fft)
            with torch.cuda.stream(st1):
                p1.stage_ct(s0)                 # restore the slot's sums / C(t) from the production kernel
        # the other real-input FFT variant (M = 8192: 4096 < F <= 5461) on the same planes, at two chunk lengths: its cost per
        # frame against the benchmark variant's
        if use_rfft_bench(args, s) and s['frames'] >= 5461:
            per_frame = alone['ct'] / (s['R'] * s['F'])
            for F2 in (5000, 5461):
                R2 = s['frames'] // F2
                ps2 = torch.empty((V * R2 * ctx.psum_stride(F2),), device=dev, dtype=torch.float64)
                alone['ct_F%d' % F2] = timed(lambda: ctx.ct_sums_dev(p1.soa.data_ptr(), p1.Npad, R2, F2, V, ps2.data_ptr()), reps=2)
                alone['ct_F%d_per_frame_vs_cfg3' % F2] = alone['ct_F%d' % F2] / (R2 * F2) / per_frame
                del ps2
        # The model-order search with the chip saturated: 32 batches' residues in ONE launch, the expensive residues first
        # (sorted by the evaluations they needed in the batch above).  A lone launch lasts as long as its slowest residue
        # (~10 ms for one 286-evaluation fit); with several batches in flight that tail overlaps the next batches' work,
        # and what a batch costs the chip is (sum of the residues' solve times) / (resident workgroups) -- which is what
        # this launch measures, because longest-first dispatch leaves no tail.
        K = int(os.environ.get('SR_BENCH_SATK', 32))
        f64 = dict(device=dev, dtype=torch.float64)
        i32 = dict(device=dev, dtype=torch.int32)
        nO, Pmax, Kmax = len(listDoG), max(listDoG), max(listDoG) // 2
        cost = s0.result['nfev'].sum(axis=0)
        order = torch.from_numpy(np.argsort(-cost, kind='stable').copy()).to(dev)
        torch.cuda.synchronize()
        tK = p1.t_dev.repeat(K, 1)
        yK, dK = s0.CtT[order].repeat_interleave(K, dim=0), s0.dCtT[order].repeat_interleave(K, dim=0)
        oK = dict(popt=torch.empty((nO, K * V, Pmax), **f64), dP=torch.empty((nO, K * V, Pmax), **f64), chisq=torch.empty((nO, K * V), **f64),
                  status=torch.empty((nO, K * V), **i32), nfev=torch.empty((nO, K * V), **i32), best=torch.empty((K * V,), **i32),
                  S2=torch.empty((K * V,), **f64), C=torch.empty((K * V, Kmax), **f64), tau=torch.empty((K * V, Kmax), **f64),
                  chi=torch.empty((K * V,), **f64), Kc=torch.empty((K * V,), **i32), work=torch.empty((K * V, s['L']), **f64))
        torch.cuda.synchronize()

        def fitK():
            ctx.order_search_dev(tK.data_ptr(), yK.data_ptr(), dK.data_ptr(), K * V, s['L'], listDoG, p1.tau_guess.data_ptr(), 1,
                                 p1.tau_max, p1.chi_thr, oK['popt'].data_ptr(), oK['dP'].data_ptr(), oK['chisq'].data_ptr(),
                                 oK['status'].data_ptr(), oK['nfev'].data_ptr(), oK['best'].data_ptr(), oK['S2'].data_ptr(),
                                 oK['C'].data_ptr(), oK['tau'].data_ptr(), oK['chi'].data_ptr(), oK['Kc'].data_ptr(),
                                 work_ptr=oK['work'].data_ptr())
        alone['fit_saturated_per_batch'] = timed(fitK, reps=2) / K
        torch.cuda.synchronize()
        del tK, yK, dK, oK
        p1.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        triples_total = synth.exact_triples(s['R'], s['F'], Vtot)
        value = triples_total / (elapsed / args.steps)
        use_fft = (args.ct_fft != 0) and 1024 < s['F'] + s['L'] <= 8192
        use_rfft = use_fft and args.ct_fft in (-1, 2, 3, 4) and s['F'] + s['L'] > 4096
        use_f32 = use_fft and use_rfft32_bench(args, s)
        kname = 'k_ct_rfft32' if use_f32 else ('k_ct_rfft' if use_rfft else ('k_ct_fft' if use_fft else 'k_ct_palmer'))
        prof, prof_src = committed_profile() if cfg == 3 and V == 512 else ({}, None)
        stale, build_id = profile_staleness() if prof else (None, None)
        stale_txt = ' -- STALE: collected with library build %s, this run loaded build %s (re-run scripts/profile_round.sh)' % (PROFILE_BUILD_ID, build_id) if stale else ''
        N, R, L = s['N'], s['R'], s['L']
        kernels = {}

        def entry(name, bound, work, unit_scale, peak, unit, t_pipe, t_alone, t_sat=None, **extra):
            """work per launch (flop or bytes); achieved = work / duration; frac against `peak`."""
            e = dict(bound=bound, work_per_launch=float(work), peak=peak, unit=unit)
            if t_pipe:
                e['in_pipeline_ms'] = t_pipe
                e['achieved'] = work / (t_pipe * 1e-3) / unit_scale
                e['frac'] = e['achieved'] / peak
            if t_alone:
                e['alone_ms'] = t_alone
                e['achieved_alone'] = work / (t_alone * 1e-3) / unit_scale
                e['frac_alone'] = e['achieved_alone'] / peak
                if 'achieved' not in e:
                    e['achieved'], e['frac'] = e['achieved_alone'], e['frac_alone']
            sat = t_sat if t_sat else t_alone
            if sat:
                e['cu_ms_per_batch'] = sat * N_CU
            pe = prof_entry(prof, name)
            e['traffic'] = pe['hbm_bytes_corrected'] if pe else None
            e['traffic_note'] = ('HBM bytes per launch (PMC 2*FETCH_SIZE + WRITE_SIZE) from the committed profile %s -- '
                                 'not measured in this run%s' % (prof_src, stale_txt)) if pe else 'no committed PMC profile for this kernel / configuration'
            e.update(extra)
            kernels[name] = e
            return e

        if use_f32:
            M, xflop = fft32_exec_flop(s, V)
            xflop_formula = float(xflop)
            pe = prof_entry(prof, kname)
            xsrc = 'formula (bench.py:fft32_exec_flop)'
            if pe and pe.get('fp32_flop_per_launch'):
                xflop, xsrc = pe['fp32_flop_per_launch'], 'PMC float32 instruction counts of the committed profile %s%s' % (prof_src, stale_txt)
            entry(kname, 'valu-fp32 (packed float32 vector arithmetic; LDS-exchange- and latency-limited)', xflop, 1e12, PEAK_FP32_TFLOPS, 'TFLOP/s',
                  ct_ms, alone.get('ct'),
                  formulation='Wiener-Khinchin on real input in the reference\'s arithmetic type: 5 + 1 FLOAT32 complex transforms of %d points on the '
                              'mean-removed traceless components (packed v_pk_*_f32 complex arithmetic), the mean terms restored in float64 by one scan '
                              'per series; three workgroups per CU (spinrelax_amd/csrc/sr_ct32.hip)' % (M // 2),
                  work_source=xsrc, work_per_launch_textbook=xflop_formula, executed_over_textbook=float(xflop) / xflop_formula,
                  algorithmic_bytes=12 * N * V + 8 * R * L * V,
                  float64_transform_kernel_alone_ms=alone.get('ct_rfft_f64'),
                  accuracy='C(t) 4e-8 relative, dC(t) 3e-9 absolute against the float64 oracle on this workload (tests/test_gpu_pipeline.py); the reference itself computes C(t) in float32',
                  note='in the pipeline the kernel is confined to %d of %d CUs' % (N_CU - reserve_used, N_CU))
            if alone.get('ct_direct'):
                entry('k_ct_palmer', 'valu-fp32 (non-MFMA vector FMA; FP32 MFMA peak is the same)', 8.0 * triples, 1e12, PEAK_FP32_TFLOPS, 'TFLOP/s',
                      None, alone['ct_direct'], formulation='direct shifted products, 8 flop per (frame, vector, lag) triple (SURVEY 8(d)); second line, timed alone',
                      algorithmic_bytes=12 * N * V + 8 * R * L * V)
        elif use_fft:
            M, xflop = fft_exec_flop(s, V, use_rfft, bool(args.ct_traceless))
            xflop_formula = float(xflop)
            pe = prof_entry(prof, kname)
            xsrc = 'formula (bench.py:fft_exec_flop)'
            if pe and pe.get('fp64_flop_per_launch'):
                xflop, xsrc = pe['fp64_flop_per_launch'], 'PMC float64 instruction counts of the committed profile %s%s' % (prof_src, stale_txt)
            form = ('Wiener-Khinchin on real input: %s float64 complex transforms of %d points (half the padded length), two workgroups per CU'
                    % ('5 (traceless components; the trace term from window sums) + 1' if (bool(args.ct_traceless) and M == 6144) else '6 + 1', M // 2)
                    if use_rfft else 'Wiener-Khinchin: 6 float64 autocorrelations by FFT, whole %d-point transform resident in LDS' % M)
            entry(kname, 'valu-fp64 (float64 vector FMA; latency- and LDS-exchange-limited)', xflop, 1e12, PEAK_FP64_TFLOPS, 'TFLOP/s',
                  ct_ms, alone.get('ct'), formulation=form, work_source=xsrc,
                  work_per_launch_textbook=xflop_formula, executed_over_textbook=float(xflop) / xflop_formula,
                  work_textbook_note='textbook operation count of the same transforms (5 N log2 N per complex N-point transform, 6 flop '
                                     'per twiddle, 24 per frequency of the spectrum step; bench.py:fft_exec_flop) -- an upper bound: the '
                                     'radix-12 / 16 register transforms have trivial twiddles and the zero blocks of the padded input are pruned, '
                                     'so the kernel executes less (work_per_launch, PMC)',
                  algorithmic_bytes=12 * N * V + 8 * R * L * V,
                  note='in the pipeline the kernel is confined to %d of %d CUs' % (N_CU - reserve_used, N_CU))
            if alone.get('ct_direct'):
                entry('k_ct_palmer', 'valu-fp32 (non-MFMA vector FMA; FP32 MFMA peak is the same)', 8.0 * triples, 1e12, PEAK_FP32_TFLOPS, 'TFLOP/s',
                      None, alone['ct_direct'], formulation='direct shifted products, 8 flop per (frame, vector, lag) triple (SURVEY 8(d)); second line, timed alone',
                      algorithmic_bytes=12 * N * V + 8 * R * L * V)
        else:
            entry('k_ct_palmer', 'valu-fp32 (non-MFMA vector FMA; FP32 MFMA peak is the same)', 8.0 * triples, 1e12, PEAK_FP32_TFLOPS, 'TFLOP/s',
                  ct_ms, alone.get('ct'), formulation='direct shifted products, 8 flop per triple', algorithmic_bytes=12 * N * V + 8 * R * L * V)
        if q is not None:
            entry('k_vechist', 'hbm', 12.0 * N * V + 8.0 * V * 2592, 1e9, PEAK_HBM_GBS, 'GB/s', hist_ms, alone.get('hist'),
                  formulation='one pass over the planes: rotation + Lambert histogram + mean vector + S2 sums; bytes = 12 N V + 8 V 2592 (SURVEY 8(d))')
        if alone.get('pack'):
            entry('k_pack_soa', 'hbm', 24.0 * s['frames'] * V, 1e9, PEAK_HBM_GBS, 'GB/s', None, alone['pack'],
                  formulation='frame-major -> per-vector planes: 12 B read + 12 B written per (frame, vector)')
        if fit_ms is not None:
            pe = prof_entry(prof, 'k_order_search')
            fflop = pe.get('fp64_flop_per_launch') if pe else None
            sat = alone.get('fit_saturated_per_batch')
            tref = sat or alone.get('fit') or fit_ms
            e = dict(bound='valu-fp64 issue + latency (one workgroup per residue, hundreds of dependent solver iterations)',
                     in_pipeline_ms=fit_ms, in_pipeline_batches_per_launch=group_used,
                     in_pipeline_ms_per_batch=fit_ms / group_used, alone_ms=alone.get('fit'), saturated_ms_per_batch=sat,
                     cu_ms_per_batch=sat * N_CU if sat else None, residues=V, evaluations_per_batch=nfev_step,
                     residues_per_s=V / (tref * 1e-3), evaluations_per_s=nfev_step / (tref * 1e-3),
                     peak=PEAK_FP64_TFLOPS, unit='TFLOP/s',
                     note='rates over the saturated duration (32 batches, longest residues first, in one launch / 32): the time the chip needs per batch when '
                          'it is full of fits; a lone launch lasts as long as its slowest residue (alone_ms)')
            if fflop:
                e['work_per_launch'] = float(fflop)
                e['achieved'] = fflop / (tref * 1e-3) / 1e12
                e['frac'] = e['achieved'] / PEAK_FP64_TFLOPS
                e['work_note'] = 'executed float64 flop per launch from the PMC pass of the committed profile %s (deterministic data)' % prof_src
                if fit_ms:
                    # the launch as it ran inside the timed region: merged over the group, residues in permuted order -- its
                    # duration includes the tail in which only the last-started expensive residues are still running
                    e['achieved_in_pipeline'] = fflop * group_used / (fit_ms * 1e-3) / 1e12
                    e['frac_in_pipeline'] = e['achieved_in_pipeline'] / PEAK_FP64_TFLOPS
                e['work_note'] += stale_txt
            else:
                e['achieved'] = e['frac'] = None
                e['work_note'] = 'no committed float64-instruction PMC pass: flop rate not stated'
            if fit_ms and sat:
                # merged launch = bulk (the chip full of fits: saturated time per batch x batches) + tail (the last-started
                # expensive residues finishing while nothing takes the freed slots; the group's histograms run there)
                bulk = sat * group_used
                e['in_pipeline_bulk_ms'] = bulk
                e['in_pipeline_tail_ms'] = max(0.0, fit_ms - bulk)
                e['in_pipeline_tail_frac'] = max(0.0, fit_ms - bulk) / fit_ms
            e['traffic'] = pe['hbm_bytes_corrected'] if pe else None
            e['traffic_note'] = 'from the committed profile %s -- not measured in this run%s' % (prof_src, stale_txt) if pe else None
            e['algorithmic_bytes'] = 24 * V * L + 1024 * V
            kernels['k_order_search'] = e
        # how close the two float64 kernels are to what their occupancy lets the vector pipe issue (both hold 2 waves per SIMD)
        probe = fp64_issue_probe()
        if probe and prof:
            ceil2 = probe['instr_per_us_per_simd_by_waves'].get('2')
            for kn, dur_ms in ((kname, alone.get('ct')), ('k_order_search', alone.get('fit_saturated_per_batch'))):
                pe = prof_entry(prof, kn)
                if not (pe and pe.get('fp64_wave_instructions') and dur_ms and ceil2 and kn in kernels):
                    continue
                n64 = float(sum(pe['fp64_wave_instructions'].values()))
                rate = n64 / 1024.0 / (dur_ms * 1e3)
                kernels[kn]['fp64_issue'] = {
                    'fp64_wave_instructions_per_launch': n64, 'duration_ms': dur_ms, 'waves_per_simd': 2,
                    'fp64_instr_per_us_per_simd': rate, 'two_wave_ceiling': ceil2, 'frac_of_two_wave_ceiling': rate / ceil2,
                    'note': 'float64 VALU wave-instructions (PMC) per SIMD and microsecond over the kernel alone / saturated, against what a '
                            'SIMD holding two waves issues in the probe (%s): the other VALU instructions of the kernel are not '
                            'counted, so the true figure is higher%s' % (probe['source'], stale_txt)}
        # the kernel that occupies most of the chip per batch
        off_step = {'k_ct_palmer'} if use_fft else set()
        ranked = sorted(((v.get('cu_ms_per_batch') or 0.0, k) for k, v in kernels.items() if k not in off_step), reverse=True)
        top = ranked[0][1] if ranked and ranked[0][0] > 0 else kname
        tk = kernels[top]
        roofline = {'kernel': top, 'bound': tk['bound'], 'achieved': tk.get('achieved'), 'peak': tk['peak'], 'unit': tk['unit'],
                    'frac': tk.get('frac'), 'traffic': tk.get('traffic'), 'traffic_note': tk.get('traffic_note'),
                    'selection': 'largest cu_ms_per_batch among the kernels of a step: ' + ', '.join('%s %.0f' % (k, c) for c, k in ranked),
                    'kernel_ms': tk.get('saturated_ms_per_batch') or tk.get('in_pipeline_ms') or tk.get('alone_ms'),
                    'direct_equivalent': {'note': 'for comparison only, NOT a roofline: the reference formulation (8 flop, 24 B streamed per triple) '
                                                  'credited to the duration of the kernel that computes C(t)',
                                          'kernel': kname, 'kernel_ms': ct_ms, 'TFLOPs_equiv': 8.0 * triples / (ct_ms * 1e-3) / 1e12,
                                          'streaming_equiv_GBps': 24.0 * triples / (ct_ms * 1e-3) / 1e9}}
        # what the whole chip does per step: executed float64 work of every kernel of a batch over the step time (the
        # per-kernel `frac` of an overlapped pipeline is diluted by sharing: two C(t) launches, the pack, the histogram and the
        # chunk statistics run at the same time -- per-step schedule: the fits of three batches as well --, so a launch's own
        # duration says little about the chip)
        prof_all, _ = committed_profile()
        step_flop = sum(float(v.get('fp64_flop_per_launch') or 0.0) * (2 if k.startswith('k_transpose') else 1) for k, v in prof_all.items()
                        if not k.startswith('k_fft_init'))
        roofline['traffic_stale'] = stale
        roofline['library_build_id'] = build_id
        roofline['profile_build_id'] = PROFILE_BUILD_ID if prof else None
        step_bytes = sum(float(v.get('hbm_bytes_corrected') or 0.0) * (2 if k.startswith('k_transpose') else 1) for k, v in prof_all.items()
                         if not k.startswith('k_fft_init')) if prof else 0.0
        if step_bytes > 0:
            compulsory = 12.0 * s['frames'] * V + 16.0 * L * V + 8.0 * V * 2592
            roofline['hbm_bytes_per_step'] = {'pmc': step_bytes, 'compulsory': compulsory, 'ratio': step_bytes / compulsory,
                                              'GBps': step_bytes / (ms_per_step * 1e-3) / 1e9,
                                              'note': 'sum over the kernels of a batch of the committed PMC pass (2*FETCH_SIZE + WRITE_SIZE: requests the L2s send to the '
                                                      'fabric, Infinity-Cache hits included) against vectors in once + C(t), dC(t), histogram out; not measured in this run. '
                                                      'The pack writes the planes once and they are read by kernel 1 and kernel 2 (1.84 GB of it); k_ct_rfft32 re-reads its '
                                                      'planes per pass and, at four workgroups per CU, 128 x 48 KB per XCD exceed its 4 MB L2: 0.9 GB of re-reads come from the '
                                                      'Infinity Cache (DESIGN.md section 4)%s' % stale_txt}
        roofline['fp64_issue_probe'] = probe
        roofline['frac_alone'] = tk.get('frac_alone')
        roofline['launches_of_this_kernel_in_flight'] = (tk.get('in_pipeline_ms') or 0.0) / ms_per_step if ms_per_step else None
        step_flop32 = sum(float(v.get('fp32_flop_per_launch') or 0.0) for k, v in prof_all.items() if not k.startswith('k_fft_init') and not k.startswith('k_ct32_init'))
        roofline['chip'] = None if not (cfg == 3 and V == 512 and step_flop > 0) else {
            'fp64_flop_per_step': step_flop, 'fp32_flop_per_step': step_flop32,
            'TFLOPs_fp64': step_flop / (ms_per_step * 1e-3) / 1e12, 'TFLOPs_fp32': step_flop32 / (ms_per_step * 1e-3) / 1e12,
            'frac': (step_flop / PEAK_FP64_TFLOPS + step_flop32 / PEAK_FP32_TFLOPS) / 1e12 / (ms_per_step * 1e-3),
            'note': 'executed float64 and float32 flop of all kernels of a batch (committed PMC pass) over ms_per_step, each against its vector '
                    'peak: the share of the step during which the vector pipes would be busy at their peak rates' + stale_txt, 'stale': stale}
        res = {
            'metric': 'frame-vector-lag triples/s, C(t) + fit + R1/R2/NOE pipeline',
            'value': value, 'unit': 'triples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'timed_region_samples_ms_per_step': [x / args.steps * 1e3 for x in samples],
            'timed_region_note': 'the timed region (exactly %d steps between two synchronisations, fill and drain inside) was run %d times; '
                                 'value and ms_per_step are the median' % (args.steps, len(samples)),
            'value_with_h2d': None if with_h2d is None else triples_total / with_h2d,
            'with_h2d': None if with_h2d is None else {
                'ms_per_step': with_h2d * 1e3, 'h2d_GBps': 12.0 * s['frames'] * V / with_h2d / 1e9, 'bytes_per_step': 12 * s['frames'] * V,
                'note': 'the same %d steps (one warm-up run of the same length first) with every step\'s %.0f MB shard arriving from page-locked host memory: '
                        'sr_vectors_append_f32 on a copy stream into two device buffers, batch k + 1 over PCIe while the kernels of batch k run '
                        '(bench.py:PinnedFeed).  `value` (vectors resident in HBM) stays the headline; this is the rate of a stream that starts on the '
                        'host, bound by the link (PCIe Gen5 x16: 63 GB/s spec)' % (args.steps, 12.0 * s['frames'] * V / 1e6)},
            'ms_per_step_random_dispatch': None if random_dispatch is None else random_dispatch * 1e3,
            'dispatch_note': 'the merged fit launch of a group takes its residues longest first by the evaluation counts of the last collected batch '
                             '(spinrelax_amd/pipeline.py:GroupedPipeline, dispatch = history): a prediction, exact in this benchmark because every step is the '
                             'same shard; ms_per_step_random_dispatch is the same timed region (once) with the fixed pseudo-random order of rounds 3-4',
            'ms_per_step_steady': None if steady is None else steady * 1e3,
            'steady_note': 'one run of %d steps, fill and drain amortised; not the headline' % args.steady_steps,
            'spinup': {'seconds': spin_s, 'steps': spin_steps, 'note': 'untimed steps of the same pipeline before the warm-up steps (clock ramp of a fresh box)'},
            'latency_ms': latency, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': ('f32 transforms + f64 mean terms and chunk statistics (C(t): the reference\'s own type); f64 (histogram, fit, J(w), R1/R2/NOE)' if use_f32 else
                      'f64 (C(t) by FFT on f32 inputs; histogram, fit, J(w), R1/R2/NOE)' if use_fft else
                      'f32 dot products / f64 accumulation (C(t)); f64 (histogram, fit, J(w), R1/R2/NOE)'),
            'data': 'synthetic',
            'config': {'workload': 'BASELINE cfg%d%s: %d frames x %d vectors%s, %d chunks x %d frames, %d lags, '
                                   '%s, fits 2/3/5/7/9 params, 1 field' % (cfg, '' if strong else ' per GPU', s['frames'], Vtot if strong else V,
                                                                           (' in all, %d per GPU' % V) if strong and world > 1 else '', s['R'], s['F'], s['L'],
                                                                           'axisymmetric D + q_ext + 72x36 histogram' if cfg >= 3 else 'isotropic D'),
                       'vectors_total': Vtot, 'vectors_per_gpu': V, 'exact_triples_per_gpu': triples, 'exact_triples_total': triples_total,
                       'sharding': 'contiguous vector ranges (spinrelax_amd/dist.py:shard_range; no data-path collective; all-gather of results)',
                       'schedule': ('grouped: pack / C(t) / chunk statistics%s of %d batches back to back, then ONE merged model-order search + '
                                    'relaxation launch over their %d residues (dispatched %s)%s; next group%s'
                                    % ('' if late_hist_used else ' / histogram', group_used, group_used * V,
                                       'in natural order' if args.no_permute else
                                       ('longest first by the evaluation counts of the last collected batch' if args.dispatch == 'history'
                                        else 'in a fixed pseudo-random order'),
                                       ', their histograms in the tail of that launch (released by a signal its last workgroup writes)' if late_hist_used else '',
                                       ' overlaps it' if not args.no_group_overlap else ' waits for it'))
                                   if grouped else 'per batch: every batch launches its own fits, %d batches in flight' % depth_used,
                       'batches_per_group': group_used, 'batches_in_flight': group_used if grouped else depth_used, 'cus_reserved_for_fits': reserve_used,
                       'pack_stream_cus': pack_cus_used, 'late_hist': late_hist_used, 'late_hist_note': late_hist_note,
                       'dispatch': args.dispatch if grouped and not args.no_permute else 'natural'},
            'roofline': roofline,
            'kernels': kernels,
            'stages_alone_ms': {k: round(v, 4) for k, v in alone.items()},
            'fit': {'residues': V, 'selected_orders': {str(listDoG[j]): int((best == j).sum()) for j in range(len(listDoG))},
                    'unfitted': int((best < 0).sum()), 'evaluations_per_batch': nfev_step,
                    'evaluations_by_order': nfev_by_order, 'fits_by_order': nfits_by_order},
            'setup': {'synth_s': gen_s},
            **({'INVALID': 'fits skipped (--dev-skip-fits)'} if args.dev_skip_fits else {}),
        }
        res['cpu_baseline'] = cpu_res            # measured before the GPU was initialised (rank 0, N = 1)
        res['checksums'] = checksums
        if world == 1 and not args.no_cli_wall:
            res['cli_wall_s'] = cli_wall(vecs_host, s, cfg, res['cpu_baseline'])
        print(json.dumps(res))
    # deterministic teardown while the HIP runtime is alive: streams, pinned mirrors and the context's work areas go
    # here, not in __del__ / static destructors at interpreter exit
    del vecs
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
