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
axisymmetric D, q_ext rotation + vecHistogram) PER GPU -- the configuration the north-star quotes its
scaling on; weak scaling: every rank owns its own 512 vectors (rank r = vectors 512 r .. 512 r + 511
of one synthetic trajectory).  `value` = exact triples of all ranks / wall time of a step.

One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (the C(t) kernel): 8 flop x exact triples per launch / mean launch time
                (HIP events on the launch stream, inside the timed region) against the 157.3 TFLOP/s
                FP32 vector peak (= the FP32 MFMA peak) of MI355X_MICROARCH.md; SURVEY.md section 8(d).
                Since the kernel computes the same sums by FFT (k_ct_fft, 4 % of the direct flop, in float64)
                `roofline.executed` gives the executed work against the FP64 vector peak as well.
  cpu_baseline  the reference's algorithm (per-lag float32 numpy einsum, calculate-Ct-from-traj.py:222-228,
                restated in oracle/sr_oracle.py) timed on this host on a bounded sample (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight batch (their fit stragglers run ~25 ms each) + the C(t) stream; the HIP runtime
# multiplexes streams onto 4 queues by default and streams that share a queue serialise (spinrelax_amd/pipeline.py).
# Not more than needed: two processes with 16 queues each on ONE GPU oversubscribe the hardware queues and the
# scheduler's time-slicing makes a step 20x slower (seen in the 2-rank rehearsal on one device).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')

PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: FP32 vector = FP32 matrix peak
PEAK_FP64_TFLOPS = 78.6           # MI355X_MICROARCH.md: FP64 vector
PEAK_HBM_GBS = 8000.0


def measured_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/, produced by
    scripts/make_profiles.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_cfg3_hbm_counters.json')))
    if not files:
        return None, None
    with open(files[-1]) as fp:
        d = json.load(fp)
    for k, v in d.get('kernels', {}).items():
        if k.startswith(kernel_prefix):
            return v['hbm_bytes_corrected'], os.path.relpath(files[-1], ROOT)
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed batches (the pipeline keeps --depth of them in flight; its fill and drain are inside the timed region)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', type=str, default='cfg3', choices=['cfg2', 'cfg3'])
    ap.add_argument('--vectors', type=int, default=None, help='vectors per GPU (default: the config value)')
    ap.add_argument('--reserve-cus', type=int, default=128, help='CUs kept free of the C(t)/histogram kernels for the latency-bound fit kernels (CU-masked stream; 0 = no partition)')
    ap.add_argument('--fits-on-reserved-only', type=int, default=1, help='1/0: confine the fit kernels to the reserved CUs (strict partition)')
    ap.add_argument('--fit-waves', type=int, default=0, help='wavefronts per residue in the model-order search (0 = library default)')
    ap.add_argument('--fit-lds', type=int, default=-1, help='1/0: keep residue data in LDS during the fits (-1 = library default)')
    ap.add_argument('--dev-skip-fits', action='store_true', help='DEVELOPMENT ONLY (invalid as a benchmark): leave the fits and the relaxation kernel out, to see the floor the C(t) side alone sets')
    ap.add_argument('--hist-on-main', action='store_true', help='keep the histogram kernel in line with C(t) (only the pack runs beside it)')
    ap.add_argument('--ct-fft', type=int, default=-1, help='1/0: FFT formulation of the C(t) kernel (-1 = library default)')
    ap.add_argument('--depth', type=int, default=6, help='batches in flight: the straggler tail of the last fit order of batch k overlaps batches k+1 .. k+depth-1 (1 = strictly serial steps)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', type=str, default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)')
    ap.add_argument('--all-ranks-on-device0', action='store_true', help='rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--stage-breakdown', action='store_true', help='after the timed loop, print a synchronised per-stage wall-time breakdown to stderr (diagnostic)')
    ap.add_argument('--cpu-sample-vectors', type=int, default=8)
    return ap.parse_args()


def cpu_baseline(vecs_sample, s):
    """Reference-equivalent CPU path for C(t) on a bounded sample; returns the cpu_baseline object."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sr_oracle as o
    v4 = vecs_sample[: s['N']].reshape(s['R'], s['F'], -1, 3)
    triples = o.exact_triples(s['R'], s['F'], v4.shape[2])
    t0 = time.time()
    o.calculate_Ct_Palmer(v4, dtype=np.float32)          # same numpy calls as the reference, float32, 1 thread
    dt = time.time() - t0
    out = dict(value=triples / dt, unit='triples/s', cores=1, kind='port',
               sample='C(t) stage only: %d chunks x %d frames x %d vectors (%.3g exact triples) of the same '
                      'trajectory, numpy float32 per-lag einsum as calculate-Ct-from-traj.py:222-228, %.1f s'
                      % (s['R'], s['F'], v4.shape[2], triples, dt))
    # the same algorithm as a multi-threaded C loop (oracle/ct_palmer_oracle.c), all host cores
    try:
        import ctypes
        import subprocess
        so = os.path.join(ROOT, 'oracle', 'libsr_oracle.so')
        if not os.path.isfile(so):
            subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libsr_oracle.so'],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        v4c = np.ascontiguousarray(v4, dtype=np.float32)
        L = s['F'] // 2
        Ct = np.empty((L, v4c.shape[2]), dtype=np.float32)
        dCt = np.empty_like(Ct)
        t0 = time.time()
        lib.sr_oracle_ct_palmer_f32_stream(v4c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(s['R']), ctypes.c_int64(s['F']),
                                           ctypes.c_int64(v4c.shape[2]), Ct.ctypes.data_as(ctypes.c_void_p),
                                           dCt.ctypes.data_as(ctypes.c_void_p))
        dt2 = time.time() - t0
        out['all_cores'] = dict(value=triples / dt2, cores=os.cpu_count(), kind='port (C + OpenMP, same per-lag streaming algorithm)')
    except Exception as exc:                                 # the extra figure is optional
        out['all_cores'] = dict(error=str(exc))
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from spinrelax_amd import synth
    from spinrelax_amd.hip import Context
    from spinrelax_amd.pipeline import DevicePipeline

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))

    cfg = 3 if args.workload == 'cfg3' else 2
    s = synth.config_shapes(cfg)
    V = args.vectors or s['V']
    aniso = synth.DANI if cfg == 3 else None
    q = synth.Q_EXT if cfg == 3 else None
    # ---- synthetic shard of this rank: generated on the host BEFORE the GPU is initialised (worker pool forks) ----
    t0 = time.time()
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'], v0=rank * V)
    gen_s = time.time() - t0

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

    vecs = torch.from_numpy(vecs_host).to(dev)           # resident in HBM before the timed region
    ctx = Context(local)
    if args.fit_waves:
        ctx.set_option('fit_waves', args.fit_waves)
    if args.fit_lds >= 0:
        ctx.set_option('fit_lds', args.fit_lds)
    if args.ct_fft >= 0:
        ctx.set_option('ct_fft', args.ct_fft)
    triples = synth.exact_triples(s['R'], s['F'], V)
    pipe = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=q, Diso=synth.DISO, aniso=aniso,
                          field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=args.depth,
                          stream=torch.cuda.Stream(device=dev), reserve_cus=args.reserve_cus,
                          fits_on_reserved_only=bool(args.fits_on_reserved_only), hist_on_aux=not args.hist_on_main)
    stream = pipe.main
    if args.dev_skip_fits:
        pipe.stage_fit = lambda s=None: None
        pipe.stage_relax = lambda s=None: None
    ctx.set_stream(stream.cuda_stream)

    with torch.cuda.stream(stream):

        # Result exchange: every rank ends up with C(t), dC(t), the histogram and the R1/R2/NOE
        # table of all vectors.  The all-gathers of a finished batch run on their own stream, beside the C(t) launches
        # the host has already queued for the following batches; the main stream only waits for them before it
        # reuses that batch's buffers (depth batches later).
        gstream = torch.cuda.Stream(device=dev) if world > 1 else None
        gbuf = {}

        def gather_results(slot):
            if world == 1:
                return
            with torch.cuda.stream(gstream):
                for name in ('Ct', 'dCt', 'hist'):
                    tns = getattr(slot, name)
                    key = (id(slot), name)
                    if key not in gbuf:
                        gbuf[key] = [torch.empty_like(tns) for _ in range(world)]
                    dist.all_gather(gbuf[key], tns)
                key = (id(slot), 'relax')
                if key not in gbuf:
                    gbuf[key] = [torch.empty_like(slot.relax) for _ in range(world)]
                dist.all_gather(gbuf[key], slot.relax)
                ev = torch.cuda.Event()
                ev.record(gstream)
            pipe.main.wait_event(ev)

        def run_batches(nb, events=None):
            pipe.run(vecs, nb, events, on_finished=gather_results)
            torch.cuda.synchronize()

        pipe.prime(vecs)                 # set-up (code objects, first touch of every in-flight slot), not a warm-up step
        if world > 1:                    # set-up: communicator and gather buffers exist before anything is timed
            for sl in pipe.slots:
                gather_results(sl)
            torch.cuda.synchronize()
            dist.barrier()
        run_batches(args.warmup)
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_batches(args.steps, events)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0

    if args.stage_breakdown and rank == 0:
        with torch.cuda.stream(stream):
            names = ['pack', 'ct', 'hist', 'transpose', 'fit', 'relax', 'download']
            s0 = pipe.slots[0]
            ctx.set_stream(stream.cuda_stream)
            fns = [lambda: pipe.stage_pack(vecs), lambda: pipe.stage_ct(s0), lambda: pipe.stage_hist(s0),
                   lambda: pipe.stage_transpose(s0), lambda: pipe.stage_fit(s0), lambda: pipe.stage_relax(s0),
                   lambda: pipe.stage_download(s0)]
            acc = {n: 0.0 for n in names}
            for _ in range(3):
                for n, fn in zip(names, fns):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    acc[n] += (time.perf_counter() - t1) / 3
            print('stage wall ms:', {k: round(v * 1e3, 3) for k, v in acc.items()}, 'fit nfev total', pipe.nfev_total, file=sys.stderr)
            for nP, nf in pipe.nfev_last.items():
                print('  order %d: %d fits, nfev mean %.1f median %d p95 %d max %d, >=%d: %d' % (nP, nf.size, nf.mean(), np.median(nf), np.percentile(nf, 95), nf.max(), 100 * nP, int((nf >= 100 * nP).sum())), file=sys.stderr)

    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ct_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    hist_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = triples * world / (elapsed / args.steps)
        achieved = 8.0 * triples / (ct_ms * 1e-3) / 1e12
        use_fft = (args.ct_fft != 0) and 1024 < s['F'] + s['L'] <= 8192
        kname = 'k_ct_fft' if use_fft else 'k_ct_palmer'
        traffic, traffic_src = measured_traffic(kname) if cfg == 3 and V == 512 else (None, None)
        if use_fft:
            need = s['F'] + s['L']
            M = 2048 if need <= 2048 else (4096 if need <= 4096 else (6144 if need <= 6144 else 8192))
            # executed float64 work of the FFT formulation: 4 complex M-point transforms (5 M log2 M flop each), the
            # power spectra of 3 packed pairs (12 flop per frequency each) and the 6 products per frame
            exec_flop = s['R'] * V * (4 * 5 * M * np.log2(M) + 3 * 12 * M + 6 * s['F'])
            executed = {'formulation': 'Wiener-Khinchin: 6 float64 autocorrelations by FFT, whole %d-point transform resident in LDS' % M,
                        'flop_per_launch': float(exec_flop), 'fraction_of_direct_flop': float(exec_flop / (8.0 * triples)),
                        'achieved': float(exec_flop / (ct_ms * 1e-3) / 1e12), 'peak': PEAK_FP64_TFLOPS, 'unit': 'TFLOP/s (float64 vector)',
                        'frac': float(exec_flop / (ct_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS),
                        'note': 'latency-bound: one 4-wave workgroup per CU (114-152 KB LDS, ~450 registers per lane)'}
        else:
            executed = {'formulation': 'direct shifted products, float32 FMA', 'flop_per_launch': 8.0 * triples, 'fraction_of_direct_flop': 1.0,
                        'achieved': achieved, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s (float32 vector)', 'frac': achieved / PEAK_FP32_TFLOPS}
        best, _ = pipe.fit_best, None
        res = {
            'metric': 'frame-vector-lag triples/s, C(t) + fit + R1/R2/NOE pipeline',
            'value': value, 'unit': 'triples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64 (C(t) by FFT on f32 inputs; histogram, fit, J(w), R1/R2/NOE)' if use_fft else 'f32 dot products / f64 accumulation (C(t)); f64 (histogram, fit, J(w), R1/R2/NOE)',
            'data': 'synthetic',
            'config': {'workload': 'BASELINE cfg%d per GPU: %d frames x %d vectors, %d chunks x %d frames, %d lags, '
                                   '%s, fits 2/3/5/7/9 params, 1 field' % (cfg, s['frames'], V, s['R'], s['F'], s['L'],
                                                                           'axisymmetric D + q_ext + 72x36 histogram' if cfg == 3 else 'isotropic D'),
                       'vectors_per_gpu': V, 'exact_triples_per_gpu': triples, 'sharding': 'vectors (no data-path collective; all-gather of results)', 'batches_in_flight': pipe.depth, 'cus_reserved_for_fits': pipe.reserve_cus},
            'roofline': {'bound': 'valu-fp32 (non-MFMA vector FMA; FP32 MFMA peak is the same 157.3): ALGORITHMIC flop of the path, 8 per triple (SURVEY 8(d)), '
                                  'over the measured duration of the kernel that computes C(t) -- see `executed` for what that kernel actually executes',
                         'kernel': kname,
                         'achieved': achieved, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_TFLOPS,
                         'traffic': traffic, 'traffic_unit': 'bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)', 'traffic_source': traffic_src,
                         'algorithmic_bytes': 12 * s['N'] * V + 8 * s['R'] * s['L'] * V, 'kernel_ms': ct_ms, 'flop_per_triple': 8,
                         'streaming_equiv_GBps': 24.0 * triples / (ct_ms * 1e-3) / 1e9,
                         'streaming_equiv_frac_of_hbm': 24.0 * triples / (ct_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         'executed': executed},
            'stages_ms': {'ct_palmer': ct_ms, 'rotate_hist': hist_ms,
                          'rotate_hist_GBps': 12.0 * s['N'] * V / (hist_ms * 1e-3) / 1e9},
            'fit': {'residues': V, 'selected_orders': {str(pipe.listDoG[j]): int((best == j).sum()) for j in range(len(pipe.listDoG))},
                    'unfitted': int((best < 0).sum())},
            'setup': {'synth_s': gen_s},
            **({'INVALID': 'fits skipped (--dev-skip-fits)'} if args.dev_skip_fits else {}),
        }
        if world == 1 and not args.no_cpu_baseline:
            nvs = min(args.cpu_sample_vectors, V)
            res['cpu_baseline'] = cpu_baseline(vecs_host[:, :nvs], s)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res))
    # deterministic teardown while the HIP runtime is alive: streams, pinned mirrors and the context's work areas go
    # here, not in __del__ / static destructors at interpreter exit
    pipe.close()
    del vecs
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
