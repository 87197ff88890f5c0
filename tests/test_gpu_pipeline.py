"""
GPU tests of the device-resident pipeline (spinrelax_amd/pipeline.py): vectors in HBM -> C(t) -> histogram -> model-order
search -> R1/R2/NOE without the host in between, serial and with several batches in flight on a partitioned chip.
The pipeline must give exactly what the staged C-ABI calls give (those are checked against the oracle and the
reference's golden vectors in test_gpu_parity.py), and the same answer for every batch however they overlap.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup():
    import torch
    from spinrelax_amd import synth
    from spinrelax_amd.hip import Context
    s = synth.config_shapes(1)
    vecs = synth.synth_config(1)
    ctx = Context(0)
    dev = torch.device('cuda', 0)
    return dict(torch=torch, synth=synth, ctx=ctx, dev=dev, s=s, vecs=vecs, dvecs=torch.from_numpy(vecs).to(dev))


def _pipe(st, depth, aniso=True, reserve=0, only=False):
    from spinrelax_amd.pipeline import DevicePipeline
    s, synth = st['s'], st['synth']
    V = st['vecs'].shape[1]
    return DevicePipeline(st['ctx'], st['dev'], s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT if aniso else None,
                          Diso=synth.DISO, aniso=synth.DANI if aniso else None, field_MHz=(synth.FIELD_MHZ, 500.0),
                          zeta=synth.ZETA, depth=depth, stream=st['torch'].cuda.Stream(device=st['dev']), reserve_cus=reserve,
                          fits_on_reserved_only=only)


@pytest.mark.parametrize('aniso', [True, False])
def test_pipeline_step_equals_staged_calls(setup, aniso):
    from spinrelax_amd import ct as hostct
    from spinrelax_amd import fitting_Ct_functions as fitCt
    from spinrelax_amd import _hostmath as hm
    st = setup
    ctx, s, synth, vecs = st['ctx'], st['s'], st['synth'], st['vecs']
    pipe = _pipe(st, 1, aniso)
    out = pipe.step(st['dvecs']).copy()
    st['torch'].cuda.synchronize()
    sl = pipe.slots[0]
    V = vecs.shape[1]
    # kernel 1 and 2: same kernels behind the host-pointer entry points
    ctx.set_stream(0)
    Ct, dCt = ctx.ct_palmer(vecs, s['R'], s['F'])
    assert np.array_equal(sl.Ct.cpu().numpy(), Ct) and np.array_equal(sl.dCt.cpu().numpy(), dCt)
    if aniso:
        e = hostct.lambert_edges()
        hist, vecsum, outer = ctx.rotate_hist(vecs[: s['N']], np.array(synth.Q_EXT), e[0], e[1], block_len=s['F'])
        assert np.array_equal(sl.hist.cpu().numpy().reshape(hist.shape), hist)
        assert np.array_equal(sl.vecsum.cpu().numpy(), vecsum)
    # kernel 3b: the search on the same C(t)
    t = np.ascontiguousarray(np.broadcast_to(hostct.calculate_dt(s['dt'], s['F'] * s['dt']), (V, s['L'])))
    ref = fitCt.order_search_device(t, np.ascontiguousarray(Ct.T), np.ascontiguousarray(dCt.T), pipe.listDoG, 0.5, ctx=ctx)
    r = sl.result
    for k in ('best', 'K', 'S2', 'C', 'tau'):
        assert np.array_equal(r[k], ref[k]), k
    assert np.array_equal(r['chi'], ref['chi'], equal_nan=True)
    # kernel 3a: zeta scaling on load == scaling on the host first
    oms, fdd, fcsa, tf, gr = pipe._relax_consts
    z = synth.ZETA
    if aniso:
        Dpar, Dperp = hm.symmtop_from_iso(synth.DISO, synth.DANI)
        want, _ = ctx.relax(2, [Dpar, Dperp], oms, fdd, fcsa, tf, gr, z * ref['S2'], z * ref['C'], ref['tau'], ref['K'],
                            binvecs=pipe.binvecs, weights=hist.reshape(V, -1), noe_mode=0)
    else:
        want, _ = ctx.relax(1, [synth.DISO], oms, fdd, fcsa, tf, gr, z * ref['S2'], z * ref['C'], ref['tau'], ref['K'])
    assert out.shape == want.shape == (2, V, 4, 2)
    assert np.array_equal(out, want, equal_nan=True)
    assert np.all(np.isfinite(out[:, :, :3, 0])) and np.all(out[:, :, 0, 0] > 0)
    pipe.close()


@pytest.mark.parametrize('depth,reserve,only', [(2, 0, False), (4, 16, False), (3, 32, True)])
def test_pipeline_batches_in_flight_are_identical_and_ordered(setup, depth, reserve, only):
    st = setup
    serial = _pipe(st, 1)
    want = serial.step(st['dvecs']).copy()
    want_best = serial.fit_best.copy()
    serial.close()
    pipe = _pipe(st, depth, True, reserve, only)
    assert pipe.reserve_cus == reserve
    seen = []

    def on_finished(slot):
        seen.append((pipe.slots.index(slot), slot.relax_out.copy(), slot.result['best'].copy(), slot.Ct.cpu().numpy().sum()))
    nb = 2 * depth + 1
    pipe.run(st['dvecs'], nb, None, on_finished)
    st['torch'].cuda.synchronize()
    assert [i for i, *_ in seen] == [k % depth for k in range(nb)]
    for _, relax, best, _ in seen:
        assert np.array_equal(relax, want, equal_nan=True)
        assert np.array_equal(best, want_best)
    assert len({c for *_, c in seen}) == 1
    pipe.close()


@pytest.mark.parametrize('group,overlap,nb,late', [(4, True, 11, False), (3, False, 7, False), (8, True, 5, False), (1, True, 3, False),
                                                    (4, True, 11, True), (3, False, 7, True), (8, True, 5, True)])
def test_grouped_schedule_gives_the_serial_results_for_every_batch(setup, group, overlap, nb, late):
    """GroupedPipeline: C(t) / histogram / chunk statistics of a group back to back, then ONE merged model-order search and
    relaxation launch over the group's residues (dispatched in a permuted order).  Every batch must come out exactly as the
    serial pipeline gives it -- every array of the result, C(t) and the histogram --, in order, through full groups, the
    short last group, both group buffers and their reuse.  late: the group's histograms run behind the merged launch, released
    by the signal its last workgroup writes (sr_signal_alloc / sr_stream_wait_signal)."""
    from spinrelax_amd.pipeline import GroupedPipeline
    st = setup
    s, synth = st['s'], st['synth']
    serial = _pipe(st, 1)
    serial.step(st['dvecs'])
    st['torch'].cuda.synchronize()
    sl = serial.slots[0]
    want = {k: v.copy() for k, v in sl.result.items()}
    want_Ct, want_dCt, want_hist = sl.Ct.cpu().numpy(), sl.dCt.cpu().numpy(), sl.hist.cpu().numpy()
    serial.close()
    V = st['vecs'].shape[1]
    pipe = GroupedPipeline(st['ctx'], st['dev'], s['frames'], V, s['R'], s['F'], s['dt'], group=group, overlap=overlap, late_hist=late,
                           q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ, 500.0), zeta=synth.ZETA,
                           stream=st['torch'].cuda.Stream(device=st['dev']))
    seen, enq = [], []

    def on_finished(b):
        seen.append((b.index, {k: v.copy() for k, v in b.result.items()}, b.Ct.cpu().numpy(), b.dCt.cpu().numpy(), b.hist.cpu().numpy()))

    def on_enqueued(grp):
        enq.append((grp.g, tuple(grp.Ct.shape), tuple(grp.relax.shape)))
        return None
    for _ in range(2):                        # a second run reuses both group buffers
        seen.clear()
        enq.clear()
        pipe.run(st['dvecs'], nb, None, on_finished, on_enqueued)
        st['torch'].cuda.synchronize()
        sizes = [min(group, nb - k) for k in range(0, nb, group)]
        assert [g for g, *_ in enq] == sizes
        assert all(cs == (g, s['L'], V) and rs == (2, g * V, 4, 2) for g, cs, rs in enq)
        assert [i for i, *_ in seen] == [j for g in sizes for j in range(g)]
        for _, r, Ct, dCt, hist in seen:
            assert set(r) == set(want)
            for k in want:
                assert np.array_equal(r[k], want[k], equal_nan=True), k
            assert np.array_equal(Ct, want_Ct) and np.array_equal(dCt, want_dCt) and np.array_equal(hist, want_hist)
    assert pipe.step(st['dvecs']) is not None and np.array_equal(pipe.relax_out, want['relax'], equal_nan=True)
    pipe.close()


def test_grouped_schedule_without_signal_memory(setup, monkeypatch):
    """A device / runtime without stream waits on signal memory (sr_signal_alloc fails): late_hist=True says what to do, and
    late_hist=False -- what that message recommends -- runs several groups through BOTH group buffers with the serial results
    (every group object has its signal / epoch attributes although the first allocation already failed)."""
    from spinrelax_amd.pipeline import GroupedPipeline
    from spinrelax_amd.hip import SpinRelaxHipError, Context
    st = setup
    s, synth = st['s'], st['synth']
    serial = _pipe(st, 1)
    serial.step(st['dvecs'])
    st['torch'].cuda.synchronize()
    want = {k: v.copy() for k, v in serial.slots[0].result.items()}
    serial.close()

    def no_signals(self):
        raise SpinRelaxHipError('sr_signal_alloc: hipExtMallocWithFlags(hipMallocSignalMemory) not supported (test)')
    monkeypatch.setattr(Context, 'signal_alloc', no_signals)
    V = st['vecs'].shape[1]
    kw = dict(group=2, q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ, 500.0), zeta=synth.ZETA)
    with pytest.raises(SpinRelaxHipError, match='late_hist=False'):
        GroupedPipeline(st['ctx'], st['dev'], s['frames'], V, s['R'], s['F'], s['dt'], late_hist=True,
                        stream=st['torch'].cuda.Stream(device=st['dev']), **kw)
    pipe = GroupedPipeline(st['ctx'], st['dev'], s['frames'], V, s['R'], s['F'], s['dt'], late_hist=False,
                           stream=st['torch'].cuda.Stream(device=st['dev']), **kw)
    assert all(g.signal is None and g.epoch == 0 for g in pipe.groups)
    seen = []
    pipe.run(st['dvecs'], 5, None, lambda b: seen.append({k: v.copy() for k, v in b.result.items()}), None)
    st['torch'].cuda.synchronize()
    assert len(seen) == 5
    for r in seen:
        for k in want:
            assert np.array_equal(r[k], want[k], equal_nan=True), k
    pipe.close()


def test_batched_order_search_entry_point(setup):
    """sr_expfit_order_search_batched_f64_dev: a shared time axis (t_rows = 1) and a dispatch permutation change nothing in the
    results; bad arguments are refused before any launch."""
    from spinrelax_amd import ct as hostct
    from spinrelax_amd.hip import SpinRelaxHipError
    st = setup
    torch, ctx, s, dev = st['torch'], st['ctx'], st['s'], st['dev']
    ctx.set_stream(0)
    serial = _pipe(st, 1)
    serial.step(st['dvecs'])
    torch.cuda.synchronize()
    sl = serial.slots[0]
    V, L = st['vecs'].shape[1], s['L']
    nO, Pmax = len(serial.listDoG), max(serial.listDoG)
    f64, i32 = dict(device=dev, dtype=torch.float64), dict(device=dev, dtype=torch.int32)
    n = 3 * V
    y, dy = sl.CtT.repeat(3, 1).contiguous(), sl.dCtT.repeat(3, 1).contiguous()
    out = dict(popt=torch.empty((nO, n, Pmax), **f64), dP=torch.empty((nO, n, Pmax), **f64), chisq=torch.empty((nO, n), **f64),
               status=torch.empty((nO, n), **i32), nfev=torch.empty((nO, n), **i32), best=torch.empty((n,), **i32), S2=torch.empty((n,), **f64),
               C=torch.empty((n, Pmax // 2), **f64), tau=torch.empty((n, Pmax // 2), **f64), chi=torch.empty((n,), **f64), K=torch.empty((n,), **i32))
    perm = torch.from_numpy(np.random.RandomState(3).permutation(n).astype(np.int32)).to(dev)

    def call(t_rows=1, nRes=n, order=perm):
        ctx.order_search_batched_dev(serial.t_dev.data_ptr(), t_rows, y.data_ptr(), dy.data_ptr(), nRes, L, serial.listDoG,
                                     serial.tau_guess.data_ptr(), 1, serial.tau_max, serial.chi_thr,
                                     out['popt'].data_ptr(), out['dP'].data_ptr(), out['chisq'].data_ptr(), out['status'].data_ptr(),
                                     out['nfev'].data_ptr(), out['best'].data_ptr(), out['S2'].data_ptr(), out['C'].data_ptr(),
                                     out['tau'].data_ptr(), out['chi'].data_ptr(), out['K'].data_ptr(),
                                     dispatch_order_ptr=None if order is None else order.data_ptr())
        ctx.sync()
        torch.cuda.synchronize()
    call()
    r = sl.result
    for j in range(3):
        assert np.array_equal(out['popt'][:, j * V:(j + 1) * V].cpu().numpy(), r['popt'], equal_nan=True)
        assert np.array_equal(out['nfev'][:, j * V:(j + 1) * V].cpu().numpy(), r['nfev'])
        assert np.array_equal(out['best'][j * V:(j + 1) * V].cpu().numpy(), r['best'])
        assert np.array_equal(out['tau'][j * V:(j + 1) * V].cpu().numpy(), r['tau'])
    with pytest.raises(SpinRelaxHipError):
        call(t_rows=2)
    serial.close()


def test_signals_release_a_waiting_stream(setup):
    """sr_signal_alloc / sr_stream_write_signal / sr_stream_wait_signal: work queued behind a wait runs once the value is there
    (written by another stream here; by the merged fit launch's last workgroup in GroupedPipeline), a value already reached does
    not block, NULL signals are refused -- and so is a wait whose release has not been submitted yet (the ordering rule)."""
    from spinrelax_amd.hip import SpinRelaxHipError
    st = setup
    torch, ctx, dev = st['torch'], st['ctx'], st['dev']
    sig = ctx.signal_alloc()
    a, b = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)
    x = torch.zeros(1 << 20, device=dev)
    big = torch.zeros((8192, 8192), device=dev)
    torch.cuda.synchronize()
    # the ordering rule of the ABI: a wait for a value nobody has submitted a release for is REFUSED (queued ahead of its own
    # release on a shared hardware queue it would never end -- round 4's hang)
    ctx.set_stream(b.cuda_stream)
    with pytest.raises(SpinRelaxHipError):
        ctx.stream_wait_signal(sig, 1)
    # the release is submitted first, behind ~20 ms of work on its (high-priority: own hardware queue) stream ...
    ctx.set_stream(a.cuda_stream)
    with torch.cuda.stream(a):
        for _ in range(8):
            big = big @ big
    ctx.stream_write_signal(sig, 1)
    # ... then the wait: what is queued behind it is parked until the value is there
    ctx.set_stream(b.cuda_stream)
    ctx.stream_wait_signal(sig, 1)
    with torch.cuda.stream(b):
        x.add_(1.0)
        done = torch.cuda.Event()
        done.record(b)
    assert not done.query()                      # parked on the signal: the writer is still behind its matrix products
    done.synchronize()
    assert float(x.sum().item()) == float(1 << 20)
    with pytest.raises(SpinRelaxHipError):
        ctx.stream_wait_signal(sig, 2)           # no release of 2 submitted
    ctx.set_stream(b.cuda_stream)
    ctx.stream_wait_signal(sig, 1)               # already reached (>=): passes
    with torch.cuda.stream(b):
        x.add_(1.0)
    torch.cuda.synchronize()
    assert float(x[0].item()) == 2.0
    with pytest.raises(SpinRelaxHipError):
        ctx.stream_wait_signal(None, 1)
    with pytest.raises(SpinRelaxHipError):
        ctx.stream_write_signal(None, 1)
    ctx.set_stream(0)
    ctx.signal_free(sig)


def test_pipeline_with_orientation_trajectory(setup):
    """Lab-frame vectors + per-frame orientation quaternions through the pipeline (de-tumbling folded into the pack
    kernel) == the pipeline fed the de-tumbled vectors."""
    from conftest import golden
    from spinrelax_amd import ct as hostct
    from spinrelax_amd.pipeline import DevicePipeline
    st = setup
    g = golden('cfg1_detumble.npz')
    s, synth, torch = st['s'], st['synth'], st['torch']
    kw = dict(q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1)
    lab = torch.from_numpy(g['lab']).to(st['dev'])
    p1 = DevicePipeline(st['ctx'], st['dev'], s['frames'], 32, s['R'], s['F'], s['dt'], q_orient=g['q32'],
                        stream=torch.cuda.Stream(device=st['dev']), **kw)
    out1 = p1.step(lab).copy()
    Ct1 = p1.Ct.cpu().numpy()
    body = torch.from_numpy(hostct.detumble_vectors(g['lab'], g['q32'], ctx=st['ctx'])).to(st['dev'])
    p2 = DevicePipeline(st['ctx'], st['dev'], s['frames'], 32, s['R'], s['F'], s['dt'],
                        stream=torch.cuda.Stream(device=st['dev']), **kw)
    out2 = p2.step(body).copy()
    assert np.array_equal(Ct1, p2.Ct.cpu().numpy())
    assert np.max(np.abs(Ct1 / g['Ct64'] - 1)) < 1e-12
    assert np.array_equal(out1, out2, equal_nan=True)
    with pytest.raises(ValueError):
        DevicePipeline(st['ctx'], st['dev'], s['frames'], 32, s['R'], s['F'], s['dt'], q_orient=g['q32'][:5], **kw)


@pytest.mark.parametrize('variant', ['isotropic', 'csa_per_residue', 'orientation_trajectory'])
def test_grouped_schedule_variants_equal_the_serial_pipeline(setup, variant):
    """the other shapes of the path through the grouped schedule: no histogram weights (isotropic diffusion), a CSA value per
    residue (the relaxation constants are tiled over the group's batches), de-tumbling inside the pack"""
    from conftest import golden
    from spinrelax_amd.pipeline import DevicePipeline, GroupedPipeline
    st = setup
    s, synth, torch = st['s'], st['synth'], st['torch']
    V = st['vecs'].shape[1]
    vecs = st['dvecs']
    kw = dict(Diso=synth.DISO, field_MHz=(synth.FIELD_MHZ, 700.0), zeta=synth.ZETA)
    if variant == 'isotropic':
        kw.update(q_rot=None, aniso=None)
    elif variant == 'csa_per_residue':
        kw.update(q_rot=synth.Q_EXT, aniso=synth.DANI, csa=-170e-6 + 1e-6 * np.arange(V))
    else:
        g = golden('cfg1_detumble.npz')
        V = 32
        vecs = torch.from_numpy(g['lab']).to(st['dev'])
        kw.update(q_rot=synth.Q_EXT, aniso=synth.DANI, q_orient=g['q32'])
    serial = DevicePipeline(st['ctx'], st['dev'], s['frames'], V, s['R'], s['F'], s['dt'], depth=1,
                            stream=torch.cuda.Stream(device=st['dev']), **kw)
    serial.step(vecs)
    torch.cuda.synchronize()
    want = {k: v.copy() for k, v in serial.slots[0].result.items()}
    want_hist = serial.slots[0].hist.cpu().numpy()
    serial.close()
    for late in (True, False):
        pipe = GroupedPipeline(st['ctx'], st['dev'], s['frames'], V, s['R'], s['F'], s['dt'], group=3, late_hist=late,
                               stream=torch.cuda.Stream(device=st['dev']), **kw)
        seen = []
        pipe.run(vecs, 5, None, lambda b: seen.append(({k: v.copy() for k, v in b.result.items()}, b.hist.cpu().numpy())))
        torch.cuda.synchronize()
        assert len(seen) == 5
        for r, hist in seen:
            for k in want:
                assert np.array_equal(r[k], want[k], equal_nan=True), (variant, late, k)
            if variant != 'isotropic':
                assert np.array_equal(hist, want_hist)
        pipe.close()


@pytest.fixture(scope='module')
def full_cfg3():
    """BASELINE cfg3, one full batch (100 000 frames x 512 vectors, L = 2048) through the pipeline, once for the tests below"""
    import torch
    from spinrelax_amd import synth
    from spinrelax_amd.hip import Context
    from spinrelax_amd.pipeline import DevicePipeline
    s = synth.config_shapes(3)
    V = 512
    vecs_host = synth.synth_vectors_parallel(s['frames'], V, s['seed'])
    ctx = Context(0)
    dev = torch.device('cuda', 0)
    vecs = torch.from_numpy(vecs_host).to(dev)
    pipe = DevicePipeline(ctx, dev, s['frames'], V, s['R'], s['F'], s['dt'], q_rot=synth.Q_EXT, Diso=synth.DISO, aniso=synth.DANI,
                          field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, depth=1, stream=torch.cuda.Stream(device=dev))
    out = pipe.step(vecs).copy()
    sl = pipe.slots[0]
    st = dict(s=s, V=V, vecs_host=vecs_host, ctx=ctx, pipe=pipe, out=out, y=sl.CtT.cpu().numpy(), dy=sl.dCtT.cpu().numpy(),
              Ct=sl.Ct.cpu().numpy(), dCt=sl.dCt.cpu().numpy(), r={k: v.copy() for k, v in sl.result.items()}, hist=sl.hist.cpu().numpy())
    del vecs
    yield st
    pipe.close()
    ctx.close()


def test_full_size_batch_search_equals_host_driven_search(full_cfg3):
    """BASELINE cfg3, one full batch (100 000 frames x 512 vectors, L = 2048): the one-launch model-order search inside
    the pipeline against the host-driven search that calls the single-order solver order by order -- identical
    selection and bit-identical parameters / chi^2 for every residue and every attempted order (1 513 fits, up to
    286 evaluations each).  Also the size-independent properties of the result."""
    from spinrelax_amd import fitting_Ct_functions as fitCt
    st = full_cfg3
    s, V, ctx, pipe, out, y, dy, r, hist = (st[k] for k in ('s', 'V', 'ctx', 'pipe', 'out', 'y', 'dy', 'r', 'hist'))
    ctx.set_stream(0)
    search = fitCt.OrderSearchBatch(pipe.t_host, y, pipe.listDoG, 0.5)
    runner = fitCt.host_runner(pipe.t_host, y, dy, ctx=ctx)
    while True:
        req = search.request()
        if req is None:
            break
        search.submit(*runner(req['nParams'], req['p0'], req['idx']))
    assert np.array_equal(r['best'], search.best)
    nfits = 0
    for j, res in enumerate(search.per_order):
        nP = res['nParams']
        idx = np.flatnonzero(r['status'][j] != -100)
        assert np.array_equal(idx, np.flatnonzero(~np.isnan(res['p0'][:, 0])))
        assert np.array_equal(r['popt'][j][idx, :nP], res['popt'][idx])
        assert np.array_equal(r['chisq'][j][idx], res['chiSq'][idx])
        nfits += idx.size
    assert nfits > 1000
    # properties: C(t) starts near 1 and decays, every residue got a model, S2 + sum C <= 1 within bounds, histograms
    # account for every frame, relaxation rates are positive and finite
    assert np.all(y[:, 0] > 0.8) and np.all(y[:, 0] <= 1.0 + 1e-12) and np.all(y[:, -1] < y[:, 0])
    assert np.all(r['best'] >= 0) and np.all(r['K'] >= 1)
    assert np.all(r['S2'] >= 0) and np.all(r['S2'] <= 1) and np.all(r['C'] >= 0) and np.all(r['tau'] > 0)
    for i in range(V):
        assert np.all(np.diff(r['tau'][i, :r['K'][i]]) >= 0)
    assert np.all(hist.sum(axis=1) == s['N'])
    assert np.all(np.isfinite(out)) and np.all(out[0, :, 0, 0] > 0) and np.all(out[0, :, 1, 0] > out[0, :, 0, 0] * 0.5)


def test_full_size_cfg3_values_against_the_oracle(full_cfg3):
    """The headline configuration compared VALUE BY VALUE, all 512 vectors (not a slice, not properties): C(t) and dC(t) of
    the production path against the C oracle's float64 per-lag loop (oracle/ct_palmer_oracle.c =
    calculate-Ct-from-traj.py:222-228 on every host core: 7.7e10 shifted products), the 512 spherical histograms against the
    numpy restatement of rotate_vector_simd + xyz_to_rtp + histogramdd (transforms3d_supplement.py:270-296,
    calculate-Ct-from-traj.py:585-626) count for count, and the (vectors, lags) copy the fit reads against the (lags, vectors)
    one the files are written from."""
    import ctypes
    import os
    import sr_oracle as o
    from conftest import ROOT
    from spinrelax_amd import synth
    st = full_cfg3
    s, V, vecs = st['s'], st['V'], st['vecs_host']
    R, F, L, N = s['R'], s['F'], s['L'], s['N']
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libsr_oracle.so'))
    v4 = np.ascontiguousarray(vecs[:N].reshape(R, F, V, 3))
    Cr, dCr = np.empty((L, V)), np.empty((L, V))
    rc = lib.sr_oracle_ct_palmer_f64(v4.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(R), ctypes.c_int64(F), ctypes.c_int64(V),
                                     Cr.ctypes.data_as(ctypes.c_void_p), dCr.ctypes.data_as(ctypes.c_void_p), None)
    assert rc == 0
    del v4
    eC = np.max(np.abs(st['Ct'] - Cr) / np.abs(Cr))
    eD = np.max(np.abs(st['dCt'] - dCr) / np.abs(dCr))
    eDa = np.max(np.abs(st['dCt'] - dCr))
    print('\n[cfg3, all 512 vectors] C(t) max rel err %.2e, dC(t) %.2e relative, %.2e absolute (%d x %d values each)' % (eC, eD, eDa, L, V))
    # the production kernel of this chunk length runs float32 transforms (k_ct_rfft32): the float32 bars of the direct path --
    # C(t) 1e-7, dC(t) 1e-6 / sqrt(F/2) / (sqrt(R) - 1) = 5.7e-9 absolute (tests/test_gpu_parity.py:dct_close); measured 5e-8, 3e-9
    assert eC < 1e-7
    assert np.all(np.abs(st['dCt'] - dCr) <= np.maximum(1e-6 * np.abs(dCr), 1e-6 / np.sqrt(F / 2.0) / (np.sqrt(R) - 1.0)))
    assert np.array_equal(st['y'], st['Ct'].T) and np.array_equal(st['dy'], st['dCt'].T)
    q = np.array(synth.Q_EXT)
    hist = st['hist'].reshape(V, 72, 36)
    for v0 in range(0, V, 64):                               # 64 vectors at a time: bounded host memory
        href, _ = o.lambert_histogram(o.rotate_vector_simd(vecs[:N, v0:v0 + 64], q))
        assert np.array_equal(hist[v0:v0 + 64], href), v0
    assert hist.sum() == N * V


def test_full_size_cfg3_table_against_the_oracle_chain(full_cfg3, tmp_path):
    """The rest of the headline configuration value by value: for ALL 512 residues the oracle runs the reference's chain on the
    C(t) the device produced -- optimised_curve_fitting through scipy (fitting_Ct_functions.py:278-345), zeta scaling, J(w) and
    R1 / R2 / NOE / rho over the 2 592 histogram bins (calculate-relaxations-from-Ct.py:125-191, 747-750) -- and the pipeline's
    model selection and table are compared with it residue by residue.  Counts are committed numbers (tests/golden/
    fit_trial_tallies.json): how many residues select the oracle's order, how many of the table's rows are within 1e-6."""
    import sr_oracle as o
    from conftest import committed_tally
    import oracle_workers
    from spinrelax_amd import synth
    st = full_cfg3
    s, V, pipe, r = st['s'], st['V'], st['pipe'], st['r']
    t = np.asarray(pipe.t_host[0], dtype=np.float64)
    y, dy = st['y'], st['dy']
    fits = oracle_workers.fit_all(t, y, dy, str(tmp_path), nproc=16)        # plain child processes: numpy + scipy, no GPU
    assert len(fits) == V and all(f is not None for f in fits)
    nP_ref = np.array([f[0] for f in fits])
    nP_dev = np.array([pipe.listDoG[b] for b in r['best']])
    same = nP_ref == nP_dev
    # the table from the ORACLE's fitted parameters and the oracle's histogram weights
    B0 = o.B0_from_Hz(synth.FIELD_MHZ * 1e6)
    Dpar, Dperp = o.symmtop_from_iso(synth.DISO, synth.DANI)
    hist = st['hist'].reshape(V, 72, 36)
    edges = [np.linspace(-np.pi, np.pi, 73), np.linspace(-1.0, 1.0, 37)]
    bv, w = o.convert_LambertCylindricalHist_to_vecs(hist, edges)
    S2 = [synth.ZETA * f[1] for f in fits]
    C = [synth.ZETA * np.asarray(f[2]) for f in fits]
    tau = [np.asarray(f[3]) for f in fits]
    ref = o.obtain_R1R2NOErho('rigid_symmtop', (Dpar, Dperp), B0, S2, C, tau, vecXH=[bv[i] for i in range(V)],
                              weights=[w[i] for i in range(V)], cast32=False)                     # (4, V, 2)
    dev = st['out'][0]                                                                            # (V, 4, 2)
    rel = np.max(np.abs(dev[:, :3, 0] / ref[:3, :, 0].T - 1.0), axis=1)
    chi_rel = np.abs(r['chi'] / np.array([f[4] for f in fits]) - 1.0)
    got = dict(residues=int(V), same_order=int(same.sum()), within_1e6=int((rel < 1e-6).sum()),
               same_order_within_1e6=int((rel[same] < 1e-6).sum()), chi_within_1e5=int((chi_rel[same] < 1e-5).sum()))
    print('\n[cfg3, all 512 residues vs the oracle chain] %s; max rel (same order) %.2e, median %.2e; orders (device) %s'
          % (got, rel[same].max(), np.median(rel), dict(zip(*np.unique(nP_dev, return_counts=True)))))
    want = committed_tally('full_cfg3_chain', 'table', got)
    assert got['residues'] == want['residues']
    for k in ('same_order', 'within_1e6', 'same_order_within_1e6', 'chi_within_1e5'):
        assert got[k] >= want[k], (k, got, want)
    assert np.all(rel[same] < 2e-4)                 # the bound test_gpu_chain.py puts on ill-conditioned fits


def _run_bench(flags, ranks=1, port=29577, env=None):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    launcher = [sys.executable] if ranks == 1 else [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks),
                                                    '--master-addr', '127.0.0.1', '--master-port', str(port)]
    p = subprocess.run(launcher + [os.path.join(root, 'bench.py')] + flags, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0 alone reports
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_device(tmp_path):
    """The N > 1 path of bench.py rehearsed with two ranks sharing this GPU over gloo (RCCL needs one GPU per rank):
    STRONG scaling (the default: the workload's 64 vectors sharded 32 + 32 as spinrelax_amd/dist.py:shard_range cuts them,
    result all-gathers on their own stream, max-over-ranks timing) must reproduce the one-rank run of the same 64 vectors --
    the checksums of the gathered C(t), dC(t), histogram and R1/R2/NOE table are those of the single-process run, `value`
    counts the workload's triples once -- and the weak mode keeps every rank on its own 64 vectors."""
    common = ['--steps', '4', '--warmup', '1', '--vectors', '64', '--no-cpu-baseline', '--no-cli-wall', '--no-kernel-profile', '--spinup-s', '0',
              '--steady-steps', '0', '--repeats', '2']
    two = ['--gpus', '2', '--backend', 'gloo', '--all-ranks-on-device0']
    one = _run_bench(['--gpus', '1'] + common)
    j = _run_bench(two + common, ranks=2, env={'GPU_MAX_HW_QUEUES': '6'})
    assert j['n_gpus'] == 2 and j['scaling'] == 'strong' and j['steps'] == 4
    assert j['config']['vectors_total'] == 64 and j['config']['vectors_per_gpu'] == 32 and j['value'] > 0
    assert j['config']['exact_triples_total'] == one['config']['exact_triples_total'] == 2 * j['config']['exact_triples_per_gpu']
    assert abs(j['value'] - j['config']['exact_triples_total'] / (j['ms_per_step'] * 1e-3)) <= 1e-6 * j['value']
    assert j['fit']['residues'] == 32 and j['fit']['unfitted'] == 0
    assert j['config']['schedule'].startswith('grouped') and j['config']['batches_per_group'] == 4   # the default schedule, with its gathers
    assert one['scaling'] == 'strong' and one['config']['vectors_per_gpu'] == 64
    assert j['checksums'] == one['checksums'] and j['checksums']['shapes']['relax'][1] == 64
    w = _run_bench(two + common + ['--scaling', 'weak'], ranks=2, port=29579, env={'GPU_MAX_HW_QUEUES': '6'})
    assert w['scaling'] == 'weak' and w['config']['vectors_per_gpu'] == 64 and w['config']['vectors_total'] == 128
    assert w['checksums']['shapes']['relax'][1] == 128 and w['checksums'] != one['checksums']


def test_bench_cfg4_workload_flag():
    """`--workload cfg4` (BASELINE configs[3]: 100 000 frames x 2 048 vectors, what 8 GPUs share at 256 vectors each) on one
    device with a reduced vector count: the line names the workload, counts its triples and carries the result checksums."""
    j = _run_bench(['--gpus', '1', '--workload', 'cfg4', '--vectors', '64', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-cli-wall',
                    '--no-kernel-profile', '--spinup-s', '0', '--steady-steps', '0', '--repeats', '1'])
    assert 'cfg4' in j['config']['workload'] and j['config']['vectors_total'] == 64 and j['scaling'] == 'strong'
    assert j['config']['exact_triples_total'] == 24 * 64 * (2048 * 4096 - 2048 * 2049 // 2)
    assert j['checksums']['shapes']['Ct'] == [2048, 64] and j['fit']['unfitted'] == 0


def test_bench_single_rank_under_torchrun_equals_plain_run():
    """The driver's SCALE run launches N = 1 the same way as N = 2, 4, 8 (torch.distributed.run) and compares it with the
    plain `python bench.py` of BENCH: both must take the same code path -- same workload, same shard (rank 0 = vectors
    0 .. V-1), the same fits evaluation for evaluation -- and agree in throughput to the box's run-to-run spread."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ['--gpus', '1', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-cli-wall', '--no-kernel-profile', '--spinup-s', '0.3',
             '--steady-steps', '0', '--repeats', '3']
    outs = []
    for launcher in ([sys.executable], [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                                        '127.0.0.1', '--master-port', '29581']):
        p = subprocess.run(launcher + [os.path.join(root, 'bench.py')] + flags, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
        assert len(lines) == 1
        outs.append(json.loads(lines[0]))
    a, b = outs
    assert a['config'] == b['config'] and a['n_gpus'] == b['n_gpus'] == 1 and a['metric'] == b['metric'] and a['dtype'] == b['dtype']
    assert a['fit'] == b['fit']                                   # same residues, same orders, same evaluation counts
    assert abs(a['value'] / b['value'] - 1) < 0.15


def test_bench_json_contract():
    """bench.py prints ONE JSON line with the agreed keys, the roofline object of the dominant kernel (measured live with
    events on its stream) and the CPU baseline timed beside it (a 1-vector sample here to keep the test short)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '1', '--cpu-sample-vectors', '1', '--spinup-s', '0.2', '--steady-steps', '20'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 6 and j['warmup'] == 1 and j['higher_is_better'] is True
    assert j['unit'] == 'triples/s' and j['data'] == 'synthetic' and j['vs_baseline'] is None and 'workload' in j['config']
    assert abs(j['value'] - j['config']['exact_triples_per_gpu'] / (j['ms_per_step'] * 1e-3)) <= 1e-6 * j['value']
    assert j['scaling'] == 'strong' and j['config']['vectors_total'] == j['config']['vectors_per_gpu'] == 512
    r = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms', 'selection', 'direct_equivalent'):
        assert k in r, k
    # every fraction is a fraction of a bound its kernel can reach: in (0, 1]
    ks = j['kernels']
    assert r['kernel'] in ks and set(ks) >= {'k_ct_rfft32', 'k_ct_palmer', 'k_vechist', 'k_pack_soa', 'k_order_search'}
    top = ks[r['kernel']]
    assert top['cu_ms_per_batch'] == max(v['cu_ms_per_batch'] for k, v in ks.items() if k != 'k_ct_palmer')
    for name, e in ks.items():
        for f in ('frac', 'frac_alone'):
            if e.get(f) is not None:
                assert 0 < e[f] <= 1.0, (name, f, e[f])
    if r['frac'] is not None:
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] <= 1
    ct = ks['k_ct_rfft32']               # the production kernel of this chunk length: float32 transforms, FP32 vector peak
    assert 'fp32' in ct['bound'] and ct['peak'] == 157.3 and 'f32' in j['dtype'] and ct['float64_transform_kernel_alone_ms'] > ct['alone_ms']
    # (two C(t) launches of consecutive batches overlap each other and the fits: a launch LASTS longer than a step)
    assert 0 < ct['in_pipeline_ms'] < j['ms_per_step'] * 3 and ct['alone_ms'] <= ct['in_pipeline_ms'] * 1.05
    assert abs(ct['frac'] - ct['work_per_launch'] / (ct['in_pipeline_ms'] * 1e-3) / 1e12 / ct['peak']) < 1e-9
    assert ct['work_per_launch'] < 0.1 * 8 * j['config']['exact_triples_per_gpu']          # the FFT formulation executes < 10 % of the direct flop
    vh = ks['k_vechist']
    assert vh['bound'] == 'hbm' and abs(vh['achieved'] - vh['work_per_launch'] / (vh['in_pipeline_ms'] * 1e-3) / 1e9) < 1e-6 * vh['achieved']
    fit = ks['k_order_search']
    assert fit['residues_per_s'] > 0 and fit['evaluations_per_s'] > fit['residues_per_s'] and fit['saturated_ms_per_batch'] <= fit['alone_ms']
    assert j['latency_ms']['min'] >= j['ms_per_step'] * 0.9          # one batch alone cannot beat the pipelined period by much
    c = j['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0
    assert set(c['stages_s']) >= {'ct_s', 'rotate_hist_s', 'fit_s', 'relax_s'} and c['all_cores']['cores'] >= 1 and 'cpu_fft_formulation' in c
    assert j['fit']['unfitted'] == 0


def test_workspace_growth_while_another_stream_is_busy(synth_cache):
    """One context driven from two streams: the C(t) call on stream A is still running (direct formulation, ~1 ms) when a
    larger call on stream B makes the context grow the very work area A is using (the raw C(t) sums).  Growth
    synchronises the DEVICE before the old buffer is freed (sr_core.hip:sr_workspace), so A's result must equal the
    result of the same call made alone, and so must B's."""
    import torch
    from spinrelax_amd import synth
    from spinrelax_amd.hip import Context
    s = synth.config_shapes(3)
    nA, nB = 16, 64
    vecs = synth_cache(3, nB)
    N, R, F, L = s['N'], s['R'], s['F'], s['L']
    dev = torch.device('cuda', 0)
    Npad = (vecs.shape[0] + 63) // 64 * 64
    d = torch.from_numpy(vecs).to(dev)
    soa = torch.zeros((nB, 3, Npad), device=dev, dtype=torch.float32)

    def run_alone(nV):
        c = Context(0)
        c.set_option('ct_fft', 0)
        c.set_stream(0)
        c.pack_soa_dev(d.data_ptr(), vecs.shape[0], nB, 0, nB, soa.data_ptr(), Npad)
        Ct = torch.empty((L, nV), device=dev, dtype=torch.float64)
        dCt = torch.empty_like(Ct)
        c.ct_palmer_dev(soa.data_ptr(), Npad, R, F, nV, Ct.data_ptr(), dCt.data_ptr())
        torch.cuda.synchronize()
        out = Ct.cpu().numpy(), dCt.cpu().numpy()
        c.close()
        return out

    refA, refB = run_alone(nA), run_alone(nB)
    assert np.array_equal(refA[0], refB[0][:, :nA])
    c = Context(0)
    c.set_option('ct_fft', 0)
    sA, sB = c.stream_create(), c.stream_create()
    CtA = torch.empty((L, nA), device=dev, dtype=torch.float64)
    dCtA = torch.empty_like(CtA)
    CtB = torch.empty((L, nB), device=dev, dtype=torch.float64)
    dCtB = torch.empty_like(CtB)
    torch.cuda.synchronize()
    try:
        for _ in range(3):
            c.set_stream(sA)
            c.ct_palmer_dev(soa.data_ptr(), Npad, R, F, nA, CtA.data_ptr(), dCtA.data_ptr())
            c.set_stream(sB)
            c.ct_palmer_dev(soa.data_ptr(), Npad, R, F, nB, CtB.data_ptr(), dCtB.data_ptr())    # grows the sums' area
            c.device_sync()
            assert np.array_equal(CtA.cpu().numpy(), refA[0]) and np.array_equal(dCtA.cpu().numpy(), refA[1])
            assert np.array_equal(CtB.cpu().numpy(), refB[0]) and np.array_equal(dCtB.cpu().numpy(), refB[1])
            # a fresh context for the next round so that the area has to grow again
            c.set_stream(0)
            c.stream_destroy(sA)
            c.stream_destroy(sB)
            c.close()
            c = Context(0)
            c.set_option('ct_fft', 0)
            sA, sB = c.stream_create(), c.stream_create()
    finally:
        c.set_stream(0)
        c.stream_destroy(sA)
        c.stream_destroy(sB)
        c.close()

