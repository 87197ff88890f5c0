"""
Device-resident single-process pipeline over one shard of vectors:

    vectors (N, V, 3) float32 in HBM
      -> kernel 0  pack into per-vector planes
      -> kernel 1  C(t), dC(t)                     (calculate-Ct-from-traj.py:200-238)
      -> kernel 2  rotation + Lambert histogram + mean vector + S2 sums   (:541-646)
      -> kernel 3b model-order search: multi-exponential fits of orders 2,3,5,7,9 with the reference's quality flags
                   and accept/reject sequence, ONE launch   (fitting_Ct_functions.py:278-345, 359-374)
      -> kernel 3a J(omega), R1/R2/NOE/rho with the histogram as weights  (calculate-relaxations-from-Ct.py:125-191)

This is what run-all.bash's Step 3 + Step 4 compute (run-all.bash:476-533) without the text files in
between; bench.py times it.  torch is used for device memory, streams and events only; every computation
goes through the C ABI.

No host in the loop.  Between the vectors and the R1/R2/NOE table nothing returns to the host: a batch is
seven kernel launches and two asynchronous copies into pinned memory.

Two kinds of work.  The C(t) kernel is throughput work: 12 288 workgroups that keep every CU full for a millisecond.  The
fits are latency work: one workgroup per residue, hundreds of dependent solver iterations, and a heavy tail (median 27
function evaluations per residue in the benchmark data, the slowest ~300: 6.5 ms on one CU while the median residue takes
0.3 ms).  Two schedules:

DevicePipeline    per batch: C(t) on two alternating main streams, the histogram of this batch and the pack of the next one on
                  an auxiliary stream, and chunk statistics, fits, relaxation and copies on ONE STREAM PER BATCH IN FLIGHT
                  (`depth`), so that the stragglers of consecutive batches overlap each other and the following C(t)
                  launches.  Lowest latency per batch; serial form (depth 1) for the drop-in calls and the tests.
GroupedPipeline   throughput over many batches: the C(t)-side kernels of a GROUP of batches back to back, then one merged
                  fit launch over the group's residues (below).  What bench.py times by default.

The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): streams that share a queue serialise
(two fit streams on one queue run their stragglers one after the other), so set GPU_MAX_HW_QUEUES >= streams in use before
the first HIP call (bench.py does).  CU-masked streams (`reserve_cus`, `aux_cus`: parts of the chip set aside for the fits or
for the bandwidth kernels) are kept as options of DevicePipeline; neither pays on MI355X.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import ct as hostct
from . import fitting_Ct_functions as fitCt
from . import _hostmath as hm
from . import spectral_densities as sd


def _pinned_array(ctx, n, dtype):
    """numpy array over page-locked memory owned by the library (sr_host_alloc); returns (array, address)."""
    dtype = np.dtype(dtype)
    nbytes = max(1, n) * dtype.itemsize
    addr = ctx.host_alloc(nbytes)
    buf = (ctypes.c_char * nbytes).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n), addr


class _Slot:
    """Device buffers, pinned host mirrors and stream of one in-flight batch."""

    def __init__(self, ctx, dev, V, L, R, nbins, nO, Pmax, E, stream, need_fitwork, psum_len=0):
        f64 = dict(device=dev, dtype=torch.float64)
        i32 = dict(device=dev, dtype=torch.int32)
        Kmax = Pmax // 2
        self.Ct = torch.empty((L, V), **f64)
        self.dCt = torch.empty((L, V), **f64)
        self.CtT = torch.empty((V, L), **f64)
        self.dCtT = torch.empty((V, L), **f64)
        self.hist = torch.empty((V, nbins), **f64)
        self.vecsum = torch.empty((V, 3), **f64)
        self.outer = torch.empty((R, V, 6), **f64)
        self.psum = torch.empty((psum_len,), **f64) if psum_len else None     # raw C(t) sums of this batch
        # every float64 / int32 result of the search in ONE buffer each, so that a batch needs two copies to the host
        self._dlayout = (('popt', (nO, V, Pmax)), ('dP', (nO, V, Pmax)), ('chisq', (nO, V)), ('S2', (V,)), ('chi', (V,)),
                         ('C', (V, Kmax)), ('tau', (V, Kmax)), ('relax', (E, V, 4, 2)))
        self._ilayout = (('status', (nO, V)), ('nfev', (nO, V)), ('best', (V,)), ('K', (V,)))
        nd = sum(int(np.prod(sh)) for _, sh in self._dlayout)
        ni = sum(int(np.prod(sh)) for _, sh in self._ilayout)
        self.dres = torch.empty((nd,), **f64)
        self.ires = torch.empty((ni,), **i32)
        for buf, layout in ((self.dres, self._dlayout), (self.ires, self._ilayout)):
            o = 0
            for name, sh in layout:
                n = int(np.prod(sh))
                setattr(self, name, buf[o:o + n].view(sh))
                o += n
        self.fitwork = torch.empty((V, L), **f64) if need_fitwork else None
        # pinned mirrors owned by the library, filled with sr_memcpy_d2h_async on the batch's stream: torch keeps no
        # per-stream record of these copies, so the streams can be destroyed when the pipeline closes
        self._ctx = ctx
        self.h_dres, self._h_dres_addr = _pinned_array(ctx, nd, np.float64)
        self.h_ires, self._h_ires_addr = _pinned_array(ctx, ni, np.int32)
        self.stream = stream
        self.front_done = None
        self.hist_done = None
        self.done = None
        self.psum_free = None      # the chunk statistics of the slot's previous batch have consumed its raw sums
        self.guard = None          # event of a device-side consumer of the slot's result buffers (run(on_enqueued=...))
        self.busy = False
        self.relax_out = None
        self.result = None

    def host_results(self):
        """COPIES of the pinned mirrors (valid after `done`): the next batch on this slot overwrites the mirrors, the
        arrays handed to the caller stay what they were."""
        out = {}
        for buf, layout in ((self.h_dres.copy(), self._dlayout), (self.h_ires.copy(), self._ilayout)):
            o = 0
            for name, sh in layout:
                n = int(np.prod(sh))
                out[name] = buf[o:o + n].reshape(sh)
                o += n
        return out

    def release(self):
        for name in ('_h_dres_addr', '_h_ires_addr'):
            addr = getattr(self, name, None)
            if addr:
                self._ctx.host_free(addr)
                setattr(self, name, None)
        self.h_dres = self.h_ires = None


class DevicePipeline:
    def __init__(self, ctx, device, frames, V, R, F, dt, q_rot=None, Diso=None, aniso=None, field_MHz=(600.133,),
                 zeta=0.890023, histBinX=72, listDoG=(2, 3, 5, 7, 9), csa=None, depth=1, stream=None, reserve_cus=0,
                 fits_on_reserved_only=False, chiSqThreshold=0.5, q_orient=None, hist_on_aux=True, v0=0, aux_cus=0, fit_priority=0, plane_buffers=3,
                 pack_cus=0):
        self.ctx = ctx
        self.dev = device
        self.frames, self.V, self.R, self.F, self.dt = frames, V, R, F, dt
        self.v0 = int(v0)         # first vector of this shard inside the (frames, Vtot, 3) array handed to the stages
        self.L = F // 2
        self.N = R * F
        self.Npad = (frames + 63) // 64 * 64
        self.q = None if q_rot is None else np.asarray(q_rot, dtype=np.float64)
        self.Diso, self.aniso = Diso, aniso
        self.fields = tuple(field_MHz)
        self.zeta = zeta
        self.listDoG = tuple(int(n) for n in listDoG)
        self.chi_thr = chiSqThreshold
        self.edges = hostct.lambert_edges(histBinX)
        self.nbins = histBinX * int(histBinX / 2)
        self.depth = max(1, int(depth))
        self.main = stream if stream is not None else torch.cuda.current_stream(device)
        self._owned_streams = []
        self._main_bits = None
        resv_words = None
        self.reserve_cus = 0
        self.fit_priority = int(fit_priority)
        info = ctx.device_info()
        if reserve_cus and self.depth > 1:
            # Mask bit i is CU i/8 of XCD i%8 on MI355X (scripts/dev/cumask.py): a multiple of 8 taken from the top
            # keeps the 8 XCDs balanced.
            ncu = info['n_cu']
            nx = 8
            r = min(ncu - nx, (int(reserve_cus) + nx - 1) // nx * nx)
            self.reserve_cus = r
            self.main = self._masked_stream(range(ncu - r), ncu)
            self._main_bits = (range(ncu - r), ncu)
            if fits_on_reserved_only:
                resv_words = self._mask_words(range(ncu - r, ncu), ncu)
        # Bandwidth kernels on their own few CUs.  The compute kernels (C(t): 249 VGPRs, fits: 256) fill the register file
        # of every CU they run on, so a pack / histogram wave launched beside them only gets a slot when a compute
        # workgroup retires -- the HBM-bound and the issue-bound work then take turns instead of overlapping.  With
        # `aux_cus` CUs (a multiple of 8: the same number from every XCD) set aside for the auxiliary stream and the
        # compute streams masked off them, pack + histogram stream at what those CUs can pull from HBM, fully beside
        # the compute kernels on the rest of the chip.
        self.aux_cus = 0
        aux_words = None
        if aux_cus and self.depth > 1 and not reserve_cus:
            ncu = info['n_cu']
            r = min(ncu - 8, (int(aux_cus) + 7) // 8 * 8)
            self.aux_cus = r
            self.main = self._masked_stream(range(ncu - r), ncu)
            self._main_bits = (range(ncu - r), ncu)
            resv_words = self._mask_words(range(ncu - r), ncu)          # fits: same complement as the main stream
            aux_words = self._mask_words(range(ncu - r, ncu), ncu)
        Pmax = max(self.listDoG)
        E = len(self.fields)
        need_fitwork = True      # per-batch weight scratch for residues that are not LDS-resident (concurrent launches)
        self.slots = []
        for i in range(self.depth):
            st = self.main if self.depth == 1 else self._fit_stream(resv_words)
            self.slots.append(_Slot(ctx, device, V, self.L, R, self.nbins, len(self.listDoG), Pmax, E, st, need_fitwork,
                                    psum_len=V * R * ctx.psum_stride(F)))
        # planes: two buffers and an auxiliary stream when batches overlap -- the pack of batch k+1 and the histogram of
        # batch k (bandwidth / FP64 work) run beside the C(t) launch of batch k (FP32 issue bound) instead of in line with it
        self.soa = torch.empty((V, 3, self.Npad), device=device, dtype=torch.float32)
        # (three plane buffers: the pack of batch k+1 only needs the C(t) launch of batch k-2 to be done, so it is off the
        # critical path of two C(t) launches that overlap each other)
        self.NB = max(2, int(plane_buffers)) if self.depth > 1 else 1
        self.soa_bufs = [self.soa] + [torch.empty_like(self.soa) for _ in range(self.NB - 1)]
        # `pack_cus`: the auxiliary stream (the pack kernel, per-batch histograms) confined to this many CUs while the compute
        # streams keep the whole chip.  The pack is HBM-bound and hardly issues (83 % of its waves' lifetime parked); spread
        # over all 256 CUs its waves sit beside every C(t) workgroup pair (16 + 2 x 248 VGPRs fill a SIMD exactly) and its
        # bursts hit the memory system while those workgroups prefetch.  On 96-160 CUs it still streams as fast as it needs
        # (0.33 ms in the pipeline either way) and half of the C(t) workgroups never meet it: steady state 2.13 -> 2.09 ms
        # per step (same box, alternating runs, two rounds each; 64 CUs: 2.17, 32: 2.6; the chunk-statistics stream
        # confined as well: no further change).  Measured with the LDS-tile pack of rounds 1-3; with the register pack (82 VGPRs, no
        # LDS) 0 / 64 / 128 / 192 CUs are within the noise of each other and 128 stayed.
        # CU bit i belongs to XCD i % 8, so a prefix of 8 k bits is k CUs per XCD.
        self.pack_cus = 0
        if pack_cus and aux_words is None and self.depth > 1:
            ncu = info['n_cu']
            n = min(ncu, (int(pack_cus) + 7) // 8 * 8)
            if n < ncu:
                self.pack_cus = n
                aux_words = self._mask_words(range(0, n), ncu)
        self.aux = (self._borrow(aux_words) if aux_words is not None else torch.cuda.Stream(device=device)) if self.depth > 1 else None
        self._packed_ev = [None] * self.NB
        self.hist_on_aux = hist_on_aux
        # Consecutive C(t) launches alternate between two streams (one per plane buffer): the next grid's workgroups fill the
        # slots the previous grid's last, partially filled round leaves (a twelfth of a launch) and its launch latency
        # (~0.15 ms between two 12 288-workgroup kernels on one stream) disappears behind it.
        self.main_alt = None
        if self.depth > 1 and hist_on_aux:
            if self._main_bits is not None:
                self.main_alt = self._masked_stream(*self._main_bits)
            else:
                self.main_alt = torch.cuda.Stream(device=device, priority=getattr(self.main, 'priority', 0))
        # the mean / std over the chunks and the transposes only feed this batch's fits: they run on the batch's own
        # stream, so that the main stream issues the C(t) kernels back to back (4.50 -> 4.41 ms per step)
        self.tail_on_slot_stream = True
        self._ct_done_ev = [None] * self.NB
        t = hostct.calculate_dt(dt, F * dt)
        self.t_host = np.ascontiguousarray(np.broadcast_to(t, (V, self.L)))
        self.t_dev = torch.from_numpy(self.t_host).to(device)
        self.tau_max = self.t_host[0, -1] * 10                     # fitting_Ct_functions.py:324
        self.tau_guess = torch.from_numpy(fitCt.tau_guesses(t, self.listDoG)).to(device)
        self.binvecs = hm.lambert_bin_vectors(self.edges)
        self.binvecs_dev = torch.from_numpy(np.ascontiguousarray(self.binvecs, dtype=np.float64)).to(device)
        # per-frame de-tumbling: lab-frame vectors in, rotated by conj(q_orient(t)) inside the pack
        self.quat_dev = None
        if q_orient is not None:
            qc = np.array(q_orient, dtype=np.float64)
            if qc.shape != (frames, 4):
                raise ValueError('q_orient must hold one quaternion (w x y z) per frame: (%d, 4)' % frames)
            qc[:, 1:] *= -1.0
            self.quat_dev = torch.from_numpy(np.ascontiguousarray(hostct.vecnorm_NDarray(qc))).to(device)
        self.csa = csa
        self._packed = False
        self.nfev_total = 0
        self.nfev_last = {}
        self.fit_best = None
        self.relax_out = None
        # field-dependent constants of the old API (spectral_densities.py:1630-1645, 1696-1701), once, resident in HBM
        oms, fdd, fcsa, tf, gr = [], [], [], [], []
        for MHz in self.fields:
            RObj = sd.relaxationModel('NH', 2.0 * np.pi * (MHz * 1e6) / 267.513e6)
            RObj.set_time_unit('ps')
            oms.append(RObj.omega)
            fdd.append(RObj.get_f_DD())
            c = np.repeat(RObj.gX.csa, V) if csa is None else np.asarray(csa, dtype=float)
            fcsa.append(RObj.get_f_CSA(c))
            tf.append(RObj.time_fact)
            gr.append(RObj.gH.gamma / RObj.gX.gamma)
        self._relax_consts = (np.array(oms), np.array(fdd), np.array(fcsa), np.array(tf), np.array(gr))
        self._relax_dev = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device) for a in self._relax_consts]
        torch.cuda.synchronize(device)

    # ---- streams ----
    @staticmethod
    def _mask_words(bits, ncu):
        w = [0] * ((ncu + 31) // 32)
        for b in bits:
            w[b // 32] |= 1 << (b % 32)
        return w

    def _borrow(self, words):
        """A CU-masked stream created by the library (hipExtStreamCreateWithCUMask) and wrapped for torch's event API.
        The pipeline owns it: close() destroys it.  Nothing but kernel launches, library copies and event records /
        waits ever happens on it, so no other runtime holds state that outlives the stream."""
        h = self.ctx.stream_create(words)
        self._owned_streams.append(h)
        return torch.cuda.ExternalStream(h, device=self.dev)

    def _masked_stream(self, bits, ncu):
        return self._borrow(self._mask_words(bits, ncu))

    def _fit_stream(self, resv_words):
        if resv_words is None:
            return torch.cuda.Stream(device=self.dev, priority=self.fit_priority)
        return self._borrow(resv_words)

    def close(self):
        """Deterministic teardown: wait for the device, drop every event / stream wrapper, free the pinned mirrors and
        destroy the CU-masked streams.  The pipeline must not be used afterwards."""
        if getattr(self, '_closed', False):
            return
        self._closed = True
        torch.cuda.synchronize(self.dev)
        self.ctx.device_sync()
        self.ctx.set_stream(0)
        for s in self.slots:
            s.front_done = s.hist_done = s.done = s.psum_free = s.guard = None
            s.stream = None
            s.release()
        self._packed_ev = [None] * self.NB
        self._ct_done_ev = [None] * self.NB
        self.main = None
        self.main_alt = None
        self.aux = None
        owned, self._owned_streams = self._owned_streams, []
        for h in owned:
            self.ctx.stream_destroy(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- stages (each enqueues on the context's current stream) ----
    def stage_pack(self, vecs, soa=None):
        soa = self.soa if soa is None else soa
        if self.quat_dev is not None:
            self.ctx.pack_soa_rot_dev(vecs.data_ptr(), self.frames, vecs.shape[1], self.v0, self.V, self.quat_dev.data_ptr(),
                                      soa.data_ptr(), self.Npad)
        else:
            self.ctx.pack_soa_dev(vecs.data_ptr(), self.frames, vecs.shape[1], self.v0, self.V, soa.data_ptr(), self.Npad)

    def stage_ct(self, s=None, soa=None, mid_event=None, finalize=True):
        """C(t): raw sums (the dominant kernel), then mean / std over the chunks.  mid_event is recorded between the two."""
        s = s or self.slots[0]
        soa = self.soa if soa is None else soa
        self.ctx.ct_sums_dev(soa.data_ptr(), self.Npad, self.R, self.F, self.V, s.psum.data_ptr())
        if mid_event is not None:
            mid_event.record(torch.cuda.current_stream(self.dev))
        if finalize:
            self.stage_ct_finalize(s)

    def stage_ct_finalize(self, s=None):
        """mean / std over the chunks, written in both orientations by one launch: (L, V) as the reference holds C(t) and
        (V, L) for the fits (stage_transpose is only needed behind a finalize that did not write them)"""
        s = s or self.slots[0]
        self.ctx.ct_finalize_dev(s.psum.data_ptr(), self.R, self.F, self.V, s.Ct.data_ptr(), s.dCt.data_ptr(),
                                 s.CtT.data_ptr(), s.dCtT.data_ptr())

    def stage_hist(self, s=None, soa=None):
        s = s or self.slots[0]
        soa = self.soa if soa is None else soa
        self.ctx.rotate_hist_dev(soa.data_ptr(), self.Npad, self.N, self.V, self.q, self.edges[0], self.edges[1],
                                 s.hist.data_ptr(), s.vecsum.data_ptr(), s.outer.data_ptr(), self.F)

    def stage_transpose(self, s=None):
        s = s or self.slots[0]
        self.ctx.transpose_dev(s.Ct.data_ptr(), self.L, self.V, s.CtT.data_ptr())
        self.ctx.transpose_dev(s.dCt.data_ptr(), self.L, self.V, s.dCtT.data_ptr())

    def stage_fit(self, s=None):
        """optimised_curve_fitting for every residue of the batch: one launch, results stay in HBM."""
        s = s or self.slots[0]
        self.ctx.order_search_dev(self.t_dev.data_ptr(), s.CtT.data_ptr(), s.dCtT.data_ptr(), self.V, self.L, self.listDoG,
                                  self.tau_guess.data_ptr(), 1, self.tau_max, self.chi_thr,
                                  s.popt.data_ptr(), s.dP.data_ptr(), s.chisq.data_ptr(), s.status.data_ptr(), s.nfev.data_ptr(),
                                  s.best.data_ptr(), s.S2.data_ptr(), s.C.data_ptr(), s.tau.data_ptr(), s.chi.data_ptr(),
                                  s.K.data_ptr(), work_ptr=None if s.fitwork is None else s.fitwork.data_ptr())

    def stage_relax(self, s=None):
        """R1/R2/NOE/rho from the selected models and the histogram, all operands resident in HBM."""
        s = s or self.slots[0]
        om, fdd, fcsa, tf, gr = self._relax_dev
        Kmax = max(self.listDoG) // 2
        E = len(self.fields)
        if self.aniso is None or self.aniso == 1.0:
            self.ctx.relax_dev(1, [self.Diso], E, om.data_ptr(), fdd.data_ptr(), fcsa.data_ptr(), tf.data_ptr(), gr.data_ptr(),
                               self.V, Kmax, self.zeta, s.S2.data_ptr(), s.C.data_ptr(), s.tau.data_ptr(), s.K.data_ptr(),
                               0, None, None, 0, s.relax.data_ptr())
        else:
            Dpar, Dperp = hm.symmtop_from_iso(self.Diso, self.aniso)
            self.ctx.relax_dev(2, [Dpar, Dperp], E, om.data_ptr(), fdd.data_ptr(), fcsa.data_ptr(), tf.data_ptr(), gr.data_ptr(),
                               self.V, Kmax, self.zeta, s.S2.data_ptr(), s.C.data_ptr(), s.tau.data_ptr(), s.K.data_ptr(),
                               self.nbins, self.binvecs_dev.data_ptr(), s.hist.data_ptr(), 0, s.relax.data_ptr())

    def stage_download(self, s=None):
        s = s or self.slots[0]
        self.ctx.memcpy_d2h_async(s._h_dres_addr, s.dres.data_ptr(), s.dres.numel() * 8)
        self.ctx.memcpy_d2h_async(s._h_ires_addr, s.ires.data_ptr(), s.ires.numel() * 4)

    # ---- batch-level API ----
    def front(self, vecs, k, events=None, pack_next=None):
        """Throughput half of batch k.  Serial form (depth 1): pack, C(t), histogram, transposes on the main stream.
        Overlapped form: C(t) + transposes on the main stream; the histogram of batch k and the pack of batch k+1
        (`pack_next`: its vectors) on the auxiliary stream, beside the C(t) launch, on alternating plane buffers.
        events: [before the C(t) kernel, after it (before its finalize), before histogram, after histogram, (4, 5: see
        back())]; the first two on the main stream."""
        s = self.slots[k % self.depth]
        if self.aux is None:
            self.ctx.set_stream(self.main.cuda_stream)
            with torch.cuda.stream(self.main):
                self.stage_pack(vecs)
                if events is not None:
                    events[0].record(self.main)
                self.stage_ct(s, mid_event=None if events is None else events[1])
                if events is not None:
                    events[2].record(self.main)
                self.stage_hist(s)
                if events is not None:
                    events[3].record(self.main)
                s.front_done = torch.cuda.Event()
                s.front_done.record(self.main)
                s.hist_done = s.front_done
            return s
        b = k % self.NB
        buf = self.soa_bufs[b]
        if not self._packed:
            # first batch of a run: nothing was packed ahead
            self.ctx.set_stream(self.aux.cuda_stream)
            with torch.cuda.stream(self.aux):
                if self._ct_done_ev[b] is not None:
                    self.aux.wait_event(self._ct_done_ev[b])
                self.stage_pack(vecs, buf)
                self._packed_ev[b] = torch.cuda.Event()
                self._packed_ev[b].record(self.aux)
        self._packed = False
        prev_done = s.done if s.busy else None       # the batch that used this slot `depth` batches ago, if still in flight
        main = self.main_alt if (k % 2 == 1 and self.main_alt is not None and self.tail_on_slot_stream) else self.main
        self.ctx.set_stream(main.cuda_stream)
        with torch.cuda.stream(main):
            main.wait_event(self._packed_ev[b])
            if s.busy and s.psum_free is not None:
                main.wait_event(s.psum_free)         # its raw sums are the only thing of that batch this kernel overwrites
            if events is not None:
                events[0].record(main)
            self.stage_ct(s, buf, mid_event=None if events is None else events[1], finalize=not self.tail_on_slot_stream)
            if not self.hist_on_aux:
                if prev_done is not None:
                    main.wait_event(prev_done)
                if s.guard is not None:
                    main.wait_event(s.guard)
                if events is not None:
                    events[2].record(main)
                self.stage_hist(s, buf)
                if events is not None:
                    events[3].record(main)
            self._ct_done_ev[b] = torch.cuda.Event()            # "the C(t) kernel is done with this plane buffer"
            self._ct_done_ev[b].record(main)
            s.front_done = torch.cuda.Event()
            s.front_done.record(main)
            s.hist_done = s.front_done
        self.ctx.set_stream(self.aux.cuda_stream)
        with torch.cuda.stream(self.aux):
            # the pack of batch k+1 first: the next C(t) launch waits for nothing else, while the histogram below may have
            # to wait for the batch that used this slot before (its relaxation kernel reads the slot's histogram)
            if pack_next is not None:
                nb = (k + 1) % self.NB
                if self._ct_done_ev[nb] is not None:
                    self.aux.wait_event(self._ct_done_ev[nb])       # C(t) of batch k+1-NB has read that buffer (its histogram
                                                                    # ran earlier on this stream)
                self.stage_pack(pack_next, self.soa_bufs[nb])
                self._packed_ev[nb] = torch.cuda.Event()
                self._packed_ev[nb].record(self.aux)
                self._packed = True
            if self.hist_on_aux:
                if prev_done is not None:
                    self.aux.wait_event(prev_done)
                if s.guard is not None:
                    self.aux.wait_event(s.guard)
                if events is not None:
                    events[2].record(self.aux)
                self.stage_hist(s, buf)                 # ordered behind the pack of this buffer on the same stream
                if events is not None:
                    events[3].record(self.aux)
                s.hist_done = torch.cuda.Event()
                s.hist_done.record(self.aux)
        self.ctx.set_stream(self.main.cuda_stream)
        return s

    def back(self, k, events=None):
        """Latency half of batch k on the slot's own stream: model-order search, relaxation, copies to pinned memory.
        events (optional): [4] and [5] are recorded around the search kernel on the slot's stream."""
        s = self.slots[k % self.depth]
        if s.stream is not self.main:
            s.stream.wait_event(s.front_done)
            s.stream.wait_event(s.hist_done)
            if s.guard is not None:
                s.stream.wait_event(s.guard)         # a device-side reader of the previous batch's C(t) / table / histogram
                s.guard = None
        self.ctx.set_stream(s.stream.cuda_stream)
        with torch.cuda.stream(s.stream):
            if self.aux is not None and self.tail_on_slot_stream:
                self.stage_ct_finalize(s)
                s.psum_free = torch.cuda.Event()
                s.psum_free.record(s.stream)
            if events is not None and len(events) > 5:
                events[4].record(s.stream)
            self.stage_fit(s)
            if events is not None and len(events) > 5:
                events[5].record(s.stream)
            self.stage_relax(s)
            self.stage_download(s)
            s.done = torch.cuda.Event()
            s.done.record(s.stream)
        s.busy = True
        self.ctx.set_stream(self.main.cuda_stream)
        return s

    def collect(self, s):
        """Wait for a batch and return its results (copies: they stay valid when the slot is reused)."""
        s.done.synchronize()
        s.busy = False
        r = s.host_results()
        s.result = r
        s.relax_out = r['relax']
        self.relax_out = r['relax']
        self.fit_best = r['best']
        tried = r['status'] != -100
        self.nfev_total += int(r['nfev'][tried].sum())
        self.nfev_last = {nP: r['nfev'][j][tried[j]] for j, nP in enumerate(self.listDoG)}
        return r

    def run(self, vecs, nb, events=None, on_finished=None, on_enqueued=None):
        """nb batches, up to `depth` in flight.  on_finished(slot) is called on the host for every finished batch, in order
        (results collected).  on_enqueued(slot) is called right after a batch's last launch: a consumer that works on the
        device (e.g. the all-gather of the results) queues itself behind slot.done there and returns an event, which the
        pipeline waits for before it overwrites that slot's result buffers.  (With batches overlapping, the throughput half
        of batch k + depth is already queued when on_finished(slot) runs for batch k: the host copies -- slot.result, and
        C(t) / dC(t) in HBM -- are still batch k's, the slot's histogram in HBM may already be the next batch's.)"""
        D = self.depth
        for k in range(nb):
            s = self.slots[k % D]
            run_ahead = self.aux is not None and self.tail_on_slot_stream
            if s.busy and not run_ahead:
                self.collect(s)
                if on_finished is not None:
                    on_finished(s)
            # Overlapped form: the throughput half of batch k is queued BEFORE the host waits for the batch that used the
            # slot `depth` batches ago -- C(t) only overwrites that batch's raw sums (free once its chunk statistics ran),
            # the histogram waits for it on the device -- so the main stream always has its next launch queued.
            self.front(vecs, k, None if events is None else events[k], pack_next=vecs if k + 1 < nb else None)
            if s.busy:
                self.collect(s)
                if on_finished is not None:
                    on_finished(s)
            self.back(k, None if events is None else events[k])
            if on_enqueued is not None:
                s.guard = on_enqueued(s)
        for k in range(max(0, nb - D), nb):
            s = self.slots[k % D]
            if s.busy:
                self.collect(s)
                if on_finished is not None:
                    on_finished(s)
        self.ctx.set_stream(self.main.cuda_stream)

    def prime(self, vecs):
        """Set-up: push one batch through every slot so that code objects are loaded, the slots' device and pinned
        buffers are resident and every stream has seen a launch before the first batch that counts."""
        self.run(vecs, self.depth)
        torch.cuda.synchronize(self.dev)
        self.nfev_total = 0

    def step(self, vecs):
        """One batch from vectors to R1/R2/NOE; returns the (E, V, 4, 2) table."""
        self.front(vecs, 0)
        s = self.back(0)
        return self.collect(s)['relax']

    def selected_params(self, s=None):
        """S2, C, tau (sorted by tau), number of components and chi^2 of the selected models of a collected batch."""
        s = s or self.slots[0]
        r = s.result
        return r['S2'], r['C'], r['tau'], r['K'], r['chi']

    # convenience views of slot 0 (serial use)
    @property
    def Ct(self):
        return self.slots[0].Ct

    @property
    def dCt(self):
        return self.slots[0].dCt

    @property
    def hist(self):
        return self.slots[0].hist


# ======================================================================================================================
# Grouped schedule: throughput halves of G batches back to back, then ONE merged launch for their latency halves
# ======================================================================================================================
class _Lease:
    """one hand-out of a recycled host buffer: exposes the memory through the array interface, so that numpy arrays made
    from it (np.asarray) and all their views hold a reference to THIS object; when the last of them is gone the lease is
    collected and _Group._host_copy may reuse the buffer"""
    __slots__ = ('__array_interface__', '_keep', '__weakref__')

    def __init__(self, arr):
        self._keep = arr
        self.__array_interface__ = dict(arr.__array_interface__)


class _BatchView:
    """What the stage functions need of one batch inside a group (views into the group's contiguous buffers)."""
    __slots__ = ('Ct', 'dCt', 'CtT', 'dCtT', 'hist', 'vecsum', 'outer', 'psum', 'result', 'relax_out', 'index')


class _Group:
    """Device buffers, pinned mirrors and stream of a GROUP of up to G batches.  Inputs of the merged launches are
    contiguous over the group's residues (batch j = rows j V .. (j + 1) V), results are laid out for the g batches a
    launch really holds (the residue axis has length g V)."""

    def __init__(self, ctx, dev, G, V, L, R, nbins, nO, Pmax, E, stream):
        f64 = dict(device=dev, dtype=torch.float64)
        i32 = dict(device=dev, dtype=torch.int32)
        self.G, self.V, self.L = G, V, L
        self._shape = (nO, Pmax, Pmax // 2, E)
        GV = G * V
        self._Ct = torch.empty((G, L, V), **f64)
        self._dCt = torch.empty((G, L, V), **f64)
        self.CtT = torch.empty((GV, L), **f64)
        self.dCtT = torch.empty((GV, L), **f64)
        self._hist = torch.empty((GV, nbins), **f64)
        self.vecsum = torch.empty((GV, 3), **f64)
        self.outer = torch.empty((G, R, V, 6), **f64)
        self.fitwork = torch.empty((GV, L), **f64)
        nd, ni = self._sizes(G)
        self.dres = torch.empty((nd,), **f64)
        self.ires = torch.empty((ni,), **i32)
        self._ctx = ctx
        self.h_dres, self._h_dres_addr = _pinned_array(ctx, nd, np.float64)
        self.h_ires, self._h_ires_addr = _pinned_array(ctx, ni, np.int32)
        # dispatch order of the merged launch: pinned host copy + device copy (refreshed per launch, stream-ordered before it)
        self.h_order, self._h_order_addr = _pinned_array(ctx, GV, np.int32)
        self.order_dev = torch.empty((GV,), **i32)
        self.stream = stream
        self.g = 0                 # batches of the group in flight / last collected
        self.first = 0             # index (in the run) of the group's first batch
        self.done = None
        self.guard = None
        self.busy = False
        self.batches = []
        self._views = {}
        self._dpool, self._ipool = [], []

    def _layouts(self, g):
        nO, Pmax, Kmax, E = self._shape
        n = g * self.V
        return ((('popt', (nO, n, Pmax), 1), ('dP', (nO, n, Pmax), 1), ('chisq', (nO, n), 1), ('S2', (n,), 0), ('chi', (n,), 0),
                 ('C', (n, Kmax), 0), ('tau', (n, Kmax), 0), ('relax', (E, n, 4, 2), 1)),
                (('status', (nO, n), 1), ('nfev', (nO, n), 1), ('best', (n,), 0), ('K', (n,), 0)))

    def _sizes(self, g):
        dl, il = self._layouts(g)
        return sum(int(np.prod(sh)) for _, sh, _ in dl), sum(int(np.prod(sh)) for _, sh, _ in il)

    def views(self, g):
        """device views of the result buffers for a launch over g batches"""
        if g not in self._views:
            out = {}
            for buf, layout in zip((self.dres, self.ires), self._layouts(g)):
                o = 0
                for name, sh, _ in layout:
                    n = int(np.prod(sh))
                    out[name] = buf[o:o + n].view(sh)
                    o += n
            self._views[g] = out
        return self._views[g]

    # what a device-side consumer of a finished group reads (bench.py's all-gather): the g batches in flight
    @property
    def Ct(self):
        return self._Ct[:self.g]

    @property
    def dCt(self):
        return self._dCt[:self.g]

    @property
    def hist(self):
        return self._hist[:self.g * self.V]

    @property
    def relax(self):
        return self.views(self.g)['relax']

    def batch(self, j):
        V = self.V
        b = _BatchView()
        b.Ct, b.dCt = self._Ct[j], self._dCt[j]
        b.CtT, b.dCtT = self.CtT[j * V:(j + 1) * V], self.dCtT[j * V:(j + 1) * V]
        b.hist, b.vecsum, b.outer = self._hist[j * V:(j + 1) * V], self.vecsum[j * V:(j + 1) * V], self.outer[j]
        b.psum = None
        b.result = b.relax_out = None
        b.index = j
        return b

    def _host_copy(self, pool, mirror, n):
        """the first n elements of a pinned mirror in pageable memory the caller may keep.  The buffers are recycled, with
        explicit ownership: every hand-out goes through a fresh _Lease object that the returned array (and every view cut
        from it) keeps alive as its base; a pool buffer is taken again only when the lease of its last hand-out is gone
        (a dead weak reference) -- no interpreter-specific reference counts.  Results stay valid for as long as anybody holds
        them, and a collect does not start with a 10 MB allocation and its page faults (1.5 ms when the allocator has just
        seen another size).  Where leases die late (no reference counting) the pool simply is not reused."""
        entry = None
        for e in pool:
            if e[1] is None or e[1]() is None:
                entry = e
                break
        if entry is None:
            entry = [np.empty(mirror.size, dtype=mirror.dtype), None]
            if len(pool) < 4:
                pool.append(entry)
        buf = entry[0]
        np.copyto(buf[:n], mirror[:n])     # 10 MB for a 20-batch group: 0.46 ms (a threaded copy was measured: no faster)
        lease = _Lease(buf[:n])
        entry[1] = weakref.ref(lease)
        return np.asarray(lease)

    def host_results(self):
        """per batch: COPIES of the pinned mirrors, split along the residue axis"""
        g, V = self.g, self.V
        nd, ni = self._sizes(g)
        whole = {}
        for buf, layout in zip((self._host_copy(self._dpool, self.h_dres, nd), self._host_copy(self._ipool, self.h_ires, ni)),
                               self._layouts(g)):
            o = 0
            for name, sh, axis in layout:
                n = int(np.prod(sh))
                whole[name] = (buf[o:o + n].reshape(sh), axis)
                o += n
        out = []
        for j in range(g):
            r = {}
            for name, (arr, axis) in whole.items():
                r[name] = arr[j * V:(j + 1) * V] if axis == 0 else arr[:, j * V:(j + 1) * V]
            out.append(r)
        return out

    def release(self):
        for name in ('_h_dres_addr', '_h_ires_addr', '_h_order_addr'):
            addr = getattr(self, name, None)
            if addr:
                self._ctx.host_free(addr)
                setattr(self, name, None)
        self.h_dres = self.h_ires = self.h_order = None


class GroupedPipeline(DevicePipeline):
    """The same stages as DevicePipeline, scheduled for throughput over MANY batches (trajectory shards):

      phase 1   pack, C(t), histogram, chunk statistics of `group` batches back to back (two C(t) streams, the bandwidth
                kernels beside them) -- no fit is in flight, so the C(t) grids have the chip's registers to themselves
                and the pack / histogram waves find slots beside them;
      phase 2   ONE model-order search over the group's g x V residues (sr_expfit_order_search_batched_f64_dev), one
                relaxation launch, one pair of copies to pinned memory; with `late_hist` (an option; the default of rounds 3-4)
                the group's histograms run HERE, in the tail of the merged launch: its last workgroup releases a signal the
                histogram stream waits for (every residue has a CU by then; only the longest fits are still running and most
                slots are free), so that phase 1 is C(t) + pack + chunk statistics only.  The planes of group + 3 batches stay
                alive for that.  It pays with the pseudo-random dispatch order (2.45 against 2.55 ms per step at 20 steps), whose
                launch has an idle straggler tail; dispatched longest first (the default) the launch ends with its bulk, the
                histograms would run behind it with the chip to themselves, and beside the C(t) kernels they cost less: 1.77
                against 1.82 ms per step at 20 steps, 1.60 against 1.635 in steady state -- hence off by default.

    Why: a fit launch lasts as long as its slowest residue (one nine-parameter fit that never converges: ~300 evaluations,
    6.5 ms) while the median residue needs 0.3 ms.  Launched per batch, the stragglers of ~3 batches are always in flight
    and their workgroups (256 VGPRs) time-share the CUs with the C(t) grids: measured 2.28 ms per batch in steady state
    against 1.15 ms (phase 1 alone) + 0.71 ms (fits with the chip to themselves).  Merged over a group, the stragglers of
    all its batches run side by side while the cheap residues fill the rest of the chip.  The residues of the merged
    launch are dispatched LONGEST FIRST by prediction (`dispatch = 'history'`, the default): a residue's cost is taken to be
    the number of model evaluations its fits needed in the most recent batch the pipeline has collected -- the shards of a
    stream are consecutive pieces of one protein's trajectory, the residue whose nine-parameter fit crawls along a flat valley
    in one shard is the likely straggler of the next -- and ties keep a fixed pseudo-random order (residues that are
    expensive for the same reason sit at the same index in every batch, and consecutive workgroup indices are served by the
    same part of the chip).  A launch lasts as long as its slowest residue: started first, the 14 ms fit ends with the bulk
    instead of 14 ms after its random place in it.  A wrong prediction costs what the random order costs; results are those
    of DevicePipeline bit for bit either way (a residue's fit does not depend on what else is in the launch).  `dispatch =
    'random'`: the pseudo-random order alone (rounds 3-4); natural order: 32 ms for 20 batches, random 21.5 ms (DESIGN.md
    section 5).  NOTE for readers of the benchmark: bench.py feeds the same shard every step, so there the prediction is exact;
    it reports the random-order figure beside the headline.

    The next group's phase 1 is queued behind the merged launch without waiting for it (`overlap`), so the tail of one
    group's stragglers is covered by the next group's C(t) kernels; two group buffers alternate.  With `late_hist` that
    overlap is bounded by the plane buffers: the next group's packs wait for the previous group's histograms (which run in
    the merged launch's tail) once the 3 spare plane buffers are taken, i.e. about three batches of the next group start
    before the previous merged launch reaches its tail -- the overlap covers the tail, not the bulk.  Memory: group + 3
    plane buffers (12 B per frame and vector each; 21 GB for 32 cfg3 batches) + two group buffers (1.7 GB each); checked
    against the free device memory at construction (falls back to late_hist off, see `late_hist_note`)."""

    def __init__(self, ctx, device, frames, V, R, F, dt, group=32, overlap=True, psum_buffers=3, late_hist=False, **kw):
        kw = dict(kw)
        kw.setdefault('pack_cus', 128)                    # the pack stream on half of the CUs (DevicePipeline.__init__)
        kw['depth'] = max(2, int(psum_buffers))          # the base class's slots: only their raw-sum buffers are used (a rotating pool)
        self.late_hist = bool(late_hist)
        self.late_hist_note = None
        if self.late_hist:
            # a group's planes stay alive until its histograms ran (in phase 2): group + 3 plane buffers of 12 B per (frame,
            # vector) each -- 35 x 0.6 GB = 21 GB for groups of 32 cfg3 batches.  When that is more than half of the free
            # device memory the histograms go back beside the C(t) kernels (late_hist off) instead of failing later in an
            # allocation; `late_hist_note` says so.
            need = (max(1, int(group)) + 3) * 12 * int(frames) * int(V)
            free = torch.cuda.mem_get_info(device)[0]
            if need > free // 2:
                self.late_hist = False
                self.late_hist_note = ('late_hist off: %d plane buffers need %.1f GB, %.1f GB of device memory are free'
                                       % (max(1, int(group)) + 3, need / 1e9, free / 1e9))
        if self.late_hist:
            kw['plane_buffers'] = max(1, int(group)) + 3
        else:
            # the histograms then run on the auxiliary stream beside the C(t) kernels: a CU mask on that stream (meant for the pack
            # alone) would confine them too, which measured slower -- no mask in this configuration
            kw['pack_cus'] = 0
        for name in ('reserve_cus', 'aux_cus'):
            if kw.get(name):
                raise ValueError('GroupedPipeline runs its phases on the whole chip: %s is not supported' % name)
        super().__init__(ctx, device, frames, V, R, F, dt, **kw)
        self.group = max(1, int(group))
        self.overlap = bool(overlap)
        self.permute = True            # False: dispatch the merged launch's residues in natural order
        self.dispatch = 'history'      # 'history': longest first by the evaluation counts of the last collected batch; 'random'
        self._cost = None              # (V,) evaluations per residue (all orders) of the most recent collected batch
        self._cost_version = 0
        self.dev_skip_fits = False     # development only
        import os as _os
        self.dev_skip_hist = bool(_os.environ.get('SR_DEV_SKIP_HIST'))      # development only: marginal cost of the histogram in phase 1
        self.pool = self.slots
        self.NP = len(self.pool)
        self._psum_free = [None] * self.NP
        self.tail = torch.cuda.Stream(device=device)
        Pmax = max(self.listDoG)
        E = len(self.fields)
        self.groups = [_Group(ctx, device, self.group, V, self.L, R, self.nbins, len(self.listDoG), Pmax, E,
                              torch.cuda.Stream(device=device)) for _ in range(2)]
        self.hist_stream = torch.cuda.Stream(device=device) if self.late_hist else None      # (confined to 64 / 128 / 192 CUs: slower, 2.40-2.76 ms per step)
        self._late = []
        self._hist_done_ev = [None] * self.NB           # late histograms: "the histogram that read this plane buffer has run"
        from .hip import SpinRelaxHipError
        for grp in self.groups:                           # every group has both attributes whatever the allocation does
            grp.signal, grp.epoch = None, 0
        try:
            for grp in self.groups:
                grp.signal = ctx.signal_alloc()
        except SpinRelaxHipError:
            # a device / runtime without stream waits on memory values: the histograms run beside the C(t) kernels instead
            for grp in self.groups:
                if grp.signal:                        # the first allocation may have succeeded
                    ctx.signal_free(grp.signal)
                grp.signal = None
            if self.late_hist:
                for grp in self.groups:               # pinned mirrors and streams of the group buffers
                    grp.stream = None
                    grp.release()
                self.slots = self.pool
                super().close()
                raise SpinRelaxHipError('late_hist needs sr_signal_alloc (hipStreamWaitValue32 on signal memory); construct the '
                                        'pipeline with late_hist=False on this device')
        # gate_next: hold the next group's C(t) launches back (same signal) until the merged launch begins to drain.  Measured, off:
        # 2.13 against 2.115 ms per step in steady state (queued behind a launch that fills the chip they hardly get a slot before
        # that moment anyway), and no help to unequal splits of a 20-batch run (16 + 4, 14 + 6, ...: 49-51 ms like one group of
        # 20) -- when the last workgroup starts, the 512 resident ones are the long-lived ones and free their slots over
        # milliseconds, so C(t) grids released at that moment run at a fraction of their speed (4 launches: 9 ms).
        self.gate_next = False
        self._prev_signal = None
        self.slots = self.groups                          # what a caller iterates over to set up per-slot consumers
        self._perm = {}
        self._fcsa = {}
        self._on_part = None
        self._last_fin = None
        self._last_hist = None
        self._prev_done = None
        self._nbatch = 0
        torch.cuda.synchronize(device)

    def group_sizes(self, nb):
        """how a run of nb batches is cut into groups (at most `group` each).  `sizes_override` (development): explicit list."""
        ov = getattr(self, 'sizes_override', None)
        if ov:
            out, rem = [], nb
            for g in ov:
                if rem <= 0:
                    break
                g = min(int(g), rem, self.group)
                out.append(g)
                rem -= g
            while rem > 0:
                g = min(self.group, rem)
                out.append(g)
                rem -= g
            return out
        out, rem = [], nb
        while rem > 0:
            g = min(self.group, rem)
            out.append(g)
            rem -= g
        return out

    def _dispatch_order(self, g):
        """order in which the workgroups of a merged launch take the g V residues (host int32 array): a fixed pseudo-random
        permutation, stably re-sorted longest first by the predicted cost when there is a history (see the class comment)"""
        hist = self.dispatch == 'history' and self._cost is not None
        key = (g, self._cost_version if hist else -1)
        if key not in self._perm:
            n = g * self.V
            p = np.random.RandomState(20240 + g).permutation(n).astype(np.int32) if g > 1 else np.arange(n, dtype=np.int32)
            if hist:
                p = p[np.argsort(-self._cost[p % self.V], kind='stable')]
            self._perm = {k: v for k, v in self._perm.items() if k[1] == -1}       # older histories are dead
            self._perm[key] = np.ascontiguousarray(p, dtype=np.int32)
        return self._perm[key]

    def _fcsa_for(self, g):
        if g not in self._fcsa:
            self._fcsa[g] = self._relax_dev[2].repeat(1, g).contiguous() if self._relax_dev[2].dim() == 2 else self._relax_dev[2].repeat(g).contiguous()
        return self._fcsa[g]

    def _pack_from(self, vecs, kk, buf):
        """pack batch kk's vectors into plane buffer `buf` on the auxiliary stream.  `vecs`: a device tensor (frames, Vtot, 3), or a
        FEED -- an object with acquire(kk, stream) -> object with data_ptr() / shape of that batch's device array (the feed makes
        `stream` wait until the frames are there) and release(kk, stream) (called once the pack has been queued: the feed may
        refill the buffer when `stream` has passed that point).  bench.py's PinnedFeed streams every batch from host memory."""
        if hasattr(vecs, 'acquire'):
            src = vecs.acquire(kk, self.aux)
            self.ctx.set_stream(self.aux.cuda_stream)          # the feed drives the context from its own stream
            self.stage_pack(src, buf)
            vecs.release(kk, self.aux)
        else:
            self.stage_pack(vecs, buf)

    def _front_grouped(self, vecs, kk, grp, j, events, pack_next):
        bv = grp.batches[j]
        b = kk % self.NB
        buf = self.soa_bufs[b]
        pi = kk % self.NP
        bv.psum = self.pool[pi].psum
        if not self._packed:
            self.ctx.set_stream(self.aux.cuda_stream)
            with torch.cuda.stream(self.aux):
                if self._ct_done_ev[b] is not None:
                    self.aux.wait_event(self._ct_done_ev[b])
                if self._hist_done_ev[b] is not None:
                    self.aux.wait_event(self._hist_done_ev[b])
                self._pack_from(vecs, kk, buf)
                self._packed_ev[b] = torch.cuda.Event()
                self._packed_ev[b].record(self.aux)
        self._packed = False
        main = self.main_alt if (kk % 2 == 1 and self.main_alt is not None) else self.main
        self.ctx.set_stream(main.cuda_stream)
        with torch.cuda.stream(main):
            main.wait_event(self._packed_ev[b])
            if self._psum_free[pi] is not None:
                main.wait_event(self._psum_free[pi])      # the chunk statistics of batch kk - NP have read these raw sums
            if j < 2 and not self.overlap and self._prev_done is not None:
                main.wait_event(self._prev_done)          # strict phases: the previous group's merged launch has finished (both C(t) streams)
            elif j < 2 and self.gate_next and self._prev_signal is not None:
                self.ctx.stream_wait_signal(*self._prev_signal)

            if events is not None:
                events[0].record(main)
            # (Offsetting the two C(t) streams by half a launch -- first launch of a run in two halves, the second stream waiting
            # for the first half -- was measured: the launches then alternate perfectly, 1.25 ms apart, and the group's C(t) phase
            # takes exactly as long as in lockstep: the phase is bound by the kernels' own time, not by coinciding tails.)
            self.ctx.ct_sums_dev(buf.data_ptr(), self.Npad, self.R, self.F, self.V, bv.psum.data_ptr())
            if events is not None:
                events[1].record(main)
            ct_ev = torch.cuda.Event()
            ct_ev.record(main)
            self._ct_done_ev[b] = ct_ev
        tail = self.tail          # (chunk statistics spread over 3 or 5 streams: measured, no difference)
        self.ctx.set_stream(tail.cuda_stream)
        with torch.cuda.stream(tail):
            tail.wait_event(ct_ev)
            if j == 0 and grp.guard is not None:
                tail.wait_event(grp.guard)                # a device-side reader of the group's previous C(t)
            self.ctx.ct_finalize_dev(bv.psum.data_ptr(), self.R, self.F, self.V, bv.Ct.data_ptr(), bv.dCt.data_ptr(),
                                     bv.CtT.data_ptr(), bv.dCtT.data_ptr())
            ev = torch.cuda.Event()
            ev.record(tail)
            self._psum_free[pi] = ev
            self._last_fin = ev
            if self._on_part is not None:
                self._on_part('ct', bv, ev)          # C(t), dC(t) of this batch are final once `ev` has passed
        self.ctx.set_stream(self.aux.cuda_stream)
        with torch.cuda.stream(self.aux):
            if pack_next is not None:
                nb = (kk + 1) % self.NB
                if self._ct_done_ev[nb] is not None:
                    self.aux.wait_event(self._ct_done_ev[nb])
                if self._hist_done_ev[nb] is not None:
                    self.aux.wait_event(self._hist_done_ev[nb])
                self._pack_from(pack_next, kk + 1, self.soa_bufs[nb])
                self._packed_ev[nb] = torch.cuda.Event()
                self._packed_ev[nb].record(self.aux)
                self._packed = True
            if j == 0 and grp.guard is not None:
                self.aux.wait_event(grp.guard)
            if events is not None and not self.late_hist:
                events[2].record(self.aux)
            if self.late_hist:
                self._late.append((bv, buf, b, events))
            elif not self.dev_skip_hist:
                self.stage_hist(bv, buf)
            if events is not None and not self.late_hist:
                events[3].record(self.aux)
            ev = torch.cuda.Event()
            ev.record(self.aux)
            self._last_hist = ev
            if self._on_part is not None and not self.late_hist:
                self._on_part('hist', bv, ev)
        self.ctx.set_stream(self.main.cuda_stream)

    def _back_grouped(self, grp, g, events):
        st = grp.stream
        st.wait_event(self._last_fin)        # the tail and auxiliary streams run in order: their last events cover the group
        st.wait_event(self._last_hist)
        grp.guard_hist = grp.guard if self.late_hist else None
        grp.guard = None
        v = grp.views(g)
        n = g * self.V
        self.ctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            order_ptr = None
            if self.permute and not self.dev_skip_fits:
                # the group's previous launch is over (collect() ran before the buffer was taken again): its order buffers are free
                grp.h_order[:n] = self._dispatch_order(g)
                self.ctx.memcpy_h2d_async(grp.order_dev.data_ptr(), grp._h_order_addr, n * 4)
                order_ptr = grp.order_dev.data_ptr()
            if events is not None and len(events) > 5:
                events[4].record(st)
            if not self.dev_skip_fits:
                self.ctx.order_search_batched_dev(self.t_dev.data_ptr(), 1, grp.CtT.data_ptr(), grp.dCtT.data_ptr(), n, self.L, self.listDoG,
                                                  self.tau_guess.data_ptr(), 1, self.tau_max, self.chi_thr,
                                                  v['popt'].data_ptr(), v['dP'].data_ptr(), v['chisq'].data_ptr(), v['status'].data_ptr(),
                                                  v['nfev'].data_ptr(), v['best'].data_ptr(), v['S2'].data_ptr(), v['C'].data_ptr(),
                                                  v['tau'].data_ptr(), v['chi'].data_ptr(), v['K'].data_ptr(),
                                                  work_ptr=grp.fitwork.data_ptr(),
                                                  dispatch_order_ptr=order_ptr,
                                                  tail_signal=grp.signal, tail_value=grp.epoch + 1 if grp.signal else 0)
            if events is not None and len(events) > 5:
                events[5].record(st)
            if grp.signal:
                grp.epoch += 1
                self.ctx.stream_write_signal(grp.signal, grp.epoch)
                self._prev_signal = (grp.signal, grp.epoch)
            if self.late_hist:
                # The group's histograms fill the TAIL of the merged launch.  Its last workgroup releases the signal when it
                # starts (every residue has a CU by then; from here on slots only free up while the longest fits finish); the
                # histogram stream has been waiting for exactly that.  The write behind the launch releases it at the latest
                # when the launch is over (skipped fits, a launch that fits the chip at once).
                hs = self.hist_stream
                hs.wait_event(self._last_hist)            # every pack of the group has run (auxiliary stream, in order)
                if grp.guard_hist is not None:
                    hs.wait_event(grp.guard_hist)
                    grp.guard_hist = None
                self.ctx.set_stream(hs.cuda_stream)
                with torch.cuda.stream(hs):
                    self.ctx.stream_wait_signal(grp.signal, grp.epoch)
                    for bv, buf, b, evs in self._late:
                        if evs is not None:
                            evs[2].record(hs)
                        self.stage_hist(bv, buf)
                        if evs is not None:
                            evs[3].record(hs)
                        ev = torch.cuda.Event()
                        ev.record(hs)
                        self._hist_done_ev[b] = ev
                        if self._on_part is not None:
                            self._on_part('hist', bv, ev)
                    hev = torch.cuda.Event()
                    hev.record(hs)
                self._late = []
                self.ctx.set_stream(st.cuda_stream)
                st.wait_event(hev)
            om, fdd, _, tf, gr = self._relax_dev
            fcsa = self._fcsa_for(g)
            Kmax = max(self.listDoG) // 2
            E = len(self.fields)
            if self.aniso is None or self.aniso == 1.0:
                self.ctx.relax_dev(1, [self.Diso], E, om.data_ptr(), fdd.data_ptr(), fcsa.data_ptr(), tf.data_ptr(), gr.data_ptr(),
                                   n, Kmax, self.zeta, v['S2'].data_ptr(), v['C'].data_ptr(), v['tau'].data_ptr(), v['K'].data_ptr(),
                                   0, None, None, 0, v['relax'].data_ptr())
            else:
                Dpar, Dperp = hm.symmtop_from_iso(self.Diso, self.aniso)
                self.ctx.relax_dev(2, [Dpar, Dperp], E, om.data_ptr(), fdd.data_ptr(), fcsa.data_ptr(), tf.data_ptr(), gr.data_ptr(),
                                   n, Kmax, self.zeta, v['S2'].data_ptr(), v['C'].data_ptr(), v['tau'].data_ptr(), v['K'].data_ptr(),
                                   self.nbins, self.binvecs_dev.data_ptr(), grp._hist.data_ptr(), 0, v['relax'].data_ptr())
            nd, ni = grp._sizes(g)
            self.ctx.memcpy_d2h_async(grp._h_dres_addr, grp.dres.data_ptr(), nd * 8)
            self.ctx.memcpy_d2h_async(grp._h_ires_addr, grp.ires.data_ptr(), ni * 4)
            grp.done = torch.cuda.Event()
            grp.done.record(st)
        self._prev_done = grp.done
        grp.busy = True
        self.ctx.set_stream(self.main.cuda_stream)

    def collect(self, grp, on_finished=None):
        grp.done.synchronize()
        grp.busy = False
        res = grp.host_results()
        for j, r in enumerate(res):
            bv = grp.batches[j]
            bv.result = r
            bv.relax_out = r['relax']
            self.relax_out = r['relax']
            self.fit_best = r['best']
            tried = r['status'] != -100
            self.nfev_total += int(r['nfev'][tried].sum())
            self.nfev_last = {nP: r['nfev'][i][tried[i]] for i, nP in enumerate(self.listDoG)}
            if on_finished is not None:
                on_finished(bv)
        if res:                                # the cost prediction of the following launches: evaluations per residue, last batch
            r = res[-1]
            self._cost = np.where(r['status'] != -100, r['nfev'], 0).sum(axis=0).astype(np.int64)
            self._cost_version += 1
        return res

    def run(self, vecs, nb, events=None, on_finished=None, on_enqueued=None, on_part=None):
        """nb batches in groups of at most `group`.  on_finished(batch view) per finished batch, in order (batch.result holds its
        host results, batch.Ct / dCt / hist its device arrays until the group buffer is reused two groups later);
        on_enqueued(group) right after a group's last launch: a device-side consumer queues itself behind group.done, reads
        group.Ct / dCt / hist / relax (the g batches in flight) and returns an event the pipeline waits for before it
        overwrites them.  on_part(kind, batch view, event): called when the launch that finishes one batch's C(t) / dC(t)
        (kind 'ct') or histogram ('hist') has been queued; `event` marks its completion -- a device-side consumer can move the
        bulky per-batch arrays while the group is still being computed and leave only the small table for on_enqueued (whose
        returned event must cover everything the consumer read of the group).  events[k] as in DevicePipeline.run; [4], [5] are
        recorded around the merged launch on the entry of the group's FIRST batch."""
        self._on_part = on_part
        k, gi = 0, self._nbatch
        pending = []
        sizes = list(self.group_sizes(nb))
        while k < nb:
            g = sizes.pop(0)
            grp = self.groups[gi % 2]
            if grp.busy:
                pending.remove(grp)
                self.collect(grp, on_finished)
            grp.g, grp.first = g, k
            grp.batches = [grp.batch(j) for j in range(g)]
            for j in range(g):
                self._front_grouped(vecs, k + j, grp, j, None if events is None else events[k + j],
                                    vecs if k + j + 1 < nb else None)
            self._back_grouped(grp, g, None if events is None else events[k])
            if on_enqueued is not None:
                grp.guard = on_enqueued(grp)
            pending.append(grp)
            k += g
            gi += 1
        for grp in pending:
            self.collect(grp, on_finished)
        self._on_part = None
        self._nbatch = gi
        self.ctx.set_stream(self.main.cuda_stream)

    def prime(self, vecs):
        """set-up: both group buffers, every stream and code object see a (small) group before the first batch that counts"""
        for _ in range(2):
            self.run(vecs, min(self.group, 4))
        torch.cuda.synchronize(self.dev)
        self.nfev_total = 0

    def step(self, vecs):
        self.run(vecs, 1)
        return self.relax_out

    def close(self):
        if getattr(self, '_closed', False):
            return
        torch.cuda.synchronize(self.dev)
        self.ctx.device_sync()
        for grp in self.groups:
            grp.done = grp.guard = None
            grp.stream = None
            grp.batches = []
            if getattr(grp, 'signal', None):
                self.ctx.signal_free(grp.signal)
                grp.signal = None
            grp.release()
        self.hist_stream = None
        self._hist_done_ev = []
        self._psum_free = [None] * self.NP
        self._last_fin = self._last_hist = self._prev_done = None
        self.tail = None
        self.slots = self.pool
        super().close()
