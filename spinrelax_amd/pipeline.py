"""
Device-resident single-process pipeline over one shard of vectors:

    vectors (N, V, 3) float32 in HBM
      -> kernel 0  pack into per-vector planes
      -> kernel 1  C(t), dC(t)                     (calculate-Ct-from-traj.py:200-238)
      -> kernel 2  rotation + Lambert histogram + mean vector + S2 sums   (:541-646)
      -> kernel 3b multi-exponential fits, model orders 2,3,5,7,9 with the reference's accept/reject
                   sequence                         (fitting_Ct_functions.py:278-345)
      -> kernel 3a J(omega), R1/R2/NOE/rho with the histogram as weights  (calculate-relaxations-from-Ct.py:125-191)

This is what run-all.bash's Step 3 + Step 4 compute (run-all.bash:476-533) without the text files in
between; bench.py times it.  torch is used for device memory, streams and events only; every computation
goes through the C ABI.

Batches and overlap.  A GPU works through its vectors in batches (512 vectors in the benchmark).  The
fits are a latency chain: the last model order (9 parameters) runs on the few residues that got that far
and its wall time is set by ONE straggler (818 of the 900 allowed evaluations in the benchmark data,
median 32), on one CU, while the rest of the chip idles.  `begin(batch)` therefore enqueues that last
order on a side stream and returns; `finish(batch)` (called `depth - 1` batches later) collects it, applies
the accept/reject rule and runs the relaxation kernel.  All per-batch device buffers exist `depth` times.
`depth = 1` is the plain serial pipeline (`step()`).

Software pipeline (`run()`), three stages per batch on three kinds of streams:
    front(k)  main stream : pack, C(t), histogram, transposes              (throughput-bound, fills the chip)
    fits(k)   fit stream  : model orders 2..7 with the host in the loop    (latency-bound, few CUs)
              side stream : last order, asynchronous
    finish(k) fit stream  : collect the last order, select, relaxation kernel
In iteration k the host enqueues front(k+1) first, then drives fits(k) while the C(t) kernel of batch k+1 keeps
the chip (and its clock: an MI355X that idles for 5 ms runs the next C(t) launch ~17 % slower) busy, then
finishes batch k-(depth-2).
"""
import numpy as np
import torch

from . import ct as hostct
from . import fitting_Ct_functions as fitCt
from . import _hostmath as hm
from . import spectral_densities as sd


class _Slot:
    """Device buffers of one in-flight batch."""

    def __init__(self, dev, V, L, R, nbins, Pmax):
        f64 = dict(device=dev, dtype=torch.float64)
        self.Ct = torch.empty((L, V), **f64)
        self.dCt = torch.empty((L, V), **f64)
        self.CtT = torch.empty((V, L), **f64)
        self.dCtT = torch.empty((V, L), **f64)
        self.hist = torch.empty((V, nbins), **f64)
        self.vecsum = torch.empty((V, 3), **f64)
        self.outer = torch.empty((R, V, 6), **f64)
        self.p0 = torch.empty((V, Pmax), **f64)
        self.popt = torch.empty((V, Pmax), **f64)
        self.pcov = torch.empty((V, Pmax * Pmax), **f64)
        self.chi = torch.empty((V,), **f64)
        self.status = torch.empty((V,), device=dev, dtype=torch.int32)
        self.nfev = torch.empty((V,), device=dev, dtype=torch.int32)
        self.skip = torch.zeros((V,), device=dev, dtype=torch.uint8)
        self.fitwork = torch.empty((V, 2, L), **f64)
        self.search = None
        self.front_done = None
        self.pending = None          # (request, event) of the asynchronous last order
        self.relax_out = None


class DevicePipeline:
    def __init__(self, ctx, device, frames, V, R, F, dt, q_rot=None, Diso=None, aniso=None, field_MHz=(600.133,),
                 zeta=0.890023, histBinX=72, listDoG=(2, 3, 5, 7, 9), csa=None, depth=1, stream=None, reserve_cus=0,
                 fits_on_reserved_only=False):
        self.ctx = ctx
        self.dev = device
        self.frames, self.V, self.R, self.F, self.dt = frames, V, R, F, dt
        self.L = F // 2
        self.N = R * F
        self.Npad = (frames + 63) // 64 * 64
        self.q = None if q_rot is None else np.asarray(q_rot, dtype=np.float64)
        self.Diso, self.aniso = Diso, aniso
        self.fields = tuple(field_MHz)
        self.zeta = zeta
        self.listDoG = tuple(listDoG)
        self.edges = hostct.lambert_edges(histBinX)
        self.nbins = histBinX * int(histBinX / 2)
        self.depth = max(1, int(depth))
        self.main = stream if stream is not None else torch.cuda.current_stream(device)
        self._owned_streams = []
        resv_words = None
        self.reserve_cus = 0
        if reserve_cus and self.depth > 1:
            # Chip partition.  A C(t) launch keeps every CU full for its whole duration and the workgroup dispatcher
            # hands a freed slot to the next workgroup of the launch already in flight: fit kernels queued meanwhile
            # (even on a high-priority queue) only start when the C(t) grid has drained (rocprofv3 kernel trace:
            # k_trf<3> 6.6 ms queued behind C(t) against 0.39 ms alone).  So the throughput kernels run on a stream
            # whose CU mask leaves `reserve_cus` CUs free for the fits.  Mask bit i is CU i/8 of XCD i%8 on MI355X
            # (scripts/dev_cumask.py), so a multiple of 8 taken from the top keeps the 8 XCDs balanced.
            ncu = ctx.device_info()['n_cu']
            nx = 8
            r = min(ncu - nx, (int(reserve_cus) + nx - 1) // nx * nx)
            self.reserve_cus = r
            self.main = self._masked_stream(range(ncu - r), ncu)
            if fits_on_reserved_only:
                resv_words = self._mask_words(range(ncu - r, ncu), ncu)
        # one side stream per slot so that the stragglers of consecutive batches overlap each other as well
        if self.depth > 2:
            self.sides = [self._fit_stream(resv_words) for _ in range(self.depth)]
        else:
            self.sides = None
        self.fitstream = self._fit_stream(resv_words) if self.depth > 1 else self.main
        self.soa = torch.empty((V, 3, self.Npad), device=device, dtype=torch.float32)
        Pmax = max(self.listDoG)
        self.slots = [_Slot(device, V, self.L, R, self.nbins, Pmax) for _ in range(self.depth)]
        t = hostct.calculate_dt(dt, F * dt)
        self.t_host = np.ascontiguousarray(np.broadcast_to(t, (V, self.L)))
        self.t_dev = torch.from_numpy(self.t_host).to(device)
        self.tau_max = self.t_host[0, -1] * 10                     # fitting_Ct_functions.py:324
        self.binvecs = hm.lambert_bin_vectors(self.edges)
        self.csa = csa
        self.nfev_total = 0
        self.nfev_last = {}
        # field-dependent constants of the old API (spectral_densities.py:1630-1645, 1696-1701), once
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

    # ---- streams ----
    @staticmethod
    def _mask_words(bits, ncu):
        w = [0] * ((ncu + 31) // 32)
        for b in bits:
            w[b // 32] |= 1 << (b % 32)
        return w

    def _masked_stream(self, bits, ncu):
        h = self.ctx.stream_create(self._mask_words(bits, ncu))
        self._owned_streams.append(h)
        return torch.cuda.ExternalStream(h, device=self.dev)

    def _fit_stream(self, resv_words):
        if resv_words is None:
            return torch.cuda.Stream(device=self.dev, priority=-1)
        h = self.ctx.stream_create(resv_words)
        self._owned_streams.append(h)
        return torch.cuda.ExternalStream(h, device=self.dev)

    def close(self):
        torch.cuda.synchronize(self.dev)
        self.ctx.set_stream(0)
        for h in self._owned_streams:
            self.ctx.stream_destroy(h)
        self._owned_streams = []

    # ---- stages on the main stream ----
    def stage_pack(self, vecs):
        self.ctx.pack_soa_dev(vecs.data_ptr(), self.frames, vecs.shape[1], 0, self.V, self.soa.data_ptr(), self.Npad)

    def stage_ct(self, s=None):
        s = s or self.slots[0]
        self.ctx.ct_palmer_dev(self.soa.data_ptr(), self.Npad, self.R, self.F, self.V, s.Ct.data_ptr(), s.dCt.data_ptr())

    def stage_hist(self, s=None):
        s = s or self.slots[0]
        self.ctx.rotate_hist_dev(self.soa.data_ptr(), self.Npad, self.N, self.V, self.q, self.edges[0], self.edges[1],
                                 s.hist.data_ptr(), s.vecsum.data_ptr(), s.outer.data_ptr(), self.F)

    # ---- fits ----
    def _launch_fit(self, s, req):
        V, nP = self.V, req['nParams']
        skip = np.ones(V, dtype=np.uint8)
        skip[req['idx']] = 0
        p0_full = np.zeros((V, nP))
        p0_full[req['idx']] = req['p0']
        s.skip.copy_(torch.from_numpy(skip))
        p0v = s.p0.view(-1)[: V * nP].view(V, nP)
        p0v.copy_(torch.from_numpy(p0_full))
        self.ctx.expfit_dev(self.t_dev.data_ptr(), s.CtT.data_ptr(), s.dCtT.data_ptr(), V, self.L, nP, p0v.data_ptr(),
                            self.tau_max, 100 * nP, s.popt.data_ptr(), s.pcov.data_ptr(), s.chi.data_ptr(),
                            s.status.data_ptr(), s.nfev.data_ptr(), skip_ptr=s.skip.data_ptr(), work_ptr=s.fitwork.data_ptr())

    def _collect_fit(self, s, req):
        V, nP, idx = self.V, req['nParams'], req['idx']
        popt = s.popt.view(-1)[: V * nP].view(V, nP).cpu().numpy()[idx]
        pcov = s.pcov.view(-1)[: V * nP * nP].view(V, nP, nP)
        dvar = torch.diagonal(pcov, dim1=1, dim2=2).cpu().numpy()[idx]
        chi = s.chi.cpu().numpy()[idx]
        status = s.status.cpu().numpy()[idx]
        nf = s.nfev.cpu().numpy()[idx]
        self.nfev_total += int(nf.sum())
        self.nfev_last[nP] = nf
        with np.errstate(invalid='ignore'):
            dP = np.sqrt(dvar)
        s.search.submit(popt, dP, chi, status)

    def stage_fit_begin(self, s, defer_last, transposed=False):
        if not transposed:
            self.ctx.transpose_dev(s.Ct.data_ptr(), self.L, self.V, s.CtT.data_ptr())
            self.ctx.transpose_dev(s.dCt.data_ptr(), self.L, self.V, s.dCtT.data_ptr())
        # the initial guesses only need the first / last ten lags of every residue (fitting_Ct_functions.py:366-368)
        head = s.CtT[:, :10].cpu().numpy()
        tail = s.CtT[:, -10:].cpu().numpy()
        s.search = fitCt.OrderSearchBatch(self.t_host, _EdgeOnly(head, tail, self.L), self.listDoG)
        s.pending = None
        while True:
            req = s.search.request()
            if req is None:
                break
            last = (s.search.j == len(self.listDoG) - 1)
            if last and defer_last:
                # enqueue behind everything already on the main stream, on the side stream; collect later
                side = self.sides[self.slots.index(s)]
                cur = torch.cuda.current_stream(self.dev)
                ready = torch.cuda.Event()
                ready.record(cur)
                side.wait_event(ready)
                self.ctx.set_stream(side.cuda_stream)
                with torch.cuda.stream(side):
                    self._launch_fit(s, req)
                    done = torch.cuda.Event()
                    done.record(side)
                self.ctx.set_stream(cur.cuda_stream)
                s.pending = (req, done)
                break
            self._launch_fit(s, req)
            self._collect_fit(s, req)

    def stage_fit_end(self, s):
        if s.pending is not None:
            req, done = s.pending
            done.synchronize()
            self._collect_fit(s, req)
            s.pending = None
        self.fit_best, self.fit_orders = s.search.best, s.search.per_order

    def selected_params(self, s=None):
        s = s or self.slots[0]
        return s.search.selected_arrays(Kmax=max(self.listDoG) // 2)

    def stage_relax(self, s=None):
        s = s or self.slots[0]
        S2, C, tau, K, _ = self.selected_params(s)
        z = self.zeta
        oms, fdd, fcsa, tf, gr = self._relax_consts
        if self.aniso is None or self.aniso == 1.0:
            out, _ = self.ctx.relax(1, [self.Diso], oms, fdd, fcsa, tf, gr, z * S2, z * C, tau, K)
        else:
            Dpar, Dperp = hm.symmtop_from_iso(self.Diso, self.aniso)
            out, _ = self.ctx.relax(2, [Dpar, Dperp], oms, fdd, fcsa, tf, gr, z * S2, z * C, tau, K,
                                    binvecs=self.binvecs, weights_dev_ptr=s.hist.data_ptr(), noe_mode=0)
        s.relax_out = out
        self.relax_out = out
        return out

    # ---- batch-level API ----
    def front(self, vecs, k, events=None):
        """Stage 1 of batch k on the main stream: pack, C(t), histogram (+ the transposes the fits read)."""
        s = self.slots[k % self.depth]
        self.ctx.set_stream(self.main.cuda_stream)
        with torch.cuda.stream(self.main):
            self.stage_pack(vecs)
            if events is not None:
                events[0].record(self.main)
            self.stage_ct(s)
            if events is not None:
                events[1].record(self.main)
            self.stage_hist(s)
            if events is not None:
                events[2].record(self.main)
            self.ctx.transpose_dev(s.Ct.data_ptr(), self.L, self.V, s.CtT.data_ptr())
            self.ctx.transpose_dev(s.dCt.data_ptr(), self.L, self.V, s.dCtT.data_ptr())
            s.front_done = torch.cuda.Event()
            s.front_done.record(self.main)
        return s

    def fits(self, k):
        """Stage 2 of batch k on the fit stream (host in the loop); the last order goes to a side stream when
        depth > 2.  (Solving all orders for all residues speculatively, without the host in the loop, was tried:
        it gives the same selection but its 512 nine-parameter workgroups at 256 VGPRs displace two C(t)
        workgroups each and cost more than the latency chain they remove.)"""
        s = self.slots[k % self.depth]
        self.fitstream.wait_event(s.front_done)
        self.ctx.set_stream(self.fitstream.cuda_stream)
        with torch.cuda.stream(self.fitstream):
            self.stage_fit_begin(s, defer_last=self.depth > 2, transposed=True)
        return s

    def finish(self, k):
        s = self.slots[k % self.depth]
        self.ctx.set_stream(self.fitstream.cuda_stream)
        with torch.cuda.stream(self.fitstream):
            self.stage_fit_end(s)
            out = self.stage_relax(s)
        return out

    def run(self, vecs, nb, events=None, on_finished=None):
        """nb batches through the software pipeline.  on_finished(slot) is called for every finished batch."""
        D = self.depth
        lag = max(0, D - 2)
        self.front(vecs, 0, None if events is None else events[0])
        for k in range(nb):
            if D > 1 and k + 1 < nb:
                self.front(vecs, k + 1, None if events is None else events[k + 1])
            self.fits(k)
            if k - lag >= 0:
                self.finish(k - lag)
                if on_finished is not None:
                    on_finished(self.slots[(k - lag) % D])
            if D == 1 and k + 1 < nb:
                self.front(vecs, k + 1, None if events is None else events[k + 1])
        for k in range(max(0, nb - lag), nb):
            self.finish(k)
            if on_finished is not None:
                on_finished(self.slots[k % D])
        self.ctx.set_stream(self.main.cuda_stream)

    def step(self, vecs, with_hist=True):
        """Serial form: one batch from vectors to R1/R2/NOE on the main stream."""
        s = self.slots[0]
        self.ctx.set_stream(self.main.cuda_stream)
        self.stage_pack(vecs)
        self.stage_ct(s)
        if with_hist:
            self.stage_hist(s)
        self.stage_fit_begin(s, defer_last=False)
        self.stage_fit_end(s)
        return self.stage_relax(s)

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


class _EdgeOnly:
    """Array stand-in exposing only what initial_guess_batch reads of C(t): y[:, :10] and y[:, -10:]."""

    def __init__(self, head, tail, L):
        self.head, self.tail, self.L = head, tail, L
        self.shape = (head.shape[0], L)

    def __getitem__(self, key):
        if isinstance(key, np.ndarray):                      # y[idx] -> same view restricted to rows idx
            return _EdgeOnly(self.head[key], self.tail[key], self.L)
        rows, cols = key
        if cols == slice(None, 10, None):
            return self.head[rows]
        if cols == slice(-10, None, None):
            return self.tail[rows]
        raise IndexError('only the first and last ten lags are resident on the host')
