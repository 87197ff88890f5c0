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
between; bench.py times it and the CLI scripts reuse its stages one at a time.  torch is used for device
memory and streams only; every computation goes through the C ABI.
"""
import numpy as np
import torch

from . import ct as hostct
from . import fitting_Ct_functions as fitCt
from . import _hostmath as hm
from . import spectral_densities as sd


class DevicePipeline:
    def __init__(self, ctx, device, frames, V, R, F, dt, q_rot=None, Diso=None, aniso=None, field_MHz=(600.133,),
                 zeta=0.890023, histBinX=72, listDoG=(2, 3, 5, 7, 9), csa=None):
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
        f64 = dict(device=device, dtype=torch.float64)
        self.soa = torch.empty((V, 3, self.Npad), device=device, dtype=torch.float32)
        self.Ct = torch.empty((self.L, V), **f64)
        self.dCt = torch.empty((self.L, V), **f64)
        self.CtT = torch.empty((V, self.L), **f64)
        self.dCtT = torch.empty((V, self.L), **f64)
        self.hist = torch.empty((V, self.nbins), **f64)
        self.vecsum = torch.empty((V, 3), **f64)
        self.outer = torch.empty((R, V, 6), **f64)
        t = hostct.calculate_dt(dt, F * dt)
        self.t_host = np.ascontiguousarray(np.broadcast_to(t, (V, self.L)))
        self.t_dev = torch.from_numpy(self.t_host).to(device)
        Pmax = max(self.listDoG)
        self.p0_dev = torch.empty((V, Pmax), **f64)
        self.popt_dev = torch.empty((V, Pmax), **f64)
        self.pcov_dev = torch.empty((V, Pmax * Pmax), **f64)
        self.chi_dev = torch.empty((V,), **f64)
        self.status_dev = torch.empty((V,), device=device, dtype=torch.int32)
        self.nfev_dev = torch.empty((V,), device=device, dtype=torch.int32)
        self.skip_dev = torch.zeros((V,), device=device, dtype=torch.uint8)
        self.binvecs = hm.lambert_bin_vectors(self.edges)
        self.csa = csa
        self.nfev_total = 0
        self.nfev_last = {}

    # ---- stages ----
    def stage_pack(self, vecs):
        self.ctx.pack_soa_dev(vecs.data_ptr(), self.frames, vecs.shape[1], 0, self.V, self.soa.data_ptr(), self.Npad)

    def stage_ct(self):
        self.ctx.ct_palmer_dev(self.soa.data_ptr(), self.Npad, self.R, self.F, self.V, self.Ct.data_ptr(), self.dCt.data_ptr())

    def stage_hist(self):
        self.ctx.rotate_hist_dev(self.soa.data_ptr(), self.Npad, self.N, self.V, self.q, self.edges[0], self.edges[1],
                                 self.hist.data_ptr(), self.vecsum.data_ptr(), self.outer.data_ptr(), self.F)

    def _runner(self, tau_max):
        V = self.V

        def run(nParams, p0, idx):
            # residues not in idx are skipped on the device; their rows keep stale values that are never read
            skip = np.ones(V, dtype=np.uint8)
            skip[idx] = 0
            p0_full = np.zeros((V, nParams))
            p0_full[idx] = p0
            self.skip_dev.copy_(torch.from_numpy(skip))
            p0v = self.p0_dev.view(-1)[: V * nParams].view(V, nParams)
            p0v.copy_(torch.from_numpy(p0_full))
            self.ctx.expfit_dev(self.t_dev.data_ptr(), self.CtT.data_ptr(), self.dCtT.data_ptr(), V, self.L, nParams,
                                p0v.data_ptr(), tau_max, 100 * nParams, self.popt_dev.data_ptr(), self.pcov_dev.data_ptr(),
                                self.chi_dev.data_ptr(), self.status_dev.data_ptr(), self.nfev_dev.data_ptr(),
                                skip_ptr=self.skip_dev.data_ptr())
            popt = self.popt_dev.view(-1)[: V * nParams].view(V, nParams).cpu().numpy()[idx]
            pcov = self.pcov_dev.view(-1)[: V * nParams * nParams].view(V, nParams, nParams)
            dvar = torch.diagonal(pcov, dim1=1, dim2=2).cpu().numpy()[idx]
            chi = self.chi_dev.cpu().numpy()[idx]
            status = self.status_dev.cpu().numpy()[idx]
            nf = self.nfev_dev.cpu().numpy()[idx]
            self.nfev_total += int(nf.sum())
            self.nfev_last[nParams] = nf
            with np.errstate(invalid='ignore'):
                dP = np.sqrt(dvar)
            return popt, dP, chi, status
        return run

    def stage_fit(self):
        self.ctx.transpose_dev(self.Ct.data_ptr(), self.L, self.V, self.CtT.data_ptr())
        self.ctx.transpose_dev(self.dCt.data_ptr(), self.L, self.V, self.dCtT.data_ptr())
        # the initial guesses only need the first / last ten lags of every residue (fitting_Ct_functions.py:366-368)
        head = self.CtT[:, :10].cpu().numpy()
        tail = self.CtT[:, -10:].cpu().numpy()
        y_small = _EdgeOnly(head, tail, self.L)
        best, per_order = fitCt.order_search_batch(self.t_host, y_small, self._runner(self.t_host[0, -1] * 10), self.listDoG)
        self.fit_best, self.fit_orders = best, per_order
        return best, per_order

    def selected_params(self):
        """S2, C (V,Kmax), tau (V,Kmax), nComps of the selected models (components sorted by tau)."""
        V = self.V
        Kmax = max(self.listDoG) // 2
        S2 = np.zeros(V)
        C = np.zeros((V, Kmax))
        tau = np.ones((V, Kmax))
        K = np.zeros(V, dtype=np.int32)
        chi = np.full(V, np.nan)
        for i in range(V):
            if self.fit_best[i] < 0:
                continue
            f = fitCt._fit_dict(self.fit_orders[self.fit_best[i]], i)
            k = len(f['C'])
            K[i] = k
            C[i, :k] = f['C']
            tau[i, :k] = f['tau']
            S2[i] = f['S2']
            chi[i] = f['chiSq']
        return S2, C, tau, K, chi

    def stage_relax(self):
        S2, C, tau, K, _ = self.selected_params()
        z = self.zeta
        outs = []
        oms, fdd, fcsa, tf, gr = [], [], [], [], []
        for MHz in self.fields:
            RObj = sd.relaxationModel('NH', 2.0 * np.pi * (MHz * 1e6) / 267.513e6)
            RObj.set_time_unit('ps')
            oms.append(RObj.omega)
            fdd.append(RObj.get_f_DD())
            csa = np.repeat(RObj.gX.csa, self.V) if self.csa is None else np.asarray(self.csa, dtype=float)
            fcsa.append(RObj.get_f_CSA(csa))
            tf.append(RObj.time_fact)
            gr.append(RObj.gH.gamma / RObj.gX.gamma)
        if self.aniso is None or self.aniso == 1.0:
            out, _ = self.ctx.relax(1, [self.Diso], np.array(oms), fdd, np.array(fcsa), tf, gr, z * S2, z * C, tau, K)
        else:
            Dpar, Dperp = hm.symmtop_from_iso(self.Diso, self.aniso)
            out, _ = self.ctx.relax(2, [Dpar, Dperp], np.array(oms), fdd, np.array(fcsa), tf, gr, z * S2, z * C, tau, K,
                                    binvecs=self.binvecs, weights_dev_ptr=self.hist.data_ptr(), noe_mode=0)
        self.relax_out = out
        return out

    def step(self, vecs, with_hist=True):
        self.stage_pack(vecs)
        self.stage_ct()
        if with_hist:
            self.stage_hist()
        self.stage_fit()
        return self.stage_relax()


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
