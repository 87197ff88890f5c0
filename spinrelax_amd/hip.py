"""
numpy-facing wrapper around the C ABI (spinrelax_amd/_lib.py -> libspinrelax_hip.so).

`Context` owns one sr_ctx (one GPU).  Methods without the `_dev` suffix take and return numpy arrays
(host buffers, blocking); `_dev` methods take raw device addresses (ints, e.g. torch
``tensor.data_ptr()``) and only enqueue work on the context's stream.
"""
import ctypes
import sys

import numpy as np

from . import _lib
from ._lib import SpinRelaxHipError, check


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Context:
    def __init__(self, device=0):
        self.lib = _lib.load()
        self.h = self.lib.sr_create(int(device))
        if not self.h:
            raise SpinRelaxHipError('sr_create(%d) failed: %s' % (device, _lib.last_error()))
        self.device = int(device)

    def close(self):
        """Deterministic teardown: synchronises the device and frees the context's work areas.  Call it (or use the
        context as a `with` block) before the interpreter exits; __del__ deliberately does NOT talk to HIP once the
        interpreter is finalising -- the runtime's own static destructors may already have run by then."""
        if getattr(self, 'h', None):
            self.lib.sr_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- plumbing ----
    def set_stream(self, stream_handle):
        check(self.lib.sr_set_stream(self.h, ctypes.c_void_p(stream_handle or 0)), 'sr_set_stream')

    def sync(self):
        check(self.lib.sr_sync(self.h), 'sr_sync')

    def set_option(self, name, value):
        check(self.lib.sr_set_option(self.h, name.encode(), int(value)), 'sr_set_option')

    def stream_create(self, cu_mask_words=None, priority=0):
        """hipStream_t handle (int); cu_mask_words = iterable of 32-bit words, bit set = CU usable."""
        out = ctypes.c_void_p()
        if cu_mask_words is not None:
            words = (ctypes.c_uint32 * len(cu_mask_words))(*[int(w) & 0xFFFFFFFF for w in cu_mask_words])
            check(self.lib.sr_stream_create(self.h, words, len(cu_mask_words), priority, ctypes.byref(out)),
                  'sr_stream_create')
        else:
            check(self.lib.sr_stream_create(self.h, None, 0, priority, ctypes.byref(out)), 'sr_stream_create')
        return out.value

    def stream_destroy(self, handle):
        check(self.lib.sr_stream_destroy(self.h, ctypes.c_void_p(handle)), 'sr_stream_destroy')

    def device_sync(self):
        check(self.lib.sr_device_sync(self.h), 'sr_device_sync')

    def host_alloc(self, nbytes):
        """Page-locked host memory owned by the library (hipHostMalloc); returns the address."""
        p = self.lib.sr_host_alloc(self.h, int(nbytes))
        if not p:
            raise SpinRelaxHipError('sr_host_alloc(%d) failed: %s' % (nbytes, _lib.last_error()))
        return p

    def host_free(self, addr):
        check(self.lib.sr_host_free(self.h, ctypes.c_void_p(addr)), 'sr_host_free')

    def memcpy_d2h_async(self, host_addr, dev_ptr, nbytes):
        check(self.lib.sr_memcpy_d2h_async(self.h, ctypes.c_void_p(host_addr), ctypes.c_void_p(dev_ptr), int(nbytes)),
              'sr_memcpy_d2h_async')

    def memcpy_h2d_async(self, dev_ptr, host_addr, nbytes):
        check(self.lib.sr_memcpy_h2d_async(self.h, ctypes.c_void_p(dev_ptr), ctypes.c_void_p(host_addr), int(nbytes)),
              'sr_memcpy_h2d_async')

    def device_info(self):
        ncu = ctypes.c_int()
        hbm = ctypes.c_int64()
        lds = ctypes.c_int()
        name = ctypes.create_string_buffer(256)
        check(self.lib.sr_device_info(self.h, ctypes.byref(ncu), ctypes.byref(hbm), ctypes.byref(lds), name, 256),
              'sr_device_info')
        return dict(n_cu=ncu.value, hbm_bytes=hbm.value, lds_per_cu=lds.value, name=name.value.decode())

    def timer_start(self):
        check(self.lib.sr_timer_start(self.h), 'sr_timer_start')

    def timer_stop_ms(self):
        ms = ctypes.c_float()
        check(self.lib.sr_timer_stop_ms(self.h, ctypes.byref(ms)), 'sr_timer_stop_ms')
        return ms.value

    def max_frames_per_chunk(self):
        return int(self.lib.sr_ct_max_frames_per_chunk(self.h))

    # ---- kernel 1 ----
    def ct_palmer(self, vecs, R, F, v0=0, nV=None, chunk_start=None, mode=0):
        """vecs (N, Vtot, 3) float32 -> Ct, dCt (F//2, nV) float64.  calculate-Ct-from-traj.py:200-238."""
        vecs = _f32(vecs)
        if vecs.ndim != 3 or vecs.shape[2] != 3:
            raise ValueError('vecs must be (N, V, 3)')
        N, Vtot, _ = vecs.shape
        nV = Vtot - v0 if nV is None else nV
        L = F // 2
        Ct = np.empty((L, nV))
        dCt = np.empty((L, nV))
        cs = None if chunk_start is None else np.ascontiguousarray(chunk_start, dtype=np.int64)
        if cs is not None and cs.shape != (R,):
            raise ValueError('chunk_start must have R entries')
        check(self.lib.sr_ct_palmer_f32(self.h, _ptr(vecs), N, Vtot, v0, nV, R, F, _ptr(cs), int(mode), _ptr(Ct), _ptr(dCt)),
              'sr_ct_palmer_f32')
        return Ct, dCt

    def pack_soa_dev(self, vecs_ptr, N, Vtot, v0, nV, soa_ptr, Npad):
        check(self.lib.sr_pack_soa_f32_dev(self.h, vecs_ptr, N, Vtot, v0, nV, soa_ptr, Npad), 'sr_pack_soa_f32_dev')

    def pack_soa_rot_dev(self, vecs_ptr, N, Vtot, v0, nV, quat_ptr, soa_ptr, Npad):
        check(self.lib.sr_pack_soa_rot_f32_dev(self.h, vecs_ptr, N, Vtot, v0, nV, quat_ptr, soa_ptr, Npad), 'sr_pack_soa_rot_f32_dev')

    def ct_palmer_dev(self, soa_ptr, Npad, R, F, nV, Ct_ptr, dCt_ptr, chunk_start=None, mode=0, psum_ptr=None):
        cs = None if chunk_start is None else np.ascontiguousarray(chunk_start, dtype=np.int64)
        check(self.lib.sr_ct_palmer_f32_dev(self.h, soa_ptr, Npad, R, F, nV, _ptr(cs), int(mode), psum_ptr, Ct_ptr, dCt_ptr),
              'sr_ct_palmer_f32_dev')

    def ct_sums_dev(self, soa_ptr, Npad, R, F, nV, psum_ptr, chunk_start=None, mode=0):
        cs = None if chunk_start is None else np.ascontiguousarray(chunk_start, dtype=np.int64)
        check(self.lib.sr_ct_palmer_sums_f32_dev(self.h, soa_ptr, Npad, R, F, nV, None if cs is None else _ptr(cs), mode, psum_ptr),
              'sr_ct_palmer_sums_f32_dev')

    def ct_finalize_dev(self, psum_ptr, R, F, nV, Ct_ptr, dCt_ptr, CtT_ptr=None, dCtT_ptr=None):
        """mean / std over the chunks; with CtT_ptr / dCtT_ptr the same launch also writes the (nV, L) copies the fits read"""
        if CtT_ptr is None:
            check(self.lib.sr_ct_finalize_f64_dev(self.h, psum_ptr, R, F, nV, Ct_ptr, dCt_ptr), 'sr_ct_finalize_f64_dev')
        else:
            check(self.lib.sr_ct_finalize_t_f64_dev(self.h, psum_ptr, R, F, nV, Ct_ptr, dCt_ptr, CtT_ptr, dCtT_ptr),
                  'sr_ct_finalize_t_f64_dev')

    def psum_stride(self, F):
        return int(self.lib.sr_ct_psum_stride(F))

    # ---- resident vectors (sr_vectors.hip) ----
    def vectors(self, nV, capacity=0):
        """An empty ResidentVectors object for nV vectors (this rank's columns)."""
        return ResidentVectors(self, nV, capacity)

    def counter(self, name):
        v = ctypes.c_uint64()
        check(self.lib.sr_counter(self.h, name.encode(), ctypes.byref(v)), 'sr_counter')
        return int(v.value)

    def ct_finalize_sums(self, sums, F):
        """mean / std over the replicates from raw sums (nV, R, F//2): Ct, dCt (F//2, nV) -- the kernel of a single-process run"""
        sums = _f64(sums)
        nV, R, L = sums.shape
        if L != F // 2:
            raise ValueError('sums must be (vectors, chunks, F//2)')
        Ct = np.empty((L, nV))
        dCt = np.empty((L, nV))
        check(self.lib.sr_ct_finalize_sums_f64(self.h, _ptr(sums), R, int(F), nV, _ptr(Ct), _ptr(dCt)), 'sr_ct_finalize_sums_f64')
        return Ct, dCt

    # ---- kernel 2 ----
    def rotate_hist(self, vecs, q, edges_phi, edges_cos, v0=0, nV=None, block_len=0, want_outer=True):
        """vecs (N, Vtot, 3) float32 -> hist (nV, nphi, ncos), vecsum (nV,3), outer (nB, nV, 6)."""
        vecs = _f32(vecs)
        N, Vtot, _ = vecs.shape
        nV = Vtot - v0 if nV is None else nV
        ep = _f64(edges_phi)
        ec = _f64(edges_cos)
        nphi, ncos = ep.size - 1, ec.size - 1
        qq = None if q is None else _f64(q)
        hist = np.empty((nV, nphi, ncos))
        vecsum = np.empty((nV, 3))
        Fb = block_len if (block_len and 0 < block_len <= N) else N
        nB = N // Fb
        outer = np.empty((nB, nV, 6)) if want_outer else None
        check(self.lib.sr_rotate_hist_f32(self.h, _ptr(vecs), N, Vtot, v0, nV, _ptr(qq), _ptr(ep), nphi, _ptr(ec), ncos,
                                          _ptr(hist), _ptr(vecsum), _ptr(outer), int(block_len or 0)), 'sr_rotate_hist_f32')
        return hist, vecsum, outer

    def rotate_hist_dev(self, soa_ptr, Npad, N, nV, q, edges_phi, edges_cos, hist_ptr, vecsum_ptr, outer_ptr, block_len):
        ep = _f64(edges_phi)
        ec = _f64(edges_cos)
        qq = None if q is None else _f64(q)
        check(self.lib.sr_rotate_hist_f32_dev(self.h, soa_ptr, Npad, N, nV, _ptr(qq), _ptr(ep), ep.size - 1, _ptr(ec),
                                              ec.size - 1, hist_ptr, vecsum_ptr, outer_ptr, int(block_len or 0)),
              'sr_rotate_hist_f32_dev')

    def rotate_vectors(self, vecs, q, v0=0, nV=None):
        """q: (4,) one rotation for everything, or (N, 4) one UNIT quaternion per frame."""
        vecs = _f32(vecs)
        N, Vtot, _ = vecs.shape
        nV = Vtot - v0 if nV is None else nV
        out = np.empty((N, nV, 3))
        qq = None if q is None else _f64(q)
        if qq is not None and qq.ndim == 2:
            if qq.shape != (N, 4):
                raise ValueError('per-frame quaternions must have shape (frames, 4)')
            check(self.lib.sr_rotate_vectors_perframe_f32(self.h, _ptr(vecs), N, Vtot, v0, nV, _ptr(qq), _ptr(out)),
                  'sr_rotate_vectors_perframe_f32')
            return out
        check(self.lib.sr_rotate_vectors_f32(self.h, _ptr(vecs), N, Vtot, v0, nV, _ptr(qq), _ptr(out)), 'sr_rotate_vectors_f32')
        return out

    # ---- trajectory front end ----
    def xh_vectors(self, xyz, indexX, indexH, fit_indices=None, ref_xyz=None, want_lab=True, want_quat=False):
        """xyz (nFrames, nAtoms, 3) float32 -> (vec_lab, vec_fit, quat): unit X-H vectors in the lab frame, after the
        per-frame superposition onto ref_xyz over fit_indices (None without a reference), and the rotations."""
        xyz = _f32(xyz)
        if xyz.ndim != 3 or xyz.shape[2] != 3:
            raise ValueError('xyz must be (frames, atoms, 3)')
        nF, nA, _ = xyz.shape
        iX = np.ascontiguousarray(indexX, dtype=np.int32)
        iH = np.ascontiguousarray(indexH, dtype=np.int32)
        if iX.shape != iH.shape or iX.ndim != 1:
            raise ValueError('indexX and indexH must be 1-D and of equal length')
        nV = iX.size
        fit = ref_xyz is not None
        fi = np.ascontiguousarray(fit_indices, dtype=np.int32) if fit else None
        rx = _f32(ref_xyz) if fit else None
        if fit and rx.shape != (nA, 3):
            raise ValueError('ref_xyz must be (atoms, 3)')
        lab = np.empty((nF, nV, 3), dtype=np.float32) if want_lab else None
        fitv = np.empty((nF, nV, 3), dtype=np.float32) if fit else None
        quat = np.empty((nF, 4)) if (fit and want_quat) else None
        check(self.lib.sr_xh_vectors_f32(self.h, _ptr(xyz), nF, nA, _ptr(iX), _ptr(iH), nV, _ptr(fi), fi.size if fit else 0,
                                         _ptr(rx), _ptr(lab), _ptr(fitv), _ptr(quat)), 'sr_xh_vectors_f32')
        return lab, fitv, quat

    # ---- global rotational diffusion (calculate-dq-distribution.py) ----
    def dq_moments(self, q, lags, nchunk=1):
        """q (N, 4) (w x y z), float32 (PLUMED's precision) or float64 (kept as float64: the gmx-rotmat route); lags: frame
        offsets -> (nlags, nchunk, 7) float64: sums of xx yy zz xy xz yz of the vector part of q_i^-1 q_{i+lag} over every
        chunk's samples, and the sample count."""
        q = np.asarray(q)
        f64 = q.dtype == np.float64
        q = _f64(q) if f64 else _f32(q)
        if q.ndim != 2 or q.shape[1] != 4:
            raise ValueError('q must be (N, 4)')
        lg = np.ascontiguousarray(lags, dtype=np.int32)
        out = np.empty((lg.size, int(nchunk), 7))
        fn = self.lib.sr_dq_moments_f64 if f64 else self.lib.sr_dq_moments_f32
        check(fn(self.h, _ptr(q), q.shape[0], _ptr(lg), lg.size, int(nchunk), _ptr(out)), 'sr_dq_moments_f64' if f64 else 'sr_dq_moments_f32')
        return out

    # ---- per-residue CSA refinement of the legacy `--opt new` mode ----
    def legacy_csa_search(self, D, omega, f_DD, gammaB0_sq, time_fact, gamma_ratio, S2, C, tau, nComps, binvecs, weights, expt,
                          csa0, step=1.0, xtol=1e-4, ftol=1e-4, maxiter=1000, maxfun=1000):
        """fmin_powell over the CSA of every residue against its measured (R1, R2, NOE) triple
        (calculate-relaxations-from-Ct.py:210-258, 935-1000): one launch, a workgroup per residue.  D = (Dpar, Dperp); S2 (n),
        C / tau (n, Kmax), nComps (n) already scaled by zeta; binvecs (B, 3), weights (n, B); expt (n, 3, 2) = value and
        uncertainty of R1, R2, NOE; csa0 (n).  Returns csa (n) = Powell's optimum, fopt (n), nfev (n)."""
        Dd = _f64(np.atleast_1d(D))
        om = _f64(np.ravel(omega))
        S2 = _f64(S2)
        n = S2.size
        C, tau = _f64(np.atleast_2d(C)), _f64(np.atleast_2d(tau))
        Kmax = C.shape[1]
        nc = np.ascontiguousarray(nComps, dtype=np.int32)
        bv, w, ex, c0 = _f64(binvecs), _f64(weights), _f64(expt), _f64(csa0)
        B = bv.shape[0]
        if Dd.shape != (2,) or om.shape != (5,):
            raise ValueError('D must be (Dpar, Dperp), omega the 5 angular frequencies')
        if C.shape != (n, Kmax) or tau.shape != (n, Kmax) or nc.shape != (n,) or bv.shape != (B, 3) or w.shape != (n, B) or \
                ex.shape != (n, 3, 2) or c0.shape != (n,):
            raise ValueError('shapes: C / tau (n, Kmax), nComps (n), binvecs (B, 3), weights (n, B), expt (n, 3, 2), csa0 (n)')
        csa, fopt = np.empty(n), np.empty(n)
        nfev = np.empty(n, dtype=np.int32)
        check(self.lib.sr_legacy_csa_search_f64(self.h, _ptr(Dd), _ptr(om), float(f_DD), float(gammaB0_sq), float(time_fact),
                                                float(gamma_ratio), n, Kmax, _ptr(S2), _ptr(C), _ptr(tau), _ptr(nc), B, _ptr(bv),
                                                _ptr(w), _ptr(ex), _ptr(c0), float(step), float(xtol), float(ftol), int(maxiter),
                                                int(maxfun), _ptr(csa), _ptr(fopt), _ptr(nfev)), 'sr_legacy_csa_search_f64')
        return csa, fopt, nfev

    # ---- residue-specific CSA search (new class API) ----
    def rscsa_search(self, stats, column, csa_prefactor, noe_factor, f_DD, target, dtarget, cover, has_err, csa0, step,
                     xtol=1e-4, ftol=1e-4):
        """One-variable Powell search per residue over the closed forms of the 12 statistics (spectral_densities.py:1371-1382,
        1430-1447).  stats (E, n, 12); target/dtarget/cover (E, n).  Returns csa (n), values (E, n), errors (E, n), fopt (n),
        nfev (n)."""
        stats = _f64(stats)
        E, n = stats.shape[0], stats.shape[1]
        if stats.shape != (E, n, 12):
            raise ValueError('stats must be (E, n, 12)')
        col = np.ascontiguousarray(column, dtype=np.int32)
        pref, cn, fdd = _f64(csa_prefactor), _f64(noe_factor), _f64(f_DD)
        y, dy = _f64(target), _f64(dtarget)
        cov = np.ascontiguousarray(cover, dtype=np.uint8)
        c0 = _f64(csa0)
        if col.shape != (E,) or pref.shape != (E,) or cn.shape != (E,) or fdd.shape != (E,):
            raise ValueError('per-experiment arrays must have E entries')
        if y.shape != (E, n) or dy.shape != (E, n) or cov.shape != (E, n) or c0.shape != (n,):
            raise ValueError('target, dtarget and cover must be (E, n), csa0 (n)')
        csa, fopt = np.empty(n), np.empty(n)
        vals, errs = np.empty((E, n)), np.empty((E, n))
        nfev = np.empty(n, dtype=np.int32)
        check(self.lib.sr_rscsa_search_f64(self.h, E, n, _ptr(stats), _ptr(col), _ptr(pref), _ptr(cn), _ptr(fdd), _ptr(y), _ptr(dy),
                                           _ptr(cov), int(bool(has_err)), _ptr(c0), float(step), float(xtol), float(ftol),
                                           _ptr(csa), _ptr(vals), _ptr(errs), _ptr(fopt), _ptr(nfev)), 'sr_rscsa_search_f64')
        return csa, vals, errs, fopt, nfev

    # ---- kernel 3b ----
    def expfit_resjac(self, t, y, sigma, params, want_jac=True):
        t = _f64(np.atleast_2d(t))
        y = _f64(np.atleast_2d(y))
        nRes, L = y.shape
        if t.shape[0] == 1 and nRes > 1:
            t = _f64(np.broadcast_to(t, (nRes, L)))
        s = None if sigma is None else _f64(np.atleast_2d(sigma))
        p = _f64(np.atleast_2d(params))
        P = p.shape[1]
        resid = np.empty((nRes, L))
        jac = np.empty((nRes, L, P)) if want_jac else None
        check(self.lib.sr_expfit_resjac_f64(self.h, _ptr(t), _ptr(y), _ptr(s), _ptr(p), nRes, L, P, _ptr(resid), _ptr(jac)),
              'sr_expfit_resjac_f64')
        return resid, jac

    def expfit(self, t, y, sigma, p0, tau_max, max_nfev=0, analytic_jac=False):
        """Batched bounded fit (scipy curve_fit/TRF semantics).  Returns popt, pcov, chisq, status, nfev."""
        y = _f64(np.atleast_2d(y))
        nRes, L = y.shape
        t = _f64(np.broadcast_to(np.atleast_2d(t), (nRes, L)))
        s = None if sigma is None else _f64(np.broadcast_to(np.atleast_2d(sigma), (nRes, L)))
        p0 = _f64(np.atleast_2d(p0))
        P = p0.shape[1]
        popt = np.empty((nRes, P))
        pcov = np.empty((nRes, P, P))
        chisq = np.empty(nRes)
        status = np.empty(nRes, dtype=np.int32)
        nfev = np.empty(nRes, dtype=np.int32)
        mi = int(max_nfev) if max_nfev else 100 * P
        if analytic_jac:
            mi = -mi
        check(self.lib.sr_expfit_lm_f64(self.h, _ptr(t), _ptr(y), _ptr(s), nRes, L, P, _ptr(p0), float(tau_max), mi,
                                        _ptr(popt), _ptr(pcov), _ptr(chisq), _ptr(status), _ptr(nfev)), 'sr_expfit_lm_f64')
        return popt, pcov, chisq, status, nfev

    def expfit_dev(self, t_ptr, y_ptr, sigma_ptr, nRes, L, P, p0_ptr, tau_max, max_nfev, popt_ptr, pcov_ptr, chisq_ptr,
                   status_ptr, nfev_ptr, skip_ptr=None, work_ptr=None):
        check(self.lib.sr_expfit_lm_f64_dev(self.h, t_ptr, y_ptr, sigma_ptr, nRes, L, P, p0_ptr, float(tau_max),
                                            int(max_nfev), skip_ptr, work_ptr, popt_ptr, pcov_ptr, chisq_ptr, status_ptr, nfev_ptr),
              'sr_expfit_lm_f64_dev')

    def order_search(self, t, y, sigma, orders, tau_guess, tau_max, chi_threshold=0.5):
        """optimised_curve_fitting for (n, L) host arrays in one launch (sr_expfit_order_search_f64).  tau_guess:
        (1 or n, sum(orders)//2... one block of order//2 guesses per order).  Returns a dict of host arrays."""
        t, y = _f64(t), _f64(y)
        n, L = y.shape
        sigma = None if sigma is None else _f64(sigma)
        orders = np.ascontiguousarray(orders, dtype=np.int32)
        nO, Pmax = orders.size, int(orders.max())
        Kmax = Pmax // 2
        tg = _f64(np.atleast_2d(tau_guess))
        out = dict(popt=np.empty((nO, n, Pmax)), dP=np.empty((nO, n, Pmax)), chisq=np.empty((nO, n)),
                   status=np.empty((nO, n), dtype=np.int32), nfev=np.empty((nO, n), dtype=np.int32),
                   best=np.empty(n, dtype=np.int32), S2=np.empty(n), C=np.empty((n, Kmax)), tau=np.empty((n, Kmax)),
                   chi=np.empty(n), K=np.empty(n, dtype=np.int32))
        check(self.lib.sr_expfit_order_search_f64(self.h, _ptr(t), _ptr(y), None if sigma is None else _ptr(sigma), n, L,
                                                  _ptr(orders), nO, _ptr(tg), tg.shape[0], float(tau_max), float(chi_threshold),
                                                  _ptr(out['popt']), _ptr(out['dP']), _ptr(out['chisq']), _ptr(out['status']),
                                                  _ptr(out['nfev']), _ptr(out['best']), _ptr(out['S2']), _ptr(out['C']),
                                                  _ptr(out['tau']), _ptr(out['chi']), _ptr(out['K'])),
              'sr_expfit_order_search_f64')
        out['orders'] = orders
        return out

    def order_search_dev(self, t_ptr, y_ptr, sigma_ptr, nRes, L, orders, tau_guess_ptr, tau_rows, tau_max, chi_threshold,
                         popt_ptr, dP_ptr, chisq_ptr, status_ptr, nfev_ptr, best_ptr, S2_ptr, C_ptr, tau_ptr, chi_ptr, K_ptr,
                         work_ptr=None):
        orders = np.ascontiguousarray(orders, dtype=np.int32)
        check(self.lib.sr_expfit_order_search_f64_dev(self.h, t_ptr, y_ptr, sigma_ptr, nRes, L, _ptr(orders), orders.size,
                                                      tau_guess_ptr, tau_rows, float(tau_max), float(chi_threshold), work_ptr,
                                                      popt_ptr, dP_ptr, chisq_ptr, status_ptr, nfev_ptr, best_ptr, S2_ptr, C_ptr,
                                                      tau_ptr, chi_ptr, K_ptr), 'sr_expfit_order_search_f64_dev')

    # ---- signals (a stream waits for a value a kernel writes) ----
    def signal_alloc(self):
        p = self.lib.sr_signal_alloc(self.h)
        if not p:
            raise SpinRelaxHipError('sr_signal_alloc failed: %s' % _lib.last_error())
        return p

    def signal_free(self, sig):
        check(self.lib.sr_signal_free(self.h, sig), 'sr_signal_free')

    def stream_wait_signal(self, sig, value):
        check(self.lib.sr_stream_wait_signal(self.h, sig, int(value)), 'sr_stream_wait_signal')

    def stream_write_signal(self, sig, value):
        check(self.lib.sr_stream_write_signal(self.h, sig, int(value)), 'sr_stream_write_signal')

    def order_search_batched_dev(self, t_ptr, t_rows, y_ptr, sigma_ptr, nRes, L, orders, tau_guess_ptr, tau_rows, tau_max, chi_threshold,
                                 popt_ptr, dP_ptr, chisq_ptr, status_ptr, nfev_ptr, best_ptr, S2_ptr, C_ptr, tau_ptr, chi_ptr, K_ptr,
                                 work_ptr=None, dispatch_order_ptr=None, tail_signal=None, tail_value=0):
        """the model-order search of several batches' residues in one launch (sr_expfit_order_search_batched_f64_dev):
        t_rows = 1 shares one time axis, dispatch_order (device int32, nRes) permutes the order in which residues start"""
        orders = np.ascontiguousarray(orders, dtype=np.int32)
        check(self.lib.sr_expfit_order_search_batched_f64_dev(self.h, t_ptr, t_rows, y_ptr, sigma_ptr, nRes, L, _ptr(orders), orders.size,
                                                              tau_guess_ptr, tau_rows, float(tau_max), float(chi_threshold),
                                                              dispatch_order_ptr, tail_signal, int(tail_value), work_ptr, popt_ptr, dP_ptr,
                                                              chisq_ptr, status_ptr,
                                                              nfev_ptr, best_ptr, S2_ptr, C_ptr, tau_ptr, chi_ptr, K_ptr),
              'sr_expfit_order_search_batched_f64_dev')

    def relax_dev(self, model, D, E, omega_ptr, fDD_ptr, fCSA_ptr, tf_ptr, gr_ptr, nRes, Kmax, zeta, S2_ptr, C_ptr, tau_ptr,
                  K_ptr, B, binvecs_ptr, weights_ptr, noe_mode, out_ptr, Jout_ptr=None, stats_ptr=None):
        Dh = _f64(np.atleast_1d(D))
        check(self.lib.sr_jomega_relax_f64_dev(self.h, model, _ptr(Dh), E, omega_ptr, fDD_ptr, fCSA_ptr, tf_ptr, gr_ptr, nRes,
                                               Kmax, float(zeta), S2_ptr, C_ptr, tau_ptr, K_ptr, B, binvecs_ptr, weights_ptr,
                                               noe_mode, out_ptr, Jout_ptr, stats_ptr), 'sr_jomega_relax_f64_dev')

    def transpose_dev(self, in_ptr, rows, cols, out_ptr):
        check(self.lib.sr_transpose_f64_dev(self.h, in_ptr, rows, cols, out_ptr), 'sr_transpose_f64_dev')

    # ---- kernel 3a ----
    def jomega(self, x, y):
        x, y = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
        xs = _f64(x)
        ys = _f64(y)
        out = np.empty(xs.shape)
        check(self.lib.sr_jomega_f64(self.h, _ptr(xs), _ptr(ys), _ptr(out), xs.size), 'sr_jomega_f64')
        return out

    def relax(self, model, D, omega, f_DD, f_CSA, time_fact, gamma_ratio, S2, C, tau, nComps,
              binvecs=None, weights=None, resvecs=None, noe_mode=0, want_J=False, weights_dev_ptr=None, want_stats=False):
        """model 0 direct / 1 sphere (D=[Diso]) / 2 symmetric top (D=[Dpar, Dperp]).
        Symmetric top: either `binvecs` (B,3) shared by all residues with optional `weights` (nRes,B), or
        `resvecs` (nRes,3), one vector per residue.  Returns out (E, nRes, 4, 2) = [R1,R2,NOE,rho] x
        [mean, sigma] and J (E, nRes, 5, 2) or None."""
        omega = _f64(np.atleast_2d(omega))
        E = omega.shape[0]
        S2 = _f64(S2)
        nRes = S2.size
        C = _f64(np.atleast_2d(C))
        tau = _f64(np.atleast_2d(tau))
        Kmax = C.shape[1]
        f_DD = _f64(np.broadcast_to(f_DD, (E,)))
        f_CSA = _f64(np.broadcast_to(f_CSA, (E, nRes)))
        time_fact = _f64(np.broadcast_to(time_fact, (E,)))
        gamma_ratio = _f64(np.broadcast_to(gamma_ratio, (E,)))
        nc = np.ascontiguousarray(nComps, dtype=np.int32)
        Dd = None if D is None else _f64(np.atleast_1d(D))
        B = 0
        bv = None
        w = None
        if model == 2:
            if binvecs is not None:
                bv = _f64(binvecs)
                B = bv.shape[0]
                if weights is not None:
                    w = _f64(weights)
                    if w.shape != (nRes, B):
                        raise ValueError('weights must be (nRes, B)')
            elif resvecs is not None:
                bv = _f64(resvecs)
                if bv.shape != (nRes, 3):
                    raise ValueError('resvecs must be (nRes, 3)')
            else:
                raise ValueError('symmetric-top model needs binvecs or resvecs')
        out = np.empty((E, nRes, 4, 2))
        J = np.empty((E, nRes, 5, 2)) if want_J else None
        stats = np.empty((E, nRes, 12)) if want_stats else None
        check(self.lib.sr_jomega_relax_f64(self.h, int(model), _ptr(Dd), E, _ptr(omega), _ptr(f_DD), _ptr(f_CSA),
                                           _ptr(time_fact), _ptr(gamma_ratio), nRes, Kmax, _ptr(S2), _ptr(C), _ptr(tau),
                                           _ptr(nc), B, _ptr(bv), weights_dev_ptr if weights_dev_ptr else _ptr(w),
                                           1 if weights_dev_ptr else 0, int(noe_mode), _ptr(out), _ptr(J), _ptr(stats)),
              'sr_jomega_relax_f64')
        if want_stats:
            return out, J, stats
        return out, J


_default = {}


def _close_default_contexts():
    for c in list(_default.values()):
        try:
            c.close()
        except Exception:
            pass
    _default.clear()


import atexit as _atexit      # noqa: E402
_atexit.register(_close_default_contexts)      # teardown while the HIP runtime is still alive, not from __del__ at exit


def default_context(device=None):
    """Process-wide context per device (created on first use).  Without an explicit device: SPINRELAX_DEVICE, else the
    LOCAL_RANK torchrun sets (one rank per GPU), else 0."""
    if device is None:
        import os
        device = int(os.environ.get('SPINRELAX_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    if device not in _default or _default[device].h is None:
        _default[device] = Context(device)
    return _default[device]


class ResidentVectors:
    """One rank's columns [v0, v0 + nV) of a (frames, vectors, 3) float32 array on the device: appended chunk by chunk
    (strided host copies of just those columns, or straight from coordinates through the GPU front end), packed once,
    then read by C(t) and by the rotation + histogram pass.  sr_vectors_* of include/spinrelax_hip.h."""

    def __init__(self, ctx, nV, capacity=0):
        self.ctx = ctx
        self.nV = int(nV)
        self.h = ctx.lib.sr_vectors_create(ctx.h, self.nV, int(capacity))
        if not self.h:
            raise SpinRelaxHipError('sr_vectors_create failed: %s' % _lib.last_error())

    def close(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.lib.sr_vectors_destroy(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def frames(self):
        return int(self.ctx.lib.sr_vectors_frames(self.h))

    def append(self, vecs, v0=0):
        """n more frames from a host array (n, Vtot, 3); only columns [v0, v0 + nV) are copied to the device"""
        vecs = _f32(vecs)
        if vecs.ndim != 3 or vecs.shape[2] != 3:
            raise ValueError('vecs must be (frames, vectors, 3)')
        check(self.ctx.lib.sr_vectors_append_f32(self.ctx.h, self.h, _ptr(vecs), vecs.shape[0], vecs.shape[1], int(v0)),
              'sr_vectors_append_f32')
        return self

    def truncate(self, n_frames):
        check(self.ctx.lib.sr_vectors_truncate(self.ctx.h, self.h, int(n_frames)), 'sr_vectors_truncate')

    def append_pinned(self, host_addr, n, Vtot, v0=0):
        """n more frames from PAGE-LOCKED host memory at `host_addr` ((n, Vtot, 3) float32): one asynchronous copy on the
        context's current stream, no staging pass; the memory must stay unchanged until the stream has passed this point"""
        check(self.ctx.lib.sr_vectors_append_f32(self.ctx.h, self.h, ctypes.c_void_p(int(host_addr)), int(n), int(Vtot), int(v0)),
              'sr_vectors_append_f32')
        return self

    def device_ptr(self):
        """device address of the (frames, nV, 3) float32 array; work queued on the context's stream behind this call sees
        every appended frame"""
        p = self.ctx.lib.sr_vectors_frame_major_dev(self.ctx.h, self.h)
        if not p:
            raise SpinRelaxHipError('sr_vectors_frame_major_dev failed: %s' % _lib.last_error())
        return int(p)

    def download(self, f0=0, n=None):
        n = self.frames - f0 if n is None else n
        out = np.empty((n, self.nV, 3), dtype=np.float32)
        check(self.ctx.lib.sr_vectors_download_f32(self.ctx.h, self.h, int(f0), int(n), _ptr(out)), 'sr_vectors_download_f32')
        return out

    def ct(self, R, F, chunk_start=None, mode=0):
        """calculate_Ct_Palmer (calculate-Ct-from-traj.py:200-238) of the resident vectors: Ct, dCt (F//2, nV) float64"""
        L = F // 2
        Ct = np.empty((L, self.nV))
        dCt = np.empty((L, self.nV))
        cs = None if chunk_start is None else np.ascontiguousarray(chunk_start, dtype=np.int64)
        if cs is not None and cs.shape != (R,):
            raise ValueError('chunk_start must have R entries')
        check(self.ctx.lib.sr_vectors_ct_f32(self.ctx.h, self.h, int(R), int(F), _ptr(cs), int(mode), _ptr(Ct), _ptr(dCt)),
              'sr_vectors_ct_f32')
        return Ct, dCt

    def ct_sums(self, R, F, chunk_start=None, mode=0):
        """raw sums S[v, r, d-1] = sum_j (u_j . u_{j+d})^2 of the R chunks held, (nV, R, F//2) float64 (replicate sharding)"""
        sums = np.empty((self.nV, R, F // 2))
        cs = None if chunk_start is None else np.ascontiguousarray(chunk_start, dtype=np.int64)
        check(self.ctx.lib.sr_vectors_ct_sums_f32(self.ctx.h, self.h, int(R), int(F), _ptr(cs), int(mode), _ptr(sums)),
              'sr_vectors_ct_sums_f32')
        return sums

    def hist(self, q, edges_phi, edges_cos, block_len=0, N_hist=0, want_outer=True):
        """rotation + Lambert histogram + vector sums + per-block outer-product sums (calculate-Ct-from-traj.py:541-646) of
        the first N_hist frames (0 = all): hist (nV, nphi, ncos), vecsum (nV, 3), outer (nB, nV, 6)"""
        ep = _f64(edges_phi)
        ec = _f64(edges_cos)
        nphi, ncos = ep.size - 1, ec.size - 1
        qq = None if q is None else _f64(q)
        N = N_hist if N_hist and N_hist > 0 else self.frames
        hist = np.empty((self.nV, nphi, ncos))
        vecsum = np.empty((self.nV, 3))
        Fb = block_len if (block_len and 0 < block_len <= N) else N
        outer = np.empty((N // Fb, self.nV, 6)) if want_outer else None
        check(self.ctx.lib.sr_vectors_hist_f32(self.ctx.h, self.h, int(N), _ptr(qq), _ptr(ep), nphi, _ptr(ec), ncos, _ptr(hist),
                                               _ptr(vecsum), _ptr(outer), int(block_len or 0)), 'sr_vectors_hist_f32')
        return hist, vecsum, outer


def append_xyz(ctx, lab, fit, xyz, indexX, indexH, fit_indices=None, ref_xyz=None):
    """obtain_XHvecs (+ centring and superposition when fit_indices / ref_xyz are given) of one chunk of coordinates
    (frames, atoms, 3) on the GPU, appended to the ResidentVectors objects `lab` and / or `fit` (either may be None);
    indexX / indexH are THIS rank's slice of the selections.  calculate-Ct-from-traj.py:64-86, 462-470."""
    xyz = _f32(xyz)
    iX = np.ascontiguousarray(indexX, dtype=np.int32)
    iH = np.ascontiguousarray(indexH, dtype=np.int32)
    fi = None if fit_indices is None else np.ascontiguousarray(fit_indices, dtype=np.int32)
    rx = None if ref_xyz is None else _f32(ref_xyz)
    check(ctx.lib.sr_vectors_append_xyz_f32(ctx.h, None if lab is None else lab.h, None if fit is None else fit.h, _ptr(xyz),
                                            xyz.shape[0], xyz.shape[1], _ptr(iX), _ptr(iH), iX.size, _ptr(fi),
                                            0 if fi is None else fi.size, _ptr(rx)), 'sr_vectors_append_xyz_f32')
