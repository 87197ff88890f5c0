"""
ctypes binding of libspinrelax_hip.so (include/spinrelax_hip.h).

There is deliberately NO CPU fallback: if the shared library is missing, cannot be loaded, or no
gfx950 device is present, every compute entry point raises SpinRelaxHipError.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libspinrelax_hip.so')


class SpinRelaxHipError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/spinrelax_hip.h one to one
SIGNATURES = {
    'sr_abi_version': (c_int, []),
    'sr_build_id': (c_char_p, []),
    'sr_last_error': (c_char_p, []),
    'sr_create': (c_void_p, [c_int]),
    'sr_destroy': (None, [c_void_p]),
    'sr_set_stream': (c_int, [c_void_p, c_void_p]),
    'sr_sync': (c_int, [c_void_p]),
    'sr_set_option': (c_int, [c_void_p, c_char_p, c_int]),
    'sr_stream_create': (c_int, [c_void_p, POINTER(ctypes.c_uint32), c_int, c_int, POINTER(c_void_p)]),
    'sr_stream_destroy': (c_int, [c_void_p, c_void_p]),
    'sr_text_write_sxydy_f64': (c_int, [c_char_p, c_char_p, c_int64, c_int64, c_char_p, c_char_p, c_void_p, c_int]),
    'sr_text_format_g8_pairs': (c_int64, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    'sr_text_open_sxydy': (c_void_p, [c_char_p, c_char_p, c_int]),
    'sr_text_close_sxydy': (None, [c_void_p]),
    'sr_text_sxydy_info': (c_int, [c_void_p, c_void_p]),
    'sr_text_sxydy_get': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'sr_signal_alloc': (c_void_p, [c_void_p]),
    'sr_signal_free': (c_int, [c_void_p, c_void_p]),
    'sr_stream_wait_signal': (c_int, [c_void_p, c_void_p, ctypes.c_uint32]),
    'sr_stream_write_signal': (c_int, [c_void_p, c_void_p, ctypes.c_uint32]),
    'sr_device_info': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int64), POINTER(c_int), c_char_p, c_int]),
    'sr_malloc': (c_void_p, [c_void_p, c_size_t]),
    'sr_free': (c_int, [c_void_p, c_void_p]),
    'sr_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'sr_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'sr_memset': (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    'sr_memcpy_d2h_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'sr_memcpy_h2d_async': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'sr_host_alloc': (c_void_p, [c_void_p, c_size_t]),
    'sr_host_free': (c_int, [c_void_p, c_void_p]),
    'sr_device_sync': (c_int, [c_void_p]),
    'sr_timer_start': (c_int, [c_void_p]),
    'sr_timer_stop_ms': (c_int, [c_void_p, POINTER(c_float)]),
    'sr_pack_soa_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64]),
    'sr_pack_soa_rot_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64]),
    'sr_ct_psum_stride': (c_int64, [c_int64]),
    'sr_ct_max_frames_per_chunk': (c_int64, [c_void_p]),
    'sr_ct_palmer_sums_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    'sr_ct_finalize_f64_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    'sr_ct_finalize_t_f64_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'sr_ct_palmer_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p]),
    'sr_ct_palmer_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                 c_void_p, c_int, c_void_p, c_void_p]),
    'sr_rotate_hist_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64]),
    'sr_rotate_hist_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                   c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64]),
    'sr_rotate_vectors_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    'sr_rotate_vectors_perframe_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    'sr_expfit_resjac_f64': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p, c_void_p]),
    'sr_expfit_lm_f64': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_double,
                                 c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'sr_expfit_lm_f64_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_double,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    # (ctx, t, C, sigma, nRes, L, orders, nOrders, tau_guess, tau_rows, tau_max, chi_thr, work, popt, dP, chisq, status,
    #  nfev, best, sel_S2, sel_C, sel_tau, sel_chi, sel_K)
    'sr_expfit_order_search_f64_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                               c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    # (ctx, t, t_rows, C, sigma, nRes, L, orders, nOrders, tau_guess, tau_rows, tau_max, chi_thr, dispatch_order, work, popt, ...)
    'sr_expfit_order_search_batched_f64_dev': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                                       c_void_p, c_int, c_double, c_double, c_void_p, c_void_p, ctypes.c_uint32,
                                                       c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p]),
    'sr_expfit_order_search_f64': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                           c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    # (ctx, model, D, E, omega, f_DD, f_CSA, time_fact, gamma_ratio, nRes, Kmax, zeta, S2, C, tau, nComps, B, binvecs,
    #  weights, noe_mode, out, Jout, stats)
    'sr_jomega_relax_f64_dev': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                        c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'sr_xh_vectors_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    'sr_xh_vectors_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    'sr_dq_moments_f32_dev': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    'sr_dq_moments_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    'sr_dq_moments_f64_dev': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    'sr_dq_moments_f64': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]),
    # (ctx, E, nRes, stats, column, csa_prefactor, noe_factor, f_DD, target, dtarget, cover, has_err, csa0, step, xtol, ftol,
    #  csa, values, errors, fopt, nfev)
    'sr_legacy_csa_search_f64': (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double,
                                         c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'sr_rscsa_search_f64': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    'sr_vectors_create': (c_void_p, [c_void_p, c_int64, c_int64]),
    'sr_vectors_destroy': (None, [c_void_p, c_void_p]),
    'sr_vectors_frames': (c_int64, [c_void_p]),
    'sr_vectors_frame_major_dev': (c_void_p, [c_void_p, c_void_p]),
    'sr_vectors_truncate': (c_int, [c_void_p, c_void_p, c_int64]),
    'sr_vectors_append_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64]),
    'sr_vectors_append_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    'sr_vectors_append_xyz_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                          c_void_p, c_int, c_void_p]),
    'sr_vectors_download_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'sr_vectors_ct_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    'sr_vectors_ct_sums_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    'sr_ct_finalize_sums_f64': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    'sr_vectors_hist_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_int64]),
    'sr_counter': (c_int, [c_void_p, c_char_p, POINTER(ctypes.c_uint64)]),
    'sr_transpose_f64_dev': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'sr_jomega_f64': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    'sr_jomega_relax_f64': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p]),
}

_lib = None
ABI_VERSION = 9
LIB_PATH = os.environ.get('SPINRELAX_HIP_LIB', LIB_PATH)      # alternative build of the same ABI


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so.7; /opt/rocm has another one with
    the same soname, and whichever is loaded first serves both this library and torch.  If this library came first
    it would bind the system runtime, torch would later bring up its bundled copy as a SECOND runtime, and that one
    finds no GPU ("No HIP GPUs are available").  So when torch is installed its runtime is loaded first (without
    importing torch); device pointers and streams are then interchangeable between the two."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.isfile(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load the shared library (once) and attach the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise SpinRelaxHipError(
            '%s not found: build it with `python -m spinrelax_amd.build` (needs hipcc). '
            'spinrelax_amd has no CPU fallback.' % LIB_PATH)
    _share_hip_runtime_with_torch()
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise SpinRelaxHipError('cannot load %s: %s' % (LIB_PATH, exc))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise SpinRelaxHipError('%s does not export %s (stale build?)' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.sr_abi_version() != ABI_VERSION:
        raise SpinRelaxHipError('ABI version mismatch: library %d, binding %d' % (lib.sr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def last_error():
    return load().sr_last_error().decode('utf-8', 'replace')


def check(rc, what):
    if rc != 0:
        raise SpinRelaxHipError('%s failed (%d): %s' % (what, rc, last_error()))
