/*
 * spinrelax_hip.h -- C ABI of libspinrelax_hip.so: the MI355X (gfx950) implementation of SpinRelax's
 * bond-vector autocorrelation -> spectral density -> R1/R2/NOE hot path.
 *
 * The reference (zharmad/SpinRelax) has no FFI for this path: its boundary is four CLI scripts,
 * their file formats, a handful of Python functions and ONE native symbol, the numpy ufunc
 * npufunc.Jomega (Jomega/Jomega.c:30-156).  Each entry point below names the reference
 * function (file:line, relative to the reference root) whose work it replaces; INTEGRATION.md shows
 * the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns int: 0 = ok, negative = error; text via sr_last_error();
 *   - plain pointers and sizes only; "_dev" entry points take DEVICE pointers and enqueue work on the
 *     context's stream without synchronising (outputs are valid after sr_sync or a stream sync);
 *     entry points without "_dev" take HOST pointers, stage through the context's workspace and
 *     block until the result is in the caller's buffer;
 *   - one sr_ctx per GPU per thread; no internal locking; no callbacks or exceptions cross the ABI;
 *   - the context owns growable work areas (raw C(t) sums, histogram counters, staging buffers).  They are
 *     tied to no stream: sr_pack_soa*, sr_ct_palmer_f32_dev, sr_rotate_hist_f32_dev and every host-pointer entry
 *     point must be issued on ONE stream at a time (sr_set_stream between them is fine once the previous work is
 *     ordered before the next, e.g. same stream).  sr_expfit_order_search_f64_dev (with its `work` argument),
 *     sr_expfit_lm_f64_dev (ditto), sr_jomega_relax_f64_dev and sr_transpose_f64_dev use caller memory only and
 *     may run on several streams at once -- that is how spinrelax_amd/pipeline.py overlaps batches;
 *   - small HOST tables handed to "_dev" entry points (atom index lists, lag lists, chunk starts, a quaternion, bin edges)
 *     are validated before anything is queued and are consumed when the call returns: the caller may free them at once;
 *   - there is no CPU fallback: without a gfx950 device sr_create fails.
 */
#ifndef SPINRELAX_HIP_H
#define SPINRELAX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sr_ctx sr_ctx;

#define SR_ABI_VERSION 9

/* ---- context, memory, timing ------------------------------------------------------------- */
int          sr_abi_version(void);
/* sha256 (hex, first 16 characters) of the sources and compiler flags this library was built from (spinrelax_amd/build.py
 * computes it and compiles it in); "unknown" for a build that bypassed build.py.  bench.py compares it with the id stored
 * in the committed PMC profile to tell whether per-launch counter figures still belong to the kernels that ran. */
const char  *sr_build_id(void);
const char  *sr_last_error(void);
sr_ctx      *sr_create(int device);                 /* NULL on failure (see sr_last_error)        */
void         sr_destroy(sr_ctx *);
int          sr_set_stream(sr_ctx *, void *hip_stream);   /* hipStream_t; NULL = default stream  */
int          sr_sync(sr_ctx *);
/* Tuning knobs: "fit_waves" = wavefronts per residue in the fits (1, 2 or 4; default 2: four workgroups per CU -- the
 * model-order search is then limited by its registers alone -- and the shortest time per batch with the chip full, 0.40 ms for the
 * 512-residue benchmark batch against 0.60 at 4 and 0.82 at 1; a lone launch lasts 6.9 ms at 2, 6.3 at 4, 24.7 at 1; results are
 * bit-identical only between runs with the same value), "fit_lds" = 1/0 keep
 * a residue's t, C(t), 1/sigma in LDS, "fit_geo" = 1/0 (default 1): when a residue's time axis is a uniform grid (t[l] = t[0] + l dt
 * to 8 ulp -- checked per residue on the device; any other axis always takes exp() per point) the fit kernels form exp(-t/tau) at
 * the points a thread owns by multiplication, exp() once per thread: within 1e-15 of exp() per point, 17 % less time per batch
 * with the chip full; 0 = exp() per point whatever the axis, "ct_fft" = formulation of kernel 1 where the chunk length allows: 3 (default) the
 * FLOAT32 real-input FFT for 4096 < F + L <= 8192 (the reference's own arithmetic type; C(t) to 4e-8) and the float64 complex
 * FFT below, 4 float32 transforms for every 1024 < F + L <= 8192, 2 the float64 real-input FFT for 4096 < F + L <= 8192 (C(t) to
 * 1e-15), 1 the float64 complex FFT everywhere, 0 always direct;
 * "ct_traceless" = 1/0 (default 0): the real-input FFT kernel for F <= 4096 transforms the five traceless components of
 * u (x) u and takes the trace term from a scan of |u|^2 - 1 (one transform fewer; series that are not unit vectors fall back
 * to six inside the kernel) -- 4 % faster alone, 3 % slower per step inside the pipeline, same results to 1e-13. */
int          sr_set_option(sr_ctx *, const char *name, int value);
/* Streams that partition the chip.  The fits of fitting_Ct_functions.py:278-345 are a latency chain of small
 * launches; queued behind a C(t) launch that fills every CU they starve (queue priority does not pre-empt
 * resident workgroups).  sr_stream_create returns a hipStream_t whose kernels may only run on the CUs whose bit is
 * set in cu_mask (n_words 32-bit words, bit i of word w = CU 32*w+i in the runtime's numbering; NULL / 0 words =
 * all CUs); priority as hipStreamCreateWithPriority (lower = more urgent, 0 = default). */
int          sr_stream_create(sr_ctx *, const uint32_t *cu_mask, int n_words, int priority, void **stream_out);
int          sr_stream_destroy(sr_ctx *, void *hip_stream);
/* Signals: 32-bit values in memory that a KERNEL (or sr_stream_write_signal) sets and a stream waits for without the host
 * (hipStreamWaitValue32 on hipMallocSignalMemory).  sr_stream_wait_signal holds back everything queued afterwards on the
 * context's stream until *sig >= value; always pair a kernel-side release with an sr_stream_write_signal of the same value
 * behind that kernel, so that the wait ends at the latest when the kernel has finished.  NULL when the device cannot do it.
 * ORDERING RULE (enforced): the release must be SUBMITTED before the wait is queued -- first the releasing launch and its
 * sr_stream_write_signal(sig, v) on their stream, then sr_stream_wait_signal(sig, v) on the waiting stream.  Streams of one
 * priority share a few hardware queues; a wait queued ahead of its own release on a shared queue sits in front of it and never
 * ends.  sr_stream_wait_signal returns -6 when no release of `value` (or more) has been submitted for the signal yet. */
uint32_t    *sr_signal_alloc(sr_ctx *);
int          sr_signal_free(sr_ctx *, uint32_t *sig);
int          sr_stream_wait_signal(sr_ctx *, uint32_t *sig, uint32_t value);
int          sr_stream_write_signal(sr_ctx *, uint32_t *sig, uint32_t value);
int          sr_device_info(sr_ctx *, int *n_cu, int64_t *hbm_bytes, int *lds_per_cu, char *name, int name_len);
void        *sr_malloc(sr_ctx *, size_t bytes);     /* device memory                              */
int          sr_free(sr_ctx *, void *dev_ptr);
int          sr_memcpy_h2d(sr_ctx *, void *dev_dst, const void *host_src, size_t bytes);
int          sr_memcpy_d2h(sr_ctx *, void *host_dst, const void *dev_src, size_t bytes);
int          sr_memset(sr_ctx *, void *dev_dst, int value, size_t bytes);
/* Asynchronous copies on the context's stream (no synchronisation; the host buffer should be pinned: sr_host_alloc) and
 * page-locked host memory owned by the library.  A pipeline that keeps several batches in flight returns each batch's
 * results with these, on that batch's own stream -- no other runtime keeps per-stream state about the copies, so the
 * streams can be destroyed deterministically (sr_stream_destroy) when the pipeline closes. */
int          sr_memcpy_d2h_async(sr_ctx *, void *host_dst, const void *dev_src, size_t bytes);
int          sr_memcpy_h2d_async(sr_ctx *, void *dev_dst, const void *host_src, size_t bytes);
void        *sr_host_alloc(sr_ctx *, size_t bytes);              /* hipHostMalloc; NULL on failure */
int          sr_host_free(sr_ctx *, void *host_ptr);
int          sr_device_sync(sr_ctx *);                           /* every stream of the context's device */
/* HIP events on the context's stream (the stream the kernels are launched on). */
int          sr_timer_start(sr_ctx *);
int          sr_timer_stop_ms(sr_ctx *, float *elapsed_ms);     /* synchronises on the stop event */

/* ---- kernel 0: frame-major -> per-vector planes -------------------------------------------
 * Input is the layout the reference holds vectors in, (nframes, nvectors, 3) float32
 * (obtain_XHvecs, calculate-Ct-from-traj.py:64-86).  Output "SoA": soa[(v*3 + c)*Npad + n] for the
 * vector range [v0, v0+nV): the shard a GPU owns.  Npad >= N, Npad % 4 == 0. */
int sr_pack_soa_f32_dev(sr_ctx *, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                        float *soa, int64_t Npad);
/* The same with per-frame de-tumbling folded in: frame n is rotated by the unit
 * quaternion quat[n] = (w, x, y, z) -- rotate_vector_simd(v, q[:, None, :]), transforms3d_supplement.py:270-296, in
 * float64 -- before it is rounded into the float32 planes.  quat: DEVICE pointer, (N, 4) float64, already
 * normalised (vecnorm_NDarray).  With q = conj(q_orient(t)) from a PLUMED colvar-qorient file
 * (plumedcolvario.py:24-81) the planes hold body-frame vectors: lab-frame vectors + orientation trajectory in,
 * C(t) of the internal motion out, no superposition step (calculate-Ct-from-traj.py:466-467) needed. */
int sr_pack_soa_rot_f32_dev(sr_ctx *, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                            const double *quat, float *soa, int64_t Npad);

/* ---- kernel 1: Palmer-chunked P2 autocorrelation ------------------------------------------
 * Replaces calculate_Ct_Palmer (calculate-Ct-from-traj.py:200-238) after reformat_vecs_by_tau
 * (:245-275).  Chunk r covers frames [chunk_start[r], chunk_start[r]+F) of the packed planes
 * (chunk_start == NULL means r*F).  For delta = 1..L, L = F/2:
 *     p[r,v]      = (1/(F-delta)) * sum_j ( 1.5 (u[r,j,v].u[r,j+delta,v])^2 - 0.5 )
 *     Ct[d-1,v]   = mean_r p ;   dCt[d-1,v] = std_r(p, ddof=0) / (sqrt(R) - 1)
 * Ct, dCt are (L, nV) float64, row-major -- the reference's (nDeltas, nResidues).
 * mode 0 (production): the fastest formulation for the chunk length --
 *           1024 < F + L <= 8192: Wiener-Khinchin, (u.u')^2 = sum_c w_c a_c a_c' with a_c products of two components, one
 *           workgroup per (chunk, vector) with the whole transform in LDS.  4096 < F + L <= 8192 (cfg3 / cfg4): float32
 *           transforms of the five mean-removed traceless components, the mean terms restored in float64 (k_ct_rfft32:
 *           C(t) to ~4e-8, the class of the direct kernel and of the reference's own float32); shorter chunks: float64
 *           transforms (~1e-15).  sr_set_option("ct_fft", 2) selects float64 transforms everywhere, 0 disables the
 *           formulation;
 *           otherwise: direct shifted products, float32 dot products and short float32 partial sums folded into
 *           float64 (accurate to ~1e-8);
 * mode 1: direct shifted products, every product and sum in float64 (validation path).
 * psum (optional, may be NULL): (nV, R, Lp) float64 raw sums  sum_j (u.u')^2, Lp = sr_ct_psum_stride(F). */
int64_t sr_ct_psum_stride(int64_t F);
int64_t sr_ct_max_frames_per_chunk(sr_ctx *);
/* the two halves of sr_ct_palmer_f32_dev as separate launches: raw sums per (vector, chunk, lag) into caller memory
 * (psum: (nV, R, Lp) float64, required here), then mean / std over the chunks (calculate-Ct-from-traj.py:226-228).
 * A pipeline with several batches in flight gives every batch its own psum; timing the first call alone gives the
 * duration of the dominant kernel. */
int sr_ct_palmer_sums_f32_dev(sr_ctx *, const float *soa, int64_t Npad, int64_t R, int64_t F, int64_t nV,
                              const int64_t *chunk_start_host, int mode, double *psum);
int sr_ct_finalize_f64_dev(sr_ctx *, const double *psum, int64_t R, int64_t F, int64_t nV, double *Ct, double *dCt);
/* the same launch also leaves the transposed copies CtT, dCtT (nV, L) the fits read (fitting_Ct_functions.py:149 works
 * per residue); both NULL = not wanted. */
int sr_ct_finalize_t_f64_dev(sr_ctx *, const double *psum, int64_t R, int64_t F, int64_t nV, double *Ct, double *dCt,
                             double *CtT, double *dCtT);
int sr_ct_palmer_f32_dev(sr_ctx *, const float *soa, int64_t Npad, int64_t R, int64_t F, int64_t nV,
                         const int64_t *chunk_start_host, int mode,
                         double *psum_ws, double *Ct, double *dCt);
/* host-buffer form: vecs is (N, Vtot, 3) float32 on the host, N >= R*F when chunk_start is NULL. */
int sr_ct_palmer_f32(sr_ctx *, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                     int64_t R, int64_t F, const int64_t *chunk_start_host, int mode,
                     double *Ct, double *dCt);

/* ---- kernel 2: quaternion rotation + Lambert-cylindrical histogram + mean vector + S2 ------
 * Replaces rotate_vector_simd (transforms3d_supplement.py:270-296), xyz_to_rtp
 * (general_maths.py:118-158), the per-bond np.histogramdd loop (calculate-Ct-from-traj.py:600-626),
 * the mean vector (:579-583) and calculate_S2_by_outerProduct (:96-145) in one pass over the planes.
 *   q           : 4 doubles (w,x,y,z), normalised inside like the reference; NULL = no rotation;
 *   edges_phi   : nphi+1 doubles, edges_cos: ncos+1 doubles -- the numpy.linspace edges; binning is
 *                 numpy's: searchsorted(edges, x, 'right')-1, last edge inclusive, outside/NaN dropped;
 *   hist        : (nV, nphi, ncos) float64 counts (the dtype the reference stores);
 *   vecsum      : (nV, 3) float64 sum over the first N frames of the rotated vectors (may be NULL);
 *   outer       : (nBlocks, nV, 6) float64 sums of xx,yy,zz,xy,xz,yz per block of block_len frames
 *                 (may be NULL; nBlocks = N_used / block_len). */
int sr_rotate_hist_f32_dev(sr_ctx *, const float *soa, int64_t Npad, int64_t N, int64_t nV,
                           const double *q_host, const double *edges_phi_host, int nphi,
                           const double *edges_cos_host, int ncos,
                           double *hist, double *vecsum, double *outer, int64_t block_len);
int sr_rotate_hist_f32(sr_ctx *, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                       const double *q, const double *edges_phi, int nphi, const double *edges_cos, int ncos,
                       double *hist, double *vecsum, double *outer, int64_t block_len);
/* rotated vectors themselves, float64 (N, nV, 3) like the reference returns (for --vecDist output). */
int sr_rotate_vectors_f32(sr_ctx *, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                          const double *q, double *out);
/* one quaternion per frame: quat (N, 4) float64 host array, already normalised; out (N, nV, 3) float64. */
int sr_rotate_vectors_perframe_f32(sr_ctx *, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                                   const double *quat, double *out);

/* ---- resident bond vectors: upload once, pack once (a rank owns a vector range) --------------
 * The reference keeps vecXH / vecXHfit in host arrays (calculate-Ct-from-traj.py:464-498) and reads them for C(t) of the
 * lab-frame vectors (:527), C(t) of the fitted ones (:530) and the rotation + histogram + mean vector + S2 pass (:541-646).
 * An sr_vectors object holds ONE rank's columns [v0, v0 + nV) of such an array on the device:
 *   sr_vectors_create        nV vectors, room for capacity_frames frames (grows when more are appended); NULL on failure
 *   sr_vectors_append_f32    n more frames from a HOST array (n, Vtot, 3): only the bytes of columns [v0, v0 + nV) cross
 *                            PCIe (row pitch 12 Vtot, width 12 nV), through two pinned staging buffers; synchronous with
 *                            respect to the host array, asynchronous on the device.  A PAGE-LOCKED source (sr_host_alloc,
 *                            hipHostMalloc / hipHostRegister) is read by the copy engine directly -- one asynchronous (2-D)
 *                            copy, no staging pass; the caller then keeps it unchanged until the stream has passed the call
 *   sr_vectors_frame_major_dev  device address of the (frames, nV, 3) array held (NULL on failure); work queued on the
 *                            context's stream behind this call sees every appended frame (the feed of a device pipeline)
 *   sr_vectors_append_dev    n more frames from a DEVICE array (n, nV, 3) (e.g. an output of sr_xh_vectors_f32_dev)
 *   sr_vectors_append_xyz_f32  n more frames straight from HOST coordinates: obtain_XHvecs (+ centre / superpose) of
 *                            sr_xh_vectors_f32_dev for the bonds idxX[i] -> idxH[i], i < nV (the rank's slice of the
 *                            selections), written into `lab` and / or `fit` (either may be NULL) -- the streaming form of
 *                            the reference's --split loop (:426-453): the host never holds more than one chunk
 *   sr_vectors_truncate      keep the first n_frames frames (reformat_vecs_by_tau drops the tail of every file that does not fill
 *                            a block of memory time, calculate-Ct-from-traj.py:259-272: a streamed file is cut when it ends)
 *   sr_vectors_download_f32  frames [f0, f0 + n) back to the host as (n, nV, 3) (the --vecDist outputs need them)
 *   sr_vectors_ct_f32        kernel 0 (once per object) + kernel 1: as sr_ct_palmer_f32, Ct / dCt (F/2, nV) on the host
 *   sr_vectors_hist_f32      kernel 0 (once per object) + kernel 2 over the first N_hist frames (<= 0: all): as
 *                            sr_rotate_hist_f32
 * Stream ordering: an append leaves its copies in flight on the context's current stream and records an event behind
 * them; every consumer (ct, ct_sums, hist, download, frame_major_dev) AND every further append first makes ITS current stream
 * wait for that event, so sr_set_stream between an append and the next use or the next append loses no ordering (appends on
 * different streams chain; the growth copy of a full object runs behind the frames it copies).  chunk_start_host tables are range-checked against the frames held.
 * sr_counter(ctx, "h2d_vector_bytes" | "vector_uploads"): bytes and calls of sr_vectors_append_f32 since sr_create (the
 * host-pointer entry points sr_ct_palmer_f32 / sr_rotate_hist_f32 go through it too). */
typedef struct sr_vectors sr_vectors;
sr_vectors *sr_vectors_create(sr_ctx *, int64_t nV, int64_t capacity_frames);
void        sr_vectors_destroy(sr_ctx *, sr_vectors *);
int64_t     sr_vectors_frames(const sr_vectors *);
const float *sr_vectors_frame_major_dev(sr_ctx *, sr_vectors *);
int sr_vectors_truncate(sr_ctx *, sr_vectors *, int64_t n_frames);
int sr_vectors_append_f32(sr_ctx *, sr_vectors *, const float *vecs_host, int64_t n, int64_t Vtot, int64_t v0);
int sr_vectors_append_dev(sr_ctx *, sr_vectors *, const float *vecs_dev, int64_t n);
int sr_vectors_append_xyz_f32(sr_ctx *, sr_vectors *lab, sr_vectors *fit, const float *xyz_host, int64_t nFrames, int64_t nAtoms,
                              const int32_t *idxX, const int32_t *idxH, int nV, const int32_t *fit_idx, int nFit,
                              const float *ref_xyz);
int sr_vectors_download_f32(sr_ctx *, const sr_vectors *, int64_t f0, int64_t n, float *out_host);
int sr_vectors_ct_f32(sr_ctx *, sr_vectors *, int64_t R, int64_t F, const int64_t *chunk_start_host, int mode,
                      double *Ct, double *dCt);
/* Fewer vectors than GPUs: ranks own ranges of CHUNKS instead (last paragraph).  A rank computes the
 * raw sums S[v][r][d-1] = sum_j (u_j . u_{j+d})^2 of ITS chunks, (nV, R, L) float64 on the host; the gathered (nV, R_total, L)
 * array goes through the same mean / std kernel that finishes a single-process run (calculate-Ct-from-traj.py:226-228: mean
 * and two-pass std over the replicates need every replicate's value, not running sums). */
int sr_vectors_ct_sums_f32(sr_ctx *, sr_vectors *, int64_t R, int64_t F, const int64_t *chunk_start_host, int mode, double *sums);
int sr_ct_finalize_sums_f64(sr_ctx *, const double *sums_host, int64_t R, int64_t F, int64_t nV, double *Ct, double *dCt);
int sr_vectors_hist_f32(sr_ctx *, sr_vectors *, int64_t N_hist, const double *q, const double *edges_phi, int nphi,
                        const double *edges_cos, int ncos, double *hist, double *vecsum, double *outer, int64_t block_len);
int sr_counter(sr_ctx *, const char *name, uint64_t *value);

/* ---- kernel 3b: multi-exponential C(t) model --------------------------------------------
 * Model of curvefit_exponential (fitting_Ct_functions.py:419-427): params = [C_1..C_K, tau_1..tau_K
 * (, S2)], P = 2K or 2K+1; even P: S2 = 1 - sum C.  resid = (model - C)/sigma (sigma may be NULL),
 * jac[i,l,p] = d resid[i,l] / d param[i,p] (analytic).  All arrays row-major float64. */
int sr_expfit_resjac_f64(sr_ctx *, const double *t /*[nRes,L]*/, const double *C, const double *sigma,
                         const double *params /*[nRes,P]*/, int nRes, int L, int P,
                         double *resid /*[nRes,L]*/, double *jac /*[nRes,L,P] or NULL*/);
/* Batched bounded least-squares fit, one workgroup per residue, whole solve on the device
 * (replaces the scipy curve_fit call of conduct_curve_fitting, fitting_Ct_functions.py:322-324:
 * bounds 0 <= C,S2 <= 1, 0 <= tau <= tau_max).  p0 in / popt out (nRes,P); pcov (nRes,P,P) is the
 * curve_fit covariance (J^T J)^-1 * 2 cost/(L-P); chisq = mean(resid_unweighted^2 / sigma)
 * (calc_chiSq, :272-276); status per residue: >0 converged (1 gtol, 2 ftol, 3 xtol), 0 max iterations,
 * <0 failure.  Host pointers. */
int sr_expfit_lm_f64(sr_ctx *, const double *t, const double *C, const double *sigma, int nRes, int L, int P,
                     const double *p0, double tau_max, int max_iter,
                     double *popt, double *pcov, double *chisq, int *status, int *n_iter);
/* device-pointer form; skip (nRes bytes, device, may be NULL): residues with skip[i] != 0 are left
 * untouched (used by the host-side model-order search, which stops residues individually);
 * max_iter < 0 selects the analytic Jacobian instead of scipy's 2-point differences;
 * work (device, nRes*2*L doubles, may be NULL = context-owned): residual scratch; pass one buffer per
 * concurrently running launch when fits are enqueued on several streams. */
int sr_expfit_lm_f64_dev(sr_ctx *, const double *t, const double *C, const double *sigma, int nRes, int L, int P,
                         const double *p0, double tau_max, int max_iter, const unsigned char *skip, double *work,
                         double *popt, double *pcov, double *chisq, int *status, int *n_iter);

/* The whole model-order search of optimised_curve_fitting (fitting_Ct_functions.py:278-304) in ONE launch, one
 * workgroup per residue: for every order in `orders` (numbers of parameters, e.g. 2,3,5,7,9; host array, at most 8)
 * the initial guess of initialise_for_fit_advanced (:359-374), the bounded fit of conduct_curve_fitting (:306-345),
 * its three quality flags and the accept / reject rule (chi_threshold = chiSqThreshold, 0.5 in the reference).
 * All other pointers are DEVICE pointers.
 *   tau_guess   (tau_guess_rows, sum orders[j]/2): the log-spaced tau guesses of :361-362 for every order, one after
 *               the other; tau_guess_rows = 1 (every residue has the same time axis) or nRes
 *   work        (nRes*L doubles) or NULL: only used when a residue (3*L doubles) does not fit into LDS
 *   popt, dP    (nOrders, nRes, Pmax) with Pmax = max(orders): optimum and sqrt(diag(pcov)) of every attempted order
 *   chisq       (nOrders, nRes) calc_chiSq of the fit, +inf when the solver failed
 *   status,nfev (nOrders, nRes); status = -100: order not attempted (the search had already stopped)
 *   best        (nRes) index into orders of the accepted model, -1 = no satisfactory fit
 *   sel_*       the accepted model with its components sorted by tau (sort_components :203-209): S2 (nRes),
 *               C and tau (nRes, Pmax/2; unused entries 0 and 1), chi (nRes, NaN if none), K (nRes) = components
 * The launch is asynchronous on the context's stream. */
int sr_expfit_order_search_f64_dev(sr_ctx *, const double *t, const double *C, const double *sigma, int nRes, int L,
                                   const int *orders, int nOrders, const double *tau_guess, int tau_guess_rows,
                                   double tau_max, double chi_threshold, double *work, double *popt, double *dP,
                                   double *chisq, int *status, int *nfev, int *best, double *sel_S2, double *sel_C,
                                   double *sel_tau, double *sel_chi, int *sel_K);
/* The same for SEVERAL batches of residues in one launch (the pipeline merges the searches of a group of trajectory shards:
 * the per-residue loop of calculate-fitted-Ct.py:161-178 over all of them).  A launch lasts as long as its slowest residue and a
 * residue's cost is not known beforehand (2 to 9 parameters, 20 to 900 function evaluations): merged launches keep the chip
 * full while the stragglers of every batch run.
 *   t_rows          1: one time axis (L values) shared by every residue; nRes: a row per residue as above
 *   dispatch_order  DEVICE, nRes residue indices, or NULL: workgroup b solves residue dispatch_order[b].  Results are stored
 *                   at the residue's own index whatever the order; a permutation spreads residues that are expensive for the
 *                   same reason (the same position in every batch) over the chip instead of over one part of it.
 *   tail_signal     NULL, or a signal (sr_signal_alloc): the launch's last workgroup stores tail_value there when it STARTS.
 *                   Workgroups start in index order, so from that moment every residue has been handed to a CU and the launch
 *                   only drains -- slots free up while the longest fits finish.  A stream that waits for the value
 *                   (sr_stream_wait_signal) starts its bandwidth-bound kernels exactly then. */
int sr_expfit_order_search_batched_f64_dev(sr_ctx *, const double *t, int t_rows, const double *C, const double *sigma, int nRes,
                                           int L, const int *orders, int nOrders, const double *tau_guess, int tau_guess_rows,
                                           double tau_max, double chi_threshold, const int *dispatch_order,
                                           uint32_t *tail_signal, uint32_t tail_value, double *work,
                                           double *popt, double *dP, double *chisq, int *status, int *nfev, int *best,
                                           double *sel_S2, double *sel_C, double *sel_tau, double *sel_chi, int *sel_K);
/* the same with HOST pointers throughout (staged through the context's work space, synchronous) */
int sr_expfit_order_search_f64(sr_ctx *, const double *t, const double *C, const double *sigma, int nRes, int L,
                               const int *orders, int nOrders, const double *tau_guess, int tau_guess_rows,
                               double tau_max, double chi_threshold, double *popt, double *dP, double *chisq,
                               int *status, int *nfev, int *best, double *sel_S2, double *sel_C, double *sel_tau,
                               double *sel_chi, int *sel_K);

/* ---- kernel 3a: J(omega) and R1/R2/NOE/rho ----------------------------------------------
 * sr_jomega_f64: elementwise x/(x*x+y*y), the double loop of Jomega/Jomega.c:49-66, evaluated on the
 * GPU (host pointers; n elements, both inputs already broadcast by the caller). */
int sr_jomega_f64(sr_ctx *, const double *x, const double *y, double *out, int64_t n);

/* Batched relaxation.  Replaces _obtain_R1R2NOErho / _obtain_Jomega
 * (calculate-relaxations-from-Ct.py:82-191) and the class API spinRelaxationR1/R2/NOE.eval
 * (spectral_densities.py:820-907) for E "experiments" (field settings) at once.
 *   model        0 direct transform (J_direct_transform, spectral_densities.py:2024-2033)
 *                1 rigid sphere, D[0] = Diso (J_combine_isotropic_exp_decayN, :2038-2050)
 *                2 symmetric top, D[0] = Dpar, D[1] = Dperp (J_combine_symmtop_exp_decayN, :2057-2077)
 *   omega        (E,5): [0, wX, wH-wX, wH, wH+wX] in 1/time-unit
 *   f_DD (E), f_CSA (E,nRes), time_fact (E), gamma_ratio (E) = gamma_H/gamma_X
 *   S2 (nRes), C/tau (nRes,Kmax) with nComps (nRes) valid entries -- already zeta-scaled
 *   binvecs (B,3) unit vectors shared by every residue (bin centres) or (nRes,3) when B == 0
 *   weights (nRes,B) or NULL (uniform); a DEVICE pointer when weights_on_device != 0 (the histogram
 *                kernel 2 left in HBM), otherwise a host pointer
 *   noe_mode     0: NOE from the per-vector R1 (old API, :1706/:1722)
 *                1: NOE from the vector-averaged R1 (new API, :881-892)
 *   out          (E, nRes, 4, 2): {R1,R2,NOE,rho} x {weighted mean, weighted sigma}; sigma = 0 when B == 0
 *   Jout         (E, nRes, 5, 2) weighted mean/sigma of J(omega) or NULL (the --Jomega output)
 *   stats        (E, nRes, 12) or NULL: sufficient statistics for CSA fitting.  The CSA enters the rates only
 *                through f_CSA (proportional to csa^2): R1 = a1 + f_CSA*b1, R2 = a2 + f_CSA*b2 per vector, so
 *                [a1, b1, Var a1, Cov(a1,b1), Var b1, a2, b2, Var a2, Cov(a2,b2), Var b2, N, Var N]
 *                (weighted means / (co)variances over the vectors, N = 6 J(wH+wX) - J(wH-wX)) give the
 *                weighted mean AND sigma of R1, R2 and the new-API NOE for ANY csa without touching the
 *                2 592 bins again (rsCSA optimiser, spectral_densities.py:1371-1382, 1430-1447).
 * Host pointers. */
int sr_jomega_relax_f64(sr_ctx *, int model, const double *D, int E, const double *omega, const double *f_DD,
                        const double *f_CSA, const double *time_fact, const double *gamma_ratio,
                        int nRes, int Kmax, const double *S2, const double *C, const double *tau, const int *nComps,
                        int B, const double *binvecs, const double *weights, int weights_on_device, int noe_mode,
                        double *out, double *Jout, double *stats);
/* Device-pointer form for a pipeline that never leaves the GPU: every array argument is a DEVICE pointer (D stays a
 * host array of 1 or 2 numbers), nComps[i] must be within 0..Kmax (not checked), S2 and C are multiplied by zeta on
 * load (the reference scales the fitted model by zeta first, calculate-relaxations-from-Ct.py:747-750; pass 1.0 for
 * already-scaled input); asynchronous on the context's stream. */
int sr_jomega_relax_f64_dev(sr_ctx *, int model, const double *D, int E, const double *omega, const double *f_DD,
                            const double *f_CSA, const double *time_fact, const double *gamma_ratio,
                            int nRes, int Kmax, double zeta, const double *S2, const double *C, const double *tau,
                            const int *nComps, int B, const double *binvecs, const double *weights, int noe_mode,
                            double *out, double *Jout, double *stats);

/* Residue-specific CSA search of the new class API: replaces the per-residue fmin_powell loop of
 * spinRelaxationExperiments.optimisation_loop_do_local_step / optimisation_loop_do_local_step's objective
 * (spectral_densities.py:1371-1382, 1430-1447).  One device thread per residue runs scipy's one-variable Powell / Brent
 * search (restated from scipy/optimize/_optimize.py, see sr_relax.hip) over the closed forms the 12 statistics of
 * sr_jomega_relax_f64 give:
 *   stats (E, nRes, 12); column[e] = 0 R1, 1 R2, 2 NOE; csa_prefactor[e]: f_CSA = csa^2 * csa_prefactor[e];
 *   noe_factor[e] = time_fact * gamma_B / gamma_A, f_DD[e]; target / dtarget (E, nRes): the measured value and its
 *   uncertainty (0 when the file has none) of residue i in experiment e, read only where cover[e, i] != 0;
 *   has_err: the model values carry a sigma (vector distribution given); csa0 (nRes) start values; step = the initial
 *   direction (dictStepSizes['rsCSA']); xtol, ftol as fmin_powell's (1e-4 both in the reference's call).
 * Outputs: csa (nRes) = the LAST evaluated CSA of each search (what the reference keeps), values / errors (E, nRes) at
 * that CSA for covered entries (0 elsewhere), fopt (nRes) Powell's final objective, nfev (nRes) objective calls.
 * Residues no experiment covers keep csa0 and nfev = 0.  Host pointers. */
int sr_rscsa_search_f64(sr_ctx *, int E, int nRes, const double *stats, const int *column, const double *csa_prefactor,
                        const double *noe_factor, const double *f_DD, const double *target, const double *dtarget,
                        const unsigned char *cover, int has_err, const double *csa0, double step, double xtol, double ftol,
                        double *csa, double *values, double *errors, double *fopt, int *nfev);

/* Per-residue CSA refinement of the legacy single-field mode `--opt new`: replaces the loop of fmin_powell calls over
 * optfunc_R1R2NOE_new (calculate-relaxations-from-Ct.py:210-258, 935-1000), axisymmetric diffusion + vector histogram only.
 * One workgroup per residue runs scipy's one-variable Powell search (as sr_rscsa_search_f64); every objective call evaluates
 * R1, R2 and the old-API NOE over the B histogram bins with the arithmetic of sr_jomega_relax_f64 (noe_mode 0), casts value and
 * sigma to float32 like the reference's datablock and returns mean_k (model_k - exp_k)^2 / (sigma_exp_k^2 + sigma_model_k^2).
 *   D = {Dpar, Dperp}; omega (5); f_DD; gammaB0_sq = (gamma_X B0)^2 (f_CSA = 2/15 csa^2 gammaB0_sq); time_fact; gamma_ratio;
 *   S2 (nRes), C / tau (nRes, Kmax), nComps (nRes): the fitted models, already scaled by zeta; binvecs (B, 3), weights (nRes, B);
 *   expt (nRes, 3, 2): measured R1, R2, NOE and their uncertainties; csa0 (nRes) start values; step = Powell's initial
 *   direction (1.0: scipy's identity default), xtol, ftol (1e-4), maxiter, maxfun (1000: scipy's N * 1000).
 * Outputs (nRes): csa = Powell's optimum, fopt = the objective there, nfev = objective calls.  Host pointers. */
int sr_legacy_csa_search_f64(sr_ctx *, const double *D, const double *omega, double f_DD, double gammaB0_sq, double time_fact,
                             double gamma_ratio, int nRes, int Kmax, const double *S2, const double *C, const double *tau,
                             const int *nComps, int B, const double *binvecs, const double *weights, const double *expt,
                             const double *csa0, double step, double xtol, double ftol, int maxiter, int maxfun, double *csa,
                             double *fopt, int *nfev);

/* ---- trajectory front end ----------------------------------------
 * Replaces obtain_XHvecs (calculate-Ct-from-traj.py:64-86) with vecnorm_NDarray (transforms3d_supplement.py:40-52) and the
 * MDTraj center_coordinates + superpose(ref, frame=0, atom_indices=fit_indices) step between its two calls (:466-467):
 *   xyz       (nFrames, nAtoms, 3) float32 coordinates as MDTraj holds them (traj.xyz);
 *   idxX/idxH nV atom indices each (the X and H selections, same order), HOST arrays in both forms;
 *   fit_idx   nFit atom indices to superpose on, ref_xyz (nAtoms, 3) float32 reference coordinates, HOST arrays
 *             (both may be NULL when neither vec_fit nor quat is requested);
 *   vec_lab   (nFrames, nV, 3) float32 unit vectors (x_H - x_X)/|.| in float32 arithmetic, 0/0 -> 0: bit-identical to
 *             the reference's numpy expression; may be NULL;
 *   vec_fit   (nFrames, nV, 3) float32: the same bond rotated by the frame's optimal (unweighted least-squares, proper)
 *             rotation onto the reference, float64 inside, normalised; may be NULL;
 *   quat      (nFrames, 4) float64 (w, x, y, z), w >= 0: that rotation; may be NULL.
 * Both outputs have the (frames, vectors, 3) layout sr_pack_soa_f32_dev / sr_ct_palmer_f32 consume. */
int sr_xh_vectors_f32_dev(sr_ctx *, const float *xyz, int64_t nFrames, int64_t nAtoms, const int32_t *idxX_host,
                          const int32_t *idxH_host, int nV, const int32_t *fit_idx_host, int nFit, const float *ref_xyz_host,
                          float *vec_lab, float *vec_fit, double *quat);
int sr_xh_vectors_f32(sr_ctx *, const float *xyz, int64_t nFrames, int64_t nAtoms, const int32_t *idxX, const int32_t *idxH,
                      int nV, const int32_t *fit_idx, int nFit, const float *ref_xyz, float *vec_lab, float *vec_fit,
                      double *quat);

/* ---- global rotational diffusion: difference-quaternion lag correlations ----------
 * Replaces the per-lag reductions of calculate-dq-distribution.py: obtain_self_dq (:102-109, dq_i = q_i^-1 * q_{i+d} with
 * quat_mult_simd / quat_invert / quat_reduce_simd of transforms3d_supplement.py:163-186, 219-227), average_LegendreP1quat
 * (:111-112), average_anisotropic_tensor (:118-126) and their *_chunk variants (:128-144) inside the main loop :554-609.
 *   q      (N, 4) float32 orientation quaternions (w, x, y, z), e.g. the q.w..q.z columns of colvar-qorient;
 *   lags   nlags frame offsets d (1 <= d < N), HOST array in both forms;
 *   nchunk number of sub-chunks (>= 1); for lag d, chunk c covers samples [nb*c, min(N-d, nb*(c+1))), nb = ceil((N-d)/nchunk);
 *   out    (nlags, nchunk, 7) float64: sums over the chunk's samples of xx, yy, zz, xy, xz, yz of v = vec(dq), then the
 *          number of samples.  <v (x) v> = sums / count, <1 - 2|v|^2> = 1 - 2 (xx+yy+zz)/count; the whole-trajectory values
 *          are the sums over the chunks.  Products and sums in float64 on the float32 input. */
int sr_dq_moments_f32_dev(sr_ctx *, const float *q, int64_t N, const int32_t *lags_host, int nlags, int nchunk, double *out);
int sr_dq_moments_f32(sr_ctx *, const float *q, int64_t N, const int32_t *lags, int nlags, int nchunk, double *out);
/* the same on float64 quaternions: the reference's gmx-rotmat route (rotmatrix_to_quaternion, calculate-dq-distribution.py:
 * 482-497) keeps float64, and its short-lag moments (|v| ~ 1e-2) would lose five digits in a float32 round trip */
int sr_dq_moments_f64_dev(sr_ctx *, const double *q, int64_t N, const int32_t *lags_host, int nlags, int nchunk, double *out);
int sr_dq_moments_f64(sr_ctx *, const double *q, int64_t N, const int32_t *lags, int nlags, int nchunk, double *out);

/* ---- small device utilities ---------------------------------------------------------------
 * out[c*rows + r] = in[r*cols + c] (float64, device pointers): C(t) leaves kernel 1 as (lags, vectors)
 * like the reference, the fit reads (residues, lags). */
int sr_transpose_f64_dev(sr_ctx *, const double *in, int64_t rows, int64_t cols, double *out);

/* ---- host-side text I/O of the reference's xmgrace-style files (no device work) ----------------------------------
 * run-all.bash passes C(t) between its scripts as text: `<prefix>_Ctint.dat` is written by print_sxylist
 * (general_scripts.py:275-290) and read back by load_sxydylist (:182-213); `_fittedCt.dat` carries "%8g %8g" rows
 * (fitting_Ct_functions.py:107-126).  For 512 residues x 2 048 lags that is 31 + 38 MB of text per run, and with the kernels
 * at milliseconds its formatting / parsing in Python was 90 % of the chain's wall time.  These entry points produce exactly
 * the bytes / doubles of the Python code (numpy's array printer for the (C, dC) pairs, strtod for reading); a positive
 * return code (or `regular` = 0) means "not the regular case -- use the Python path" and nothing has been written.
 *   sr_text_write_sxydy_f64   legend_lines / xstr: nsets / npts strings separated by '\0'; ydy (nsets, npts, 2)
 *   sr_text_format_g8_pairs   n rows "%8g %8g\n" into out (>= 32 n bytes); returns the bytes written; offsets[k] = byte position
 *                             of row bounds[k] (optional: where the caller cuts the text into blocks)
 *   sr_text_open_sxydy        parse a file; info6 = {regular, nsets, npts, has_dy, legends, bytes of the '\0'-joined legends};
 *                             sr_text_sxydy_get copies x, y, dy (nsets, npts) and the legends out; sr_text_close_sxydy frees */
typedef struct sr_sxydy sr_sxydy;
int       sr_text_write_sxydy_f64(const char *path, const char *header, int64_t nsets, int64_t npts, const char *legend_lines,
                                  const char *xstr, const double *ydy, int nthreads);
int64_t   sr_text_format_g8_pairs(const double *a, const double *b, int64_t n, char *out, int64_t out_bytes, int nthreads,
                                  const int64_t *bounds, int64_t nb, int64_t *offsets);
sr_sxydy *sr_text_open_sxydy(const char *path, const char *key, int nthreads);
void      sr_text_close_sxydy(sr_sxydy *);
int       sr_text_sxydy_info(const sr_sxydy *, int64_t *info6);
int       sr_text_sxydy_get(const sr_sxydy *, double *x, double *y, double *dy, char *legends);

#ifdef __cplusplus
}
#endif
#endif /* SPINRELAX_HIP_H */
