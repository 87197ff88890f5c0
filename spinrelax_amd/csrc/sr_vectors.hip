// sr_vectors.hip -- resident bond vectors: the product path's "upload once, pack once".
//
// The reference holds the unit vectors of a run in host arrays (vecXH / vecXHfit, calculate-Ct-from-traj.py:464-498) and
// walks them three times: C(t) of the lab-frame vectors (:527), C(t) of the fitted ones (:530), rotation + histogram + mean
// vector + S2 of the fitted ones (:541-646).  A GPU rank owns a contiguous range of vectors; it
// needs exactly the columns [v0, v0 + nV) of the (frames, vectors, 3) array, once.  An sr_vectors object is that shard on
// the device: frames are APPENDED chunk by chunk (from host memory: a strided copy of the rank's columns through two pinned
// staging buffers, 12 * frames * nV bytes over PCIe and not a byte more; from device memory: what the trajectory front end
// sr_xh_vectors_f32_dev just produced), packed once into per-vector planes (kernel 0), and then read by kernel 1 and
// kernel 2 as often as the caller asks.  The host-pointer entry points sr_ct_palmer_f32 / sr_rotate_hist_f32 are this
// object used once.
#include "sr_internal.h"
#include <cstdlib>

struct sr_vectors {
    int64_t nV;            // vectors of this shard
    int64_t N;             // frames appended so far
    int64_t cap;           // frames the frame-major buffer can hold
    float *fm;             // (cap, nV, 3) frame-major, device
    float *soa;            // (nV, 3, Npad) planes, device; built by the first computation
    int64_t Npad;
    int packed;            // planes are current
    hipEvent_t ready;      // recorded behind the last append / pack on the stream that did it: a consumer on ANOTHER stream
                           // (sr_set_stream between an append and the next use) waits for it on the device
};

namespace {

constexpr size_t kStageBytes = (size_t)16 << 20;          // two pinned staging buffers of this size per context

int ensure_staging(sr_ctx *ctx)
{
    for (int i = 0; i < 2; ++i) {
        if (!ctx->stage[i]) {
            SR_HIP(hipHostMalloc(&ctx->stage[i], kStageBytes, hipHostMallocDefault));
            SR_HIP(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
            ctx->stage_busy[i] = 0;
        }
    }
    return 0;
}

int reserve(sr_ctx *ctx, sr_vectors *h, int64_t frames)
{
    if (frames <= h->cap) return 0;
    int64_t want = h->cap > 0 ? h->cap + h->cap / 2 : frames;
    if (want < frames) want = frames;
    float *p = nullptr;
    SR_HIP(hipMalloc((void **)&p, (size_t)want * h->nV * 3 * sizeof(float)));
    if (h->fm) {
        if (h->N > 0) {
            hipError_t e = hipMemcpyAsync(p, h->fm, (size_t)h->N * h->nV * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { (void)hipFree(p); SR_HIP(e); }
        } else {
            SR_HIP(hipStreamSynchronize(ctx->stream));
        }
        SR_HIP(hipFree(h->fm));
    }
    h->fm = p;
    h->cap = want;
    return 0;
}

int mark_ready(sr_ctx *ctx, sr_vectors *h)
{
    if (!h->ready) SR_HIP(hipEventCreateWithFlags(&h->ready, hipEventDisableTiming));
    SR_HIP(hipEventRecord(h->ready, ctx->stream));
    return 0;
}

/* the contents of h (frames and planes) are valid on ctx->stream behind this call, whichever stream wrote them */
int wait_ready(sr_ctx *ctx, sr_vectors *h)
{
    if (h->ready) SR_HIP(hipStreamWaitEvent(ctx->stream, h->ready, 0));
    return 0;
}

int pack(sr_ctx *ctx, sr_vectors *h)
{
    if (int rc = wait_ready(ctx, h)) return rc;
    if (h->packed) return 0;
    SR_REQUIRE(h->N > 0, -3, "sr_vectors: no frames appended");
    const int64_t Npad = sr_round_up(h->N, 64);
    if (!h->soa || Npad != h->Npad) {
        if (h->soa) {
            SR_HIP(hipStreamSynchronize(ctx->stream));
            SR_HIP(hipFree(h->soa));
            h->soa = nullptr;
        }
        SR_HIP(hipMalloc((void **)&h->soa, (size_t)h->nV * 3 * Npad * sizeof(float)));
        h->Npad = Npad;
    }
    int rc = sr_pack_soa_f32_dev(ctx, h->fm, h->N, h->nV, 0, h->nV, h->soa, Npad);
    if (rc) return rc;
    h->packed = 1;
    return mark_ready(ctx, h);
}

}  // namespace

extern "C" {

int sr_counter(sr_ctx *ctx, const char *name, uint64_t *value)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(name && value, -2, "sr_counter: null pointer");
    if (!strcmp(name, "h2d_vector_bytes")) { *value = ctx->h2d_bytes; return 0; }
    if (!strcmp(name, "vector_uploads")) { *value = ctx->h2d_calls; return 0; }
    sr_set_error("sr_counter: unknown counter '%s'", name);
    return -3;
}

sr_vectors *sr_vectors_create(sr_ctx *ctx, int64_t nV, int64_t capacity_frames)
{
    if (!ctx) { sr_set_error("null sr_ctx"); return nullptr; }
    if (nV < 1 || capacity_frames < 0) { sr_set_error("sr_vectors_create: bad sizes nV=%lld capacity=%lld", (long long)nV, (long long)capacity_frames); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sr_set_error("hipSetDevice failed"); return nullptr; }
    sr_vectors *h = (sr_vectors *)calloc(1, sizeof(sr_vectors));
    if (!h) { sr_set_error("sr_vectors_create: out of host memory"); return nullptr; }
    h->nV = nV;
    if (capacity_frames > 0 && reserve(ctx, h, capacity_frames) != 0) { free(h); return nullptr; }
    return h;
}

void sr_vectors_destroy(sr_ctx *ctx, sr_vectors *h)
{
    if (!h) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (h->fm) (void)hipFree(h->fm);
    if (h->soa) (void)hipFree(h->soa);
    if (h->ready) (void)hipEventDestroy(h->ready);
    free(h);
}

int64_t sr_vectors_frames(const sr_vectors *h) { return h ? h->N : -1; }

const float *sr_vectors_frame_major_dev(sr_ctx *ctx, sr_vectors *h)
{
    if (!ctx || !h) { sr_set_error("sr_vectors_frame_major_dev: null pointer"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sr_set_error("hipSetDevice failed"); return nullptr; }
    if (wait_ready(ctx, h) != 0) return nullptr;        // readers queued on ctx->stream behind this call see every appended frame
    return h->fm;
}

int sr_vectors_truncate(sr_ctx *ctx, sr_vectors *h, int64_t n_frames)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h != nullptr, -2, "sr_vectors_truncate: null pointer");
    SR_REQUIRE(n_frames >= 0 && n_frames <= h->N, -3, "sr_vectors_truncate: %lld of %lld frames", (long long)n_frames, (long long)h->N);
    if (n_frames != h->N) { h->N = n_frames; h->packed = 0; }
    return 0;
}

int sr_vectors_append_f32(sr_ctx *ctx, sr_vectors *h, const float *vecs, int64_t n, int64_t Vtot, int64_t v0)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h && vecs, -2, "sr_vectors_append_f32: null pointer");
    SR_REQUIRE(n >= 1 && Vtot >= 1 && v0 >= 0 && v0 + h->nV <= Vtot, -3,
               "sr_vectors_append_f32: bad shape n=%lld Vtot=%lld v0=%lld nV=%lld", (long long)n, (long long)Vtot, (long long)v0, (long long)h->nV);
    // chain behind whatever touched the object last, on whichever stream: an append on stream B after one on stream A (the single
    // `ready` event is re-recorded below) and the growth copy of reserve() both see the earlier frames complete
    if (int rc = wait_ready(ctx, h)) return rc;
    if (int rc = reserve(ctx, h, h->N + n)) return rc;
    const size_t row = (size_t)h->nV * 3 * sizeof(float);                 // bytes of this rank's columns in one frame
    const size_t pitch = (size_t)Vtot * 3 * sizeof(float);
    {
        // page-locked source (sr_host_alloc, hipHostMalloc, hipHostRegister): the DMA engine reads it directly -- one (2-D when the
        // rank owns a column range) asynchronous copy, no pass through the staging buffers.  The caller keeps the source alive
        // and unchanged until the stream has passed this point (as with any asynchronous copy).
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof(at));
        if (hipPointerGetAttributes(&at, vecs) == hipSuccess && at.type == hipMemoryTypeHost) {
            const char *src = reinterpret_cast<const char *>(vecs) + (size_t)v0 * 3 * sizeof(float);
            char *dst = reinterpret_cast<char *>(h->fm) + (size_t)h->N * row;
            if (row == pitch) SR_HIP(hipMemcpyAsync(dst, src, (size_t)n * row, hipMemcpyHostToDevice, ctx->stream));
            else SR_HIP(hipMemcpy2DAsync(dst, row, src, pitch, row, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            ctx->h2d_bytes += (unsigned long long)n * row;
            ctx->h2d_calls += 1;
            h->N += n;
            h->packed = 0;
            return mark_ready(ctx, h);
        }
        (void)hipGetLastError();                        // pageable memory: the attribute query reports an error, not a type
    }
    if (int rc = ensure_staging(ctx)) return rc;
    SR_REQUIRE(row <= kStageBytes, -3, "sr_vectors_append_f32: %lld vectors per frame exceed the staging buffer", (long long)h->nV);
    const int64_t rows_per_buf = (int64_t)(kStageBytes / row);
    const char *src = reinterpret_cast<const char *>(vecs) + (size_t)v0 * 3 * sizeof(float);
    char *dst = reinterpret_cast<char *>(h->fm) + (size_t)h->N * row;
    int b = 0;
    for (int64_t f0 = 0; f0 < n; f0 += rows_per_buf, b ^= 1) {
        const int64_t nr = n - f0 < rows_per_buf ? n - f0 : rows_per_buf;
        if (ctx->stage_busy[b]) {                                         // the copy that last used this buffer
            SR_HIP(hipEventSynchronize(ctx->stage_ev[b]));
            ctx->stage_busy[b] = 0;
        }
        char *st = reinterpret_cast<char *>(ctx->stage[b]);
        if (row == pitch) {
            memcpy(st, src + (size_t)f0 * pitch, (size_t)nr * row);
        } else {
            for (int64_t r = 0; r < nr; ++r) memcpy(st + (size_t)r * row, src + (size_t)(f0 + r) * pitch, row);
        }
        SR_HIP(hipMemcpyAsync(dst + (size_t)f0 * row, st, (size_t)nr * row, hipMemcpyHostToDevice, ctx->stream));
        SR_HIP(hipEventRecord(ctx->stage_ev[b], ctx->stream));
        ctx->stage_busy[b] = 1;
    }
    ctx->h2d_bytes += (unsigned long long)n * row;
    ctx->h2d_calls += 1;
    h->N += n;
    h->packed = 0;
    return mark_ready(ctx, h);               // the copies are still in flight on ctx->stream: a later use on another stream waits
}

int sr_vectors_append_dev(sr_ctx *ctx, sr_vectors *h, const float *vecs_dev, int64_t n)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h && vecs_dev, -2, "sr_vectors_append_dev: null pointer");
    SR_REQUIRE(n >= 1, -3, "sr_vectors_append_dev: bad frame count %lld", (long long)n);
    if (int rc = wait_ready(ctx, h)) return rc;
    if (int rc = reserve(ctx, h, h->N + n)) return rc;
    const size_t row = (size_t)h->nV * 3 * sizeof(float);
    SR_HIP(hipMemcpyAsync(reinterpret_cast<char *>(h->fm) + (size_t)h->N * row, vecs_dev, (size_t)n * row, hipMemcpyDeviceToDevice,
                          ctx->stream));
    h->N += n;
    h->packed = 0;
    return mark_ready(ctx, h);
}

int sr_vectors_append_xyz_f32(sr_ctx *ctx, sr_vectors *lab, sr_vectors *fit, const float *xyz, int64_t nFrames, int64_t nAtoms,
                              const int32_t *idxX, const int32_t *idxH, int nV, const int32_t *fit_idx, int nFit,
                              const float *ref_xyz)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(xyz && idxX && idxH && (lab || fit), -2, "sr_vectors_append_xyz_f32: null pointer");
    SR_REQUIRE(nFrames >= 1 && nAtoms >= 1 && nV >= 1, -3, "sr_vectors_append_xyz_f32: bad sizes");
    SR_REQUIRE((!lab || lab->nV == nV) && (!fit || fit->nV == nV), -3, "sr_vectors_append_xyz_f32: the objects hold another number of vectors");
    if (lab) if (int rc = reserve(ctx, lab, lab->N + nFrames)) return rc;
    if (fit) if (int rc = reserve(ctx, fit, fit->N + nFrames)) return rc;
    const size_t in_bytes = (size_t)nFrames * nAtoms * 3 * sizeof(float);
    float *xyz_d = (float *)sr_workspace(ctx, SR_WS_VECS, in_bytes);
    if (!xyz_d) return -5;
    SR_HIP(hipMemcpyAsync(xyz_d, xyz, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    const size_t row = (size_t)nV * 3;
    int rc = sr_xh_vectors_f32_dev(ctx, xyz_d, nFrames, nAtoms, idxX, idxH, nV, fit_idx, nFit, ref_xyz,
                                   lab ? lab->fm + (size_t)lab->N * row : nullptr, fit ? fit->fm + (size_t)fit->N * row : nullptr, nullptr);
    if (rc) return rc;
    SR_HIP(hipStreamSynchronize(ctx->stream));          // the caller may reuse xyz / the index tables now
    if (lab) { lab->N += nFrames; lab->packed = 0; }
    if (fit) { fit->N += nFrames; fit->packed = 0; }
    return 0;
}

int sr_vectors_download_f32(sr_ctx *ctx, const sr_vectors *h, int64_t f0, int64_t n, float *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h && out, -2, "sr_vectors_download_f32: null pointer");
    SR_REQUIRE(f0 >= 0 && n >= 1 && f0 + n <= h->N, -3, "sr_vectors_download_f32: frames [%lld, %lld) out of range", (long long)f0, (long long)(f0 + n));
    const size_t row = (size_t)h->nV * 3 * sizeof(float);
    if (int rc = wait_ready(ctx, const_cast<sr_vectors *>(h))) return rc;
    SR_HIP(hipMemcpyAsync(out, reinterpret_cast<const char *>(h->fm) + (size_t)f0 * row, (size_t)n * row, hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_vectors_ct_f32(sr_ctx *ctx, sr_vectors *h, int64_t R, int64_t F, const int64_t *chunk_start_host, int mode, double *Ct,
                      double *dCt)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h && Ct && dCt, -2, "sr_vectors_ct_f32: null pointer");
    if (int rc = pack(ctx, h)) return rc;
    const int64_t L = F / 2;
    double *Ct_d = (double *)sr_workspace(ctx, SR_WS_OUT0, (size_t)(L * h->nV) * sizeof(double));
    double *dCt_d = (double *)sr_workspace(ctx, SR_WS_OUT1, (size_t)(L * h->nV) * sizeof(double));
    if (!Ct_d || !dCt_d) return -5;
    if (!chunk_start_host) SR_REQUIRE(R * F <= h->N, -3, "sr_vectors_ct_f32: R*F=%lld exceeds the %lld frames held", (long long)(R * F), (long long)h->N);
    else
        for (int64_t r = 0; r < R; ++r)
            SR_REQUIRE(chunk_start_host[r] >= 0 && chunk_start_host[r] + F <= h->N, -3, "sr_vectors_ct_f32: chunk %lld out of range", (long long)r);
    int rc = sr_ct_palmer_f32_dev(ctx, h->soa, h->Npad, R, F, h->nV, chunk_start_host, mode, nullptr, Ct_d, dCt_d);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(Ct, Ct_d, (size_t)(L * h->nV) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(dCt, dCt_d, (size_t)(L * h->nV) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* raw sums S[v][r][d-1] = sum_j (u_j . u_{j+d})^2 per (vector, chunk, lag) on the HOST, (nV, R, L) compact: what a rank that
 * owns a range of CHUNKS (replicates) contributes when there are fewer vectors than GPUs */
int sr_vectors_ct_sums_f32(sr_ctx *ctx, sr_vectors *h, int64_t R, int64_t F, const int64_t *chunk_start_host, int mode, double *sums)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h && sums, -2, "sr_vectors_ct_sums_f32: null pointer");
    SR_REQUIRE(R >= 1 && F >= 2, -3, "sr_vectors_ct_sums_f32: bad shape R=%lld F=%lld", (long long)R, (long long)F);
    if (int rc = pack(ctx, h)) return rc;
    const int64_t L = F / 2, Lp = sr_ct_psum_stride(F);
    double *psum = (double *)sr_workspace(ctx, SR_WS_PSUM, (size_t)(h->nV * R * Lp) * sizeof(double));
    if (!psum) return -5;
    if (!chunk_start_host) SR_REQUIRE(R * F <= h->N, -3, "sr_vectors_ct_sums_f32: R*F=%lld exceeds the %lld frames held", (long long)(R * F), (long long)h->N);
    else
        for (int64_t r = 0; r < R; ++r)
            SR_REQUIRE(chunk_start_host[r] >= 0 && chunk_start_host[r] + F <= h->N, -3, "sr_vectors_ct_sums_f32: chunk %lld out of range", (long long)r);
    int rc = sr_ct_palmer_sums_f32_dev(ctx, h->soa, h->Npad, R, F, h->nV, chunk_start_host, mode, psum);
    if (rc) return rc;
    SR_HIP(hipMemcpy2DAsync(sums, (size_t)L * sizeof(double), psum + 1, (size_t)Lp * sizeof(double), (size_t)L * sizeof(double),
                            (size_t)(h->nV * R), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* mean / std over the R replicate chunks (calculate-Ct-from-traj.py:226-228) from HOST raw sums (nV, R, L) -- the SAME kernel
 * that finishes a single-process run, so the root of a replicate-sharded run gets the single-process bits */
int sr_ct_finalize_sums_f64(sr_ctx *ctx, const double *sums, int64_t R, int64_t F, int64_t nV, double *Ct, double *dCt)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(sums && Ct && dCt, -2, "sr_ct_finalize_sums_f64: null pointer");
    SR_REQUIRE(R >= 1 && F >= 2 && nV >= 1, -3, "sr_ct_finalize_sums_f64: bad shape");
    const int64_t L = F / 2, Lp = sr_ct_psum_stride(F);
    double *psum = (double *)sr_workspace(ctx, SR_WS_PSUM, (size_t)(nV * R * Lp) * sizeof(double));
    double *Ct_d = (double *)sr_workspace(ctx, SR_WS_OUT0, (size_t)(L * nV) * sizeof(double));
    double *dCt_d = (double *)sr_workspace(ctx, SR_WS_OUT1, (size_t)(L * nV) * sizeof(double));
    if (!psum || !Ct_d || !dCt_d) return -5;
    SR_HIP(hipMemcpy2DAsync(psum + 1, (size_t)Lp * sizeof(double), sums, (size_t)L * sizeof(double), (size_t)L * sizeof(double),
                            (size_t)(nV * R), hipMemcpyHostToDevice, ctx->stream));
    int rc = sr_ct_finalize_f64_dev(ctx, psum, R, F, nV, Ct_d, dCt_d);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(Ct, Ct_d, (size_t)(L * nV) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(dCt, dCt_d, (size_t)(L * nV) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_vectors_hist_f32(sr_ctx *ctx, sr_vectors *h, int64_t N_hist, const double *q, const double *edges_phi, int nphi,
                        const double *edges_cos, int ncos, double *hist, double *vecsum, double *outer, int64_t block_len)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(h && hist, -2, "sr_vectors_hist_f32: null pointer");
    if (int rc = pack(ctx, h)) return rc;
    if (N_hist <= 0) N_hist = h->N;
    SR_REQUIRE(N_hist <= h->N, -3, "sr_vectors_hist_f32: N_hist=%lld exceeds the %lld frames held", (long long)N_hist, (long long)h->N);
    const int64_t nV = h->nV;
    const int nbins = nphi * ncos;
    const int64_t Fb = (block_len > 0 && block_len <= N_hist) ? block_len : N_hist;
    const int64_t nB = N_hist / Fb;
    double *hist_d = (double *)sr_workspace(ctx, SR_WS_OUT0, (size_t)nV * nbins * sizeof(double));
    double *vs_d = (double *)sr_workspace(ctx, SR_WS_OUT1, (size_t)nV * 3 * sizeof(double));
    double *outer_d = (double *)sr_workspace(ctx, SR_WS_IN0, (size_t)nB * nV * 6 * sizeof(double));
    if (!hist_d || !vs_d || !outer_d) return -5;
    int rc = sr_rotate_hist_f32_dev(ctx, h->soa, h->Npad, N_hist, nV, q, edges_phi, nphi, edges_cos, ncos, hist_d, vs_d, outer_d, block_len);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(hist, hist_d, (size_t)nV * nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (vecsum) SR_HIP(hipMemcpyAsync(vecsum, vs_d, (size_t)nV * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (outer) SR_HIP(hipMemcpyAsync(outer, outer_d, (size_t)nB * nV * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
