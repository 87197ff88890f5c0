// sr_dq.hip -- global rotational diffusion from the orientation trajectory:
// lag correlations of the difference quaternion  dq_i(delta) = q_i^-1 * q_{i+delta}.
//
// Reference semantics (calculate-dq-distribution.py):
//   obtain_self_dq                 :102-109  quat_reduce_simd(quat_mult_simd(quat_invert(q[:-d]), q[d:]))
//                                            (transforms3d_supplement.py:163-186, 219-227)
//   average_LegendreP1quat         :111-112  isotropic decay  <1 - 2 |v_q|^2>
//   average_anisotropic_tensor     :118-126  mean of v_q (x) v_q  (3 x 3, symmetric)
//   *_chunk                        :128-144  the same per sub-chunk c of nblock = ceil(ndat / nchunk) samples
//   main loop over delta           :554-609
// Everything the reference derives per lag is a function of the six second moments of v_q = (x, y, z) of dq, so the
// device produces exactly those: for every lag k and chunk c the sums of xx, yy, zz, xy, xz, yz over the chunk's samples
// plus the sample count; eigen-decomposition, frame rotation (R M R^T) and the Powell fits stay on the host
// (spinrelax_amd/dq_distribution.py).  quat_reduce's sign flip changes v_q -> -v_q and leaves v_q (x) v_q unchanged.
//
// Arithmetic: float32 quaternions in (PLUMED prints single precision, plumedcolvario.py:14-15), every product and sum in
// float64 -- the parity definition of SURVEY.md section 8(c) (the reference functions evaluated in float64).
//
// Shape of the work: the same shifted-pair reduction as kernel 1 on an (N, 4) array.  The whole array (16 B per frame)
// lives in L2 / Infinity Cache after the first lag, so HBM sees it once; the kernel is bound by float64 VALU issue
// (~45 float64 instructions per (sample, lag)).  Grid = (sub-range, chunk, lag): a workgroup owns one (lag, chunk)
// and a contiguous sample range, reduces its six sums by DPP wave sums + a fixed-order combine (bitwise reproducible),
// and a second tiny kernel adds the sub-ranges in order.
#include "sr_internal.h"

namespace {

struct DqArgs {
    const void *q;          // (N) quaternions w, x, y, z: float4, or 4 doubles each (the F64 kernel)
    int64_t N;
    const int *lags;        // (nlags) device
    int nlags, nchunk, nsub;
    double *partials;       // (nlags, nchunk, nsub, 6)
};

// F64: float64 quaternions in (the gmx rotmat .xvg route of the reference keeps float64: rotmatrix_to_quaternion,
// calculate-dq-distribution.py:482-497); same arithmetic, the conversions drop out
template <bool F64>
__global__ __launch_bounds__(256) void k_dq_moments(DqArgs a)
{
#pragma clang fp contract(off)
    __shared__ double red[4 * 6];
    const int tid = threadIdx.x;
    const int s = blockIdx.x, c = blockIdx.y, k = blockIdx.z;
    const int64_t d = a.lags[k];
    const int64_t ndat = a.N - d;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (ndat > 0) {
        const int64_t nblock = (ndat + a.nchunk - 1) / a.nchunk;           // average_*_chunk :129, :138
        const int64_t jmin = nblock * c;
        int64_t jmax = nblock * (c + 1);
        if (jmax > ndat) jmax = ndat;
        if (jmin < jmax) {
            const int64_t per = (jmax - jmin + a.nsub - 1) / a.nsub;
            const int64_t lo = jmin + per * s;
            int64_t hi = lo + per;
            if (hi > jmax) hi = jmax;
            for (int64_t i = lo + tid; i < hi; i += 256) {
                double w1, x1, y1, z1, w2, x2, y2, z2;
                // q1 = quat_invert(q_i) = (w, -x, -y, -z); out = quat_mult_simd(q1, q2), vector part only:
                //   w1 v2 + w2 v1 + v1 x v2      (transforms3d_supplement.py:182)
                if (F64) {
                    const double2 *qd = reinterpret_cast<const double2 *>(a.q);
                    const double2 A0 = qd[2 * i], A1 = qd[2 * i + 1], B0 = qd[2 * (i + d)], B1 = qd[2 * (i + d) + 1];
                    w1 = A0.x; x1 = -A0.y; y1 = -A1.x; z1 = -A1.y;
                    w2 = B0.x; x2 = B0.y; y2 = B1.x; z2 = B1.y;
                } else {
                    const float4 *qf = reinterpret_cast<const float4 *>(a.q);
                    const float4 A = qf[i], B = qf[i + d];
                    w1 = (double)A.x; x1 = -(double)A.y; y1 = -(double)A.z; z1 = -(double)A.w;
                    w2 = (double)B.x; x2 = (double)B.y; y2 = (double)B.z; z2 = (double)B.w;
                }
                const double vx = (w1 * x2 + w2 * x1) + (y1 * z2 - z1 * y2);
                const double vy = (w1 * y2 + w2 * y1) + (z1 * x2 - x1 * z2);
                const double vz = (w1 * z2 + w2 * z1) + (x1 * y2 - y1 * x2);
                acc[0] = fma(vx, vx, acc[0]);
                acc[1] = fma(vy, vy, acc[1]);
                acc[2] = fma(vz, vz, acc[2]);
                acc[3] = fma(vx, vy, acc[3]);
                acc[4] = fma(vx, vz, acc[4]);
                acc[5] = fma(vy, vz, acc[5]);
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const double t = sr_wave_sum_f64(acc[m]);
        if (lane == 0) red[wave * 6 + m] = t;
    }
    __syncthreads();
    if (tid < 6) {
        const double t = ((red[tid] + red[6 + tid]) + red[12 + tid]) + red[18 + tid];
        a.partials[(((int64_t)k * a.nchunk + c) * a.nsub + s) * 6 + tid] = t;
    }
}

__global__ __launch_bounds__(64) void k_dq_finalize(const double *__restrict__ partials, const int *__restrict__ lags,
                                                    int64_t N, int nlags, int nchunk, int nsub, double *__restrict__ out)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= nlags * nchunk * 7) return;
    const int m = idx % 7, kc = idx / 7;
    const int c = kc % nchunk, k = kc / nchunk;
    if (m < 6) {
        double s = 0.0;
        for (int j = 0; j < nsub; ++j) s += partials[((int64_t)kc * nsub + j) * 6 + m];
        out[(int64_t)kc * 7 + m] = s;
    } else {
        const int64_t ndat = N - lags[k];
        double cnt = 0.0;
        if (ndat > 0) {
            const int64_t nblock = (ndat + nchunk - 1) / nchunk;
            const int64_t jmin = nblock * c;
            int64_t jmax = nblock * (c + 1);
            if (jmax > ndat) jmax = ndat;
            if (jmax > jmin) cnt = (double)(jmax - jmin);
        }
        out[(int64_t)kc * 7 + 6] = cnt;
    }
}

}  // namespace

extern "C" {

static int dq_moments_dev(sr_ctx *ctx, const void *q, bool f64, int64_t N, const int32_t *lags_host, int nlags, int nchunk, double *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(q && lags_host && out, -2, "sr_dq_moments_dev: null pointer");
    SR_REQUIRE(N >= 2 && nlags >= 1 && nchunk >= 1 && nchunk <= 65535 && nlags <= 65535, -3,
               "sr_dq_moments_dev: bad sizes N=%lld nlags=%d nchunk=%d", (long long)N, nlags, nchunk);
    int64_t dmin = N;
    for (int k = 0; k < nlags; ++k) {
        SR_REQUIRE(lags_host[k] >= 1 && lags_host[k] < N, -3, "sr_dq_moments_dev: lag %d = %d out of range (1..%lld)", k,
                   lags_host[k], (long long)(N - 1));
        if (lags_host[k] < dmin) dmin = lags_host[k];
    }
    // sub-ranges: enough workgroups to fill the chip (>= ~2048), each at least 2048 samples
    const int64_t chunk_len = (N - dmin + nchunk - 1) / nchunk;
    int64_t nsub = (2048 + (int64_t)nlags * nchunk - 1) / ((int64_t)nlags * nchunk);
    const int64_t maxsub = (chunk_len + 2047) / 2048;
    if (nsub > maxsub) nsub = maxsub;
    if (nsub < 1) nsub = 1;
    int *lags_d = (int *)sr_workspace(ctx, SR_WS_IN3, (size_t)nlags * sizeof(int));
    double *partials = (double *)sr_workspace(ctx, SR_WS_OUT3, (size_t)nlags * nchunk * nsub * 6 * sizeof(double));
    if (!lags_d || !partials) return -5;
    // the lag table is tiny: its copy is complete when this function returns (the caller may free it, pinned or not)
    SR_HIP(hipMemcpyAsync(lags_d, lags_host, (size_t)nlags * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    DqArgs a;
    a.q = q; a.N = N; a.lags = lags_d; a.nlags = nlags; a.nchunk = nchunk;
    a.nsub = (int)nsub; a.partials = partials;
    if (f64) hipLaunchKernelGGL(k_dq_moments<true>, dim3((unsigned)nsub, (unsigned)nchunk, (unsigned)nlags), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k_dq_moments<false>, dim3((unsigned)nsub, (unsigned)nchunk, (unsigned)nlags), dim3(256), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    const int tot = nlags * nchunk * 7;
    hipLaunchKernelGGL(k_dq_finalize, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, ctx->stream, partials, lags_d, N, nlags,
                       nchunk, (int)nsub, out);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_dq_moments_f32_dev(sr_ctx *ctx, const float *q, int64_t N, const int32_t *lags_host, int nlags, int nchunk, double *out)
{
    return dq_moments_dev(ctx, q, false, N, lags_host, nlags, nchunk, out);
}

int sr_dq_moments_f64_dev(sr_ctx *ctx, const double *q, int64_t N, const int32_t *lags_host, int nlags, int nchunk, double *out)
{
    return dq_moments_dev(ctx, q, true, N, lags_host, nlags, nchunk, out);
}

int sr_dq_moments_f64(sr_ctx *ctx, const double *q, int64_t N, const int32_t *lags, int nlags, int nchunk, double *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(q && lags && out, -2, "sr_dq_moments_f64: null pointer");
    SR_REQUIRE(N >= 2 && nlags >= 1 && nchunk >= 1, -3, "sr_dq_moments_f64: bad sizes");
    const size_t nout = (size_t)nlags * nchunk * 7;
    double *q_d = (double *)sr_workspace(ctx, SR_WS_VECS, (size_t)N * 4 * sizeof(double));
    double *out_d = (double *)sr_workspace(ctx, SR_WS_OUT0, nout * sizeof(double));
    if (!q_d || !out_d) return -5;
    SR_HIP(hipMemcpyAsync(q_d, q, (size_t)N * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = sr_dq_moments_f64_dev(ctx, q_d, N, lags, nlags, nchunk, out_d);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(out, out_d, nout * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_dq_moments_f32(sr_ctx *ctx, const float *q, int64_t N, const int32_t *lags, int nlags, int nchunk, double *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(q && lags && out, -2, "sr_dq_moments_f32: null pointer");
    SR_REQUIRE(N >= 2 && nlags >= 1 && nchunk >= 1, -3, "sr_dq_moments_f32: bad sizes");
    const size_t nout = (size_t)nlags * nchunk * 7;
    float *q_d = (float *)sr_workspace(ctx, SR_WS_VECS, (size_t)N * 4 * sizeof(float));
    double *out_d = (double *)sr_workspace(ctx, SR_WS_OUT0, nout * sizeof(double));
    if (!q_d || !out_d) return -5;
    SR_HIP(hipMemcpyAsync(q_d, q, (size_t)N * 4 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = sr_dq_moments_f32_dev(ctx, q_d, N, lags, nlags, nchunk, out_d);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(out, out_d, nout * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
