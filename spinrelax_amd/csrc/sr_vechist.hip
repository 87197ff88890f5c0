// sr_vechist.hip -- kernel 2: quaternion rotation into the diffusion principal-axis frame, Lambert
// cylindrical (phi, cos theta) histogram, mean vector and S2 outer-product sums, in ONE pass over the
// per-vector planes produced by sr_pack_soa_f32_dev.
//
// Reference semantics (float64 throughout, like the reference once a float64 quaternion is applied):
//   rotate_vector_simd          transforms3d_supplement.py:270-296
//   xyz_to_rtp                  general_maths.py:118-158   (r = |v|, phi = atan2(y,x), theta = acos(z/r))
//   cos(theta) + histogramdd    calculate-Ct-from-traj.py:611-626 (numpy binning: searchsorted 'right',
//                               last edge inclusive, out-of-range and NaN dropped)
//   mean vector                 calculate-Ct-from-traj.py:579-583
//   S2 outer products           calculate-Ct-from-traj.py:96-145
//
// Binning cost: float64 atan2 / acos / cos for every sample made the kernel FP64-issue bound (0.75 ms for cfg3,
// 10 % of the HBM roofline).  A float32 estimate now classifies every sample that is clearly inside a bin; only
// samples within a guard band of a bin edge (and NaNs) take the float64 path, so the counts stay bit-identical
// to numpy's (tests/test_gpu_parity.py::test_vechist_*).
//
// Work decomposition: grid = (frame ranges, vectors).  A workgroup owns one vector and a frame range
// that lies inside one S2 block; its histogram lives in LDS as 32-bit counters (nphi*ncos*4 B = 10 KB
// for 72x36) and is flushed with integer atomics (deterministic); the 9 float64 sums go to a
// per-workgroup slot and are combined in fixed order by k_vechist_finalize (bitwise reproducible).
#include "sr_internal.h"

namespace {

constexpr int kMaxEdges = 1024;
constexpr int kInlineEdges = 128;      // 72 + 36 + 2 edges of the default histogram fit the kernel argument block

struct VhArgs {
    const float *soa;
    int64_t Npad, N;
    int64_t Fb;          // S2 block length (== N when no block averaging)
    int64_t sub;         // frames per workgroup inside a block
    int m;               // workgroups per S2 block
    int nB;              // number of S2 blocks
    int nranges;         // nB*m (+1 when a tail exists)
    int nphi, ncos;
    int rotate;
    double qw, qx, qy, qz;
    const double *edges; // device: nphi+1 then ncos+1; null = the edges travel in edges_inline (no upload per call)
    double edges_inline[kInlineEdges];
    unsigned int *hist_u32;   // (nV, nphi*ncos)
    double *partials;         // (nV, nranges, 9)
};

// numpy.searchsorted(edges, x, 'right') - 1 with histogramdd's last-edge rule; -1 = not counted
__device__ __forceinline__ int np_bin(const double *e, int nb, double x)
{
    if (!(x >= e[0]) || !(x <= e[nb])) return -1;           // also rejects NaN
    int k = (int)((x - e[0]) / (e[nb] - e[0]) * (double)nb);
    k = k < 0 ? 0 : (k > nb - 1 ? nb - 1 : k);
    while (k > 0 && x < e[k]) --k;
    while (k < nb - 1 && x >= e[k + 1]) ++k;
    return k;
}

__device__ __forceinline__ void rotate_q(double qw, double qx, double qy, double qz, double vx, double vy, double vz,
                                         double &ox, double &oy, double &oz)
{
#pragma clang fp contract(off)
    // a = qv x v + qw v ; b = qv x a ; out = b + b + v        (numpy.cross component order)
    const double ax = (qy * vz - qz * vy) + qw * vx;
    const double ay = (qz * vx - qx * vz) + qw * vy;
    const double az = (qx * vy - qy * vx) + qw * vz;
    const double bx = qy * az - qz * ay;
    const double by = qz * ax - qx * az;
    const double bz = qx * ay - qy * ax;
    ox = (bx + bx) + vx;
    oy = (by + by) + vy;
    oz = (bz + bz) + vz;
}

// Accumulator state of one thread
struct VhAcc {
    double sx, sy, sz, oxx, oyy, ozz, oxy, oxz, oyz;
};

// exact (reference-order, float64) bin coordinates of a rotated vector: xyz_to_rtp + cos(theta)
__device__ __noinline__ void exact_phi_cos(double x, double y, double z, double &phi, double &c)
{
#pragma clang fp contract(off)
    const double r = sqrt((x * x + y * y) + z * z);
    phi = atan2(y, x);
    c = cos(acos(z / r));
}

// One sample: rotate (float64, reference operation order), accumulate the sums, histogram it.
// Binning: a float32 estimate of (phi, cos theta) decides the bin when it lies at least kEdgeGuard bin widths
// away from both edges of that bin -- the estimate is good to < 1e-6 rad / 5e-7, the guard band is 1.7e-5 rad /
// 1.1e-5, so the decision is the one numpy's searchsorted makes on the float64 values; otherwise (4e-4 of the
// samples, and every NaN / out-of-range value) the float64 atan2 / acos / cos path decides exactly like before.
constexpr float kEdgeGuard = 2e-4f;

__device__ __forceinline__ void vh_sample(const VhArgs &a, float xf, float yf, float zf, bool in_block,
                                          const double *ephi, const double *ecos, unsigned int *h, VhAcc &s,
                                          float phi_scale, float cos_scale)
{
    double x = (double)xf, y = (double)yf, z = (double)zf;
    if (a.rotate) {
        double rx, ry, rz;
        rotate_q(a.qw, a.qx, a.qy, a.qz, x, y, z, rx, ry, rz);
        x = rx; y = ry; z = rz;
    }
    s.sx += x; s.sy += y; s.sz += z;
    if (in_block) {
        s.oxx += x * x; s.oyy += y * y; s.ozz += z * z;
        s.oxy += x * y; s.oxz += x * z; s.oyz += y * z;
    }
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    const float tp = (atan2f(fy, fx) + 3.14159265358979f) * phi_scale;          // position in phi-bin units
    const float tc = (fz * rsqrtf((fx * fx + fy * fy) + fz * fz) + 1.0f) * cos_scale;
    const float kpf = floorf(tp), kcf = floorf(tc);
    int kp = (int)kpf, kc = (int)kcf;
    const bool sure = (tp - kpf > kEdgeGuard) && (tp - kpf < 1.0f - kEdgeGuard) && (tc - kcf > kEdgeGuard) &&
                      (tc - kcf < 1.0f - kEdgeGuard) && kp >= 0 && kp < a.nphi && kc >= 0 && kc < a.ncos;
    if (!sure) {
        double phi, c;
        exact_phi_cos(x, y, z, phi, c);
        kp = np_bin(ephi, a.nphi, phi);
        kc = np_bin(ecos, a.ncos, c);
    }
    if (kp >= 0 && kc >= 0) atomicAdd(&h[kp * a.ncos + kc], 1u);
}

__global__ __launch_bounds__(256) void k_vechist(VhArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *edges = reinterpret_cast<double *>(smem);                 // nphi+1 + ncos+1
    const int ne = a.nphi + 1 + a.ncos + 1;
    double *red = edges + ne;                                         // 4 waves x 9
    unsigned int *h = reinterpret_cast<unsigned int *>(red + 36);     // nphi*ncos
    const int nbins = a.nphi * a.ncos;
    const int tid = threadIdx.x;
    const int rid = blockIdx.x;
    const int64_t v = blockIdx.y;

    for (int i = tid; i < ne; i += 256) edges[i] = a.edges ? a.edges[i] : a.edges_inline[i];
    for (int i = tid; i < nbins; i += 256) h[i] = 0u;
    __syncthreads();
    const double *ephi = edges, *ecos = edges + a.nphi + 1;
    // the float32 estimate assumes the uniform numpy.linspace edges of calculate-Ct-from-traj.py:618
    const float phi_scale = (float)((double)a.nphi / (ephi[a.nphi] - ephi[0]));
    const float cos_scale = (float)((double)a.ncos / (ecos[a.ncos] - ecos[0]));

    int64_t start, end;
    bool in_block;
    if (rid < a.nB * a.m) {
        const int b = rid / a.m, i = rid - b * a.m;
        start = (int64_t)b * a.Fb + (int64_t)i * a.sub;
        end = min(start + a.sub, (int64_t)(b + 1) * a.Fb);
        in_block = true;
    } else {
        start = (int64_t)a.nB * a.Fb;
        end = a.N;
        in_block = false;
    }
    const float *px = a.soa + (v * 3) * a.Npad;
    const float *py = px + a.Npad;
    const float *pz = py + a.Npad;

    VhAcc s = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t n0 = start;
    if ((start & 3) == 0) {
        // 16-byte loads: 4 consecutive frames per thread
        const int64_t nvec = (end - start) >> 2;
        for (int64_t q = tid; q < nvec; q += 256) {
            const int64_t n = start + (q << 2);
            const float4 X = *reinterpret_cast<const float4 *>(px + n);
            const float4 Y = *reinterpret_cast<const float4 *>(py + n);
            const float4 Z = *reinterpret_cast<const float4 *>(pz + n);
            vh_sample(a, X.x, Y.x, Z.x, in_block, ephi, ecos, h, s, phi_scale, cos_scale);
            vh_sample(a, X.y, Y.y, Z.y, in_block, ephi, ecos, h, s, phi_scale, cos_scale);
            vh_sample(a, X.z, Y.z, Z.z, in_block, ephi, ecos, h, s, phi_scale, cos_scale);
            vh_sample(a, X.w, Y.w, Z.w, in_block, ephi, ecos, h, s, phi_scale, cos_scale);
        }
        n0 = start + (nvec << 2);
    }
    for (int64_t n = n0 + tid; n < end; n += 256)
        vh_sample(a, px[n], py[n], pz[n], in_block, ephi, ecos, h, s, phi_scale, cos_scale);

    // block reduction of the 9 sums (fixed order: lanes by DPP butterfly, then waves 0..3)
    double vals[9] = {s.sx, s.sy, s.sz, s.oxx, s.oyy, s.ozz, s.oxy, s.oxz, s.oyz};
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double t = sr_wave_sum_f64(vals[k]);
        if (lane == 0) red[wave * 9 + k] = t;
    }
    __syncthreads();
    if (tid < 9) {
        const double t = ((red[tid] + red[9 + tid]) + red[18 + tid]) + red[27 + tid];
        a.partials[(v * a.nranges + rid) * 9 + tid] = t;
    }
    unsigned int *gh = a.hist_u32 + v * nbins;
    for (int i = tid; i < nbins; i += 256) {
        const unsigned int c = h[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

__global__ __launch_bounds__(256) void k_vechist_finalize(const unsigned int *__restrict__ hist_u32,
                                                          const double *__restrict__ partials, int64_t nV, int nbins,
                                                          int nranges, int nB, int m, double *__restrict__ hist,
                                                          double *__restrict__ vecsum, double *__restrict__ outer)
{
    const int64_t v = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < nbins; i += 256) hist[v * nbins + i] = (double)hist_u32[v * nbins + i];
    const double *p = partials + v * nranges * 9;
    if (vecsum && tid < 3) {
        double s = 0.0;
        for (int r = 0; r < nranges; ++r) s += p[r * 9 + tid];
        vecsum[v * 3 + tid] = s;
    }
    if (outer) {
        for (int i = tid; i < nB * 6; i += 256) {
            const int b = i / 6, k = i - b * 6;
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += p[(b * m + j) * 9 + 3 + k];
            outer[((int64_t)b * nV + v) * 6 + k] = s;
        }
    }
}

// rotated vectors themselves: (N, Vtot, 3) float32 slice -> (N, nV, 3) float64
// quat != null: one unit quaternion per frame, (N, 4) as w x y z -- rotate_vector_simd with q of shape (N, 1, 4)
__global__ __launch_bounds__(256) void k_rotate_vectors(const float *__restrict__ vecs, int64_t N, int64_t Vtot,
                                                        int64_t v0, int64_t nV, int rotate, double qw, double qx,
                                                        double qy, double qz, const double *__restrict__ quat,
                                                        double *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * nV) return;
    const int64_t n = idx / nV, v = idx - n * nV;
    const float *s = vecs + (n * Vtot + v0 + v) * 3;
    double x = (double)s[0], y = (double)s[1], z = (double)s[2];
    if (quat) {
        qw = quat[n * 4 + 0]; qx = quat[n * 4 + 1]; qy = quat[n * 4 + 2]; qz = quat[n * 4 + 3];
    }
    if (rotate) {
        double rx, ry, rz;
        rotate_q(qw, qx, qy, qz, x, y, z, rx, ry, rz);
        x = rx; y = ry; z = rz;
    }
    out[idx * 3 + 0] = x;
    out[idx * 3 + 1] = y;
    out[idx * 3 + 2] = z;
}

// normalise q like vecnorm_NDarray (transforms3d_supplement.py:40-52): q / |q|, 0/0 -> 0
void normalise_q(const double *q, double *o)
{
    const double n = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) {
        double t = q[i] / n;
        if (t != t) t = 0.0;
        o[i] = t;
    }
}

}  // namespace

extern "C" {

int sr_rotate_hist_f32_dev(sr_ctx *ctx, const float *soa, int64_t Npad, int64_t N, int64_t nV, const double *q_host,
                           const double *edges_phi_host, int nphi, const double *edges_cos_host, int ncos,
                           double *hist, double *vecsum, double *outer, int64_t block_len)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(soa && hist && edges_phi_host && edges_cos_host, -2, "sr_rotate_hist_f32_dev: null pointer");
    SR_REQUIRE(N >= 1 && nV >= 1 && N <= Npad, -3, "sr_rotate_hist_f32_dev: bad shape N=%lld nV=%lld Npad=%lld",
               (long long)N, (long long)nV, (long long)Npad);
    SR_REQUIRE(nphi >= 1 && ncos >= 1 && nphi + ncos + 2 <= kMaxEdges && (int64_t)nphi * ncos <= 32768, -3,
               "sr_rotate_hist_f32_dev: unsupported histogram size %d x %d", nphi, ncos);
    SR_REQUIRE(nV <= 65535, -3, "sr_rotate_hist_f32_dev: at most 65535 vectors per call");
    const int nbins = nphi * ncos;
    VhArgs a;
    a.soa = soa; a.Npad = Npad; a.N = N;
    a.Fb = (block_len > 0 && block_len <= N) ? block_len : N;
    a.nB = (int)(N / a.Fb);
    // enough workgroups to fill the chip: aim at >= 4096 in total, each at least 1024 frames
    int64_t want = (4096 + nV - 1) / nV;
    int64_t per_block = (want + a.nB - 1) / a.nB;
    if (per_block < 1) per_block = 1;
    int64_t maxm = (a.Fb + 1023) / 1024;
    if (per_block > maxm) per_block = maxm;
    a.m = (int)per_block;
    a.sub = sr_round_up((a.Fb + a.m - 1) / a.m, 4);
    a.m = (int)((a.Fb + a.sub - 1) / a.sub);
    const bool tail = (int64_t)a.nB * a.Fb < N;
    a.nranges = a.nB * a.m + (tail ? 1 : 0);
    a.nphi = nphi; a.ncos = ncos;
    a.rotate = q_host ? 1 : 0;
    a.qw = 1; a.qx = a.qy = a.qz = 0;
    if (q_host) {
        double qn[4];
        normalise_q(q_host, qn);
        a.qw = qn[0]; a.qx = qn[1]; a.qy = qn[2]; a.qz = qn[3];
    }
    const int ne = nphi + 1 + ncos + 1;
    const size_t misc_bytes = (size_t)ne * sizeof(double);
    double *edges_d = nullptr;
    if (ne > kInlineEdges) {
        edges_d = (double *)sr_workspace(ctx, SR_WS_IN3, misc_bytes);
        if (!edges_d) return -5;
    }
    unsigned int *h32 = (unsigned int *)sr_workspace(ctx, SR_WS_OUT2, (size_t)nV * nbins * sizeof(unsigned int));
    double *partials = (double *)sr_workspace(ctx, SR_WS_OUT3, (size_t)nV * a.nranges * 9 * sizeof(double));
    if (!h32 || !partials) return -5;
    if (edges_d) {
        SR_HIP(hipMemcpyAsync(edges_d, edges_phi_host, (size_t)(nphi + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SR_HIP(hipMemcpyAsync(edges_d + nphi + 1, edges_cos_host, (size_t)(ncos + 1) * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
    } else {
        for (int i = 0; i <= nphi; ++i) a.edges_inline[i] = edges_phi_host[i];
        for (int i = 0; i <= ncos; ++i) a.edges_inline[nphi + 1 + i] = edges_cos_host[i];
    }
    SR_HIP(hipMemsetAsync(h32, 0, (size_t)nV * nbins * sizeof(unsigned int), ctx->stream));
    a.edges = edges_d; a.hist_u32 = h32; a.partials = partials;
    const size_t lds = (size_t)ne * sizeof(double) + 36 * sizeof(double) + (size_t)nbins * sizeof(unsigned int);
    if (int rc = sr_grant_lds(ctx, SR_K_VECHIST, reinterpret_cast<const void *>(&k_vechist), lds)) return rc;
    hipLaunchKernelGGL(k_vechist, dim3((unsigned)a.nranges, (unsigned)nV), dim3(256), lds, ctx->stream, a);
    SR_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_vechist_finalize, dim3((unsigned)nV), dim3(256), 0, ctx->stream, h32, partials, nV, nbins,
                       a.nranges, a.nB, a.m, hist, vecsum, outer);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_rotate_hist_f32(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV, const double *q,
                       const double *edges_phi, int nphi, const double *edges_cos, int ncos, double *hist,
                       double *vecsum, double *outer, int64_t block_len)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && hist, -2, "sr_rotate_hist_f32: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3, "sr_rotate_hist_f32: bad shape");
    const int64_t Npad = sr_round_up(N, 64);
    const size_t in_bytes = (size_t)N * Vtot * 3 * sizeof(float);
    const int nbins = nphi * ncos;
    const int64_t Fb = (block_len > 0 && block_len <= N) ? block_len : N;
    const int64_t nB = N / Fb;
    float *dvecs = (float *)sr_workspace(ctx, SR_WS_VECS, in_bytes);
    float *soa = (float *)sr_workspace(ctx, SR_WS_SOA, (size_t)nV * 3 * Npad * sizeof(float));
    double *hist_d = (double *)sr_workspace(ctx, SR_WS_OUT0, (size_t)nV * nbins * sizeof(double));
    double *vs_d = (double *)sr_workspace(ctx, SR_WS_OUT1, (size_t)nV * 3 * sizeof(double));
    double *outer_d = (double *)sr_workspace(ctx, SR_WS_IN0, (size_t)nB * nV * 6 * sizeof(double));
    if (!dvecs || !soa || !hist_d || !vs_d || !outer_d) return -5;
    SR_HIP(hipMemcpyAsync(dvecs, vecs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = sr_pack_soa_f32_dev(ctx, dvecs, N, Vtot, v0, nV, soa, Npad);
    if (rc) return rc;
    rc = sr_rotate_hist_f32_dev(ctx, soa, Npad, N, nV, q, edges_phi, nphi, edges_cos, ncos, hist_d, vs_d, outer_d, block_len);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(hist, hist_d, (size_t)nV * nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (vecsum) SR_HIP(hipMemcpyAsync(vecsum, vs_d, (size_t)nV * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (outer) SR_HIP(hipMemcpyAsync(outer, outer_d, (size_t)nB * nV * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_rotate_vectors_f32(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                          const double *q, double *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && out, -2, "sr_rotate_vectors_f32: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3, "sr_rotate_vectors_f32: bad shape");
    const size_t in_bytes = (size_t)N * Vtot * 3 * sizeof(float);
    const size_t out_bytes = (size_t)N * nV * 3 * sizeof(double);
    float *dvecs = (float *)sr_workspace(ctx, SR_WS_VECS, in_bytes);
    double *dout = (double *)sr_workspace(ctx, SR_WS_OUT0, out_bytes);
    if (!dvecs || !dout) return -5;
    double qn[4] = {1, 0, 0, 0};
    if (q) normalise_q(q, qn);
    SR_HIP(hipMemcpyAsync(dvecs, vecs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    const int64_t tot = N * nV;
    hipLaunchKernelGGL(k_rotate_vectors, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dvecs, N, Vtot, v0,
                       nV, q ? 1 : 0, qn[0], qn[1], qn[2], qn[3], (const double *)nullptr, dout);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_rotate_vectors_perframe_f32(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                                   const double *quat, double *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && out && quat, -2, "sr_rotate_vectors_perframe_f32: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3, "sr_rotate_vectors_perframe_f32: bad shape");
    const size_t in_bytes = (size_t)N * Vtot * 3 * sizeof(float);
    const size_t out_bytes = (size_t)N * nV * 3 * sizeof(double);
    float *dvecs = (float *)sr_workspace(ctx, SR_WS_VECS, in_bytes);
    double *dout = (double *)sr_workspace(ctx, SR_WS_OUT0, out_bytes);
    double *dq = (double *)sr_workspace(ctx, SR_WS_IN0, (size_t)N * 4 * sizeof(double));
    if (!dvecs || !dout || !dq) return -5;
    SR_HIP(hipMemcpyAsync(dvecs, vecs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(dq, quat, (size_t)N * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int64_t tot = N * nV;
    hipLaunchKernelGGL(k_rotate_vectors, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dvecs, N, Vtot, v0,
                       nV, 1, 1.0, 0.0, 0.0, 0.0, (const double *)dq, dout);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
