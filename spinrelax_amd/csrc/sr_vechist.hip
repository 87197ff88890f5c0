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
// The kernel is HBM-bound by algorithm (one pass over 12 B per sample); what it must NOT do is spend float64 issue slots
// per sample.  Round 1 rotated every sample in float64, ran nine float64 sums on the rotated values and decided the bin
// from them (0.25 ms for cfg3 = 30 % of HBM).  Now:
//   * sums: the rotation is linear, so sum R v = R sum v and sum (R v)(R v)^T = R (sum v v^T) R^T.  The kernel
//     accumulates the UNROTATED float32 samples in float64 (products of two float32 are exact in float64) and
//     k_vechist_finalize rotates the 3 + 6 sums once per (vector, block): 12 float64 instructions per sample instead of ~60;
//   * bins: a float32 rotation + float32 (phi, cos theta) estimate classifies every sample that is clearly inside a
//     bin; samples within a guard band of a bin edge, near the poles (where the float32 rotation error is amplified
//     in phi), NaNs and out-of-range values take the exact path -- float64 rotation in the reference's operation
//     order, atan2 / acos / cos, numpy's searchsorted rule -- so the counts stay bit-identical to numpy's
//     (tests/test_gpu_parity.py::test_vechist_*).
//
// Work decomposition: grid = (frame ranges, vectors).  A workgroup owns one vector and a frame range
// that lies inside one S2 block; its histogram lives in LDS as 32-bit counters (nphi*ncos*4 B = 10 KB
// for 72x36) and is flushed with integer atomics (deterministic); the 9 float64 sums go to a
// per-workgroup slot and are combined in fixed order by k_vechist_finalize (bitwise reproducible).
#include "sr_internal.h"
#include <cstdlib>

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
    float rm[9];              // rotation matrix of the same quaternion in float32, row-major (bin estimate only)
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

// One sample.  Sums: the unrotated float32 components in float64 (rotated once in k_vechist_finalize).
// Binning: a float32 rotation and a float32 estimate of (phi, cos theta) decide the bin when the estimate lies at least
// kEdgeGuard bin widths away from both edges of that bin and the vector is not within ~2 degrees of a pole.  Error
// budget of the estimate: the float32 rotation is good to ~2.4e-7 absolute per component of a unit vector, i.e. 5.4e-6 rad
// in phi once x^2 + y^2 > 4e-3 r^2 (farther than 3.6 degrees from a pole), and 6e-7 in cos theta; atan2f, rsqrtf and the
// float32 scaling add < 1.5e-6 rad / 3e-7.  The guard band is the larger of 2e-4 bin widths and 1.7e-5 rad / 1.1e-5 (the
// two coincide for the default 72 x 36 grid), so the decision is the one numpy's searchsorted makes on the float64
// values; every other sample (~3e-3 of them, and every NaN / out-of-range value) takes the exact float64 path.
constexpr float kEdgeGuard = 2e-4f;
constexpr float kPhiGuardRad = 1.7e-5f, kCosGuard = 1.1e-5f;

constexpr int kMaxRange = 8192;      // frames per range: bounds the LDS mask (1 KB) and list (16 KB) of undecided samples

// exact (reference-order, float64) classification of one sample.  PK: two 16-bit counters per word (the fused pack +
// histogram kernel keeps eight vectors' histograms in LDS; a workgroup never counts more than 65 535 samples)
template <bool PK = false>
__device__ __forceinline__ void vh_exact(const VhArgs &a, float xf, float yf, float zf, const double *ephi, const double *ecos,
                                         unsigned int *h)
{
    const double x = (double)xf, y = (double)yf, z = (double)zf;
    double rx = x, ry = y, rz = z, phi, c;
    if (a.rotate) rotate_q(a.qw, a.qx, a.qy, a.qz, x, y, z, rx, ry, rz);
    exact_phi_cos(rx, ry, rz, phi, c);
    const int kp = np_bin(ephi, a.nphi, phi), kc = np_bin(ecos, a.ncos, c);
    if (kp >= 0 && kc >= 0) {
        const int bin = kp * a.ncos + kc;
        if (PK) atomicAdd(&h[bin >> 1], 1u << ((bin & 1) << 4));
        else atomicAdd(&h[bin], 1u);
    }
}

// atan2(y, x) in float32 without libm: octant reduction, one v_rcp_f32, a degree-7 polynomial in (min/max)^2 fitted to
// atan(a)/a on [0, 1] (max error 1.5e-7 rad in float32 Horner form), quadrant fix-ups.  Good to < 6e-7 rad; signed
// zeros and x = y = 0 do not matter here (such samples sit on a bin edge or are NaN and take the exact path anyway).
__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float u = t * t;
    float p = -0.00455979211f;
    p = fmaf(p, u, 0.0237805191f);
    p = fmaf(p, u, -0.0588297546f);
    p = fmaf(p, u, 0.0986886546f);
    p = fmaf(p, u, -0.140032902f);
    p = fmaf(p, u, 0.199669614f);
    p = fmaf(p, u, -0.333318114f);
    p = fmaf(p, u, 0.999999881f);
    float r = t * p;
    r = ay > ax ? 1.57079632679f - r : r;
    r = x < 0.f ? 3.14159265359f - r : r;
    return y < 0.f ? -r : r;
}

// One sample, branch-free.  Sums: the unrotated float32 components in float64 (rotated once in k_vechist_finalize).
// Bin: decided from the float32 estimate when it is safe (see above); ONE LDS atomic either way -- a decided sample adds 1
// to its bin, an undecided one sets its bit (idx = its position in the range) in the LDS mask; the marked samples are
// collected and classified exactly, one per thread, when the range has been streamed.  (Taken inline, the ~600-instruction
// float64 path of one lane in ~300 would stall the other 63 lanes of its wave in one of six iterations.)
template <bool ROT, bool INB, bool PK = false>
__device__ __forceinline__ void vh_sample(const VhArgs &a, float xf, float yf, float zf, unsigned int idx, unsigned int *h,
                                          unsigned int *mask, VhAcc &s, float phi_scale, float cos_scale, float hp, float hc)
{
    const double x = (double)xf, y = (double)yf, z = (double)zf;
    s.sx += x; s.sy += y; s.sz += z;
    if (INB) {
        s.oxx = fma(x, x, s.oxx); s.oyy = fma(y, y, s.oyy); s.ozz = fma(z, z, s.ozz);
        s.oxy = fma(x, y, s.oxy); s.oxz = fma(x, z, s.oxz); s.oyz = fma(y, z, s.oyz);
    }
    float fx = xf, fy = yf, fz = zf;
    if (ROT) {                                     // float32 rotation matrix of the unit quaternion (estimate only)
        fx = fmaf(a.rm[0], xf, fmaf(a.rm[1], yf, a.rm[2] * zf));
        fy = fmaf(a.rm[3], xf, fmaf(a.rm[4], yf, a.rm[5] * zf));
        fz = fmaf(a.rm[6], xf, fmaf(a.rm[7], yf, a.rm[8] * zf));
    }
    const float rxy2 = fmaf(fx, fx, fy * fy), r2 = fmaf(fz, fz, rxy2);
    const float tp = (fast_atan2f(fy, fx) + 3.14159265358979f) * phi_scale;     // position in phi-bin units
    const float tc = fmaf(fz, rsqrtf(r2), 1.0f) * cos_scale;
    const float kpf = floorf(tp), kcf = floorf(tc);
    // inside the bin by more than the guard band on both sides:  |frac - 1/2| < 1/2 - guard   (hp, hc = 1/2 - guard).
    // A position outside the grid (or NaN) fails this by itself: tp, tc only leave [0, n] by rounding, i.e. next to an edge.
    const bool sure = (int)(fabsf((tp - kpf) - 0.5f) < hp) & (int)(fabsf((tc - kcf) - 0.5f) < hc) & (int)(rxy2 > 4e-3f * r2) &
                      (int)(r2 > 1e-30f);
    const int bin = (int)kpf * a.ncos + (int)kcf;
    unsigned int *addr = sure ? h + (PK ? bin >> 1 : bin) : mask + (idx >> 5);
    atomicAdd(addr, sure ? (PK ? 1u << ((bin & 1) << 4) : 1u) : (1u << (idx & 31)));
}

// The samples of one range, by ONE WAVE, through the fast classification (sums into s, decided samples into h, the others
// marked in the wave's mask).  16-byte loads of 4 consecutive frames, software-pipelined: a lane owns groups lane + 64 k of
// the range and always has the loads of the NEXT two groups (6 x 16 B) in flight while it classifies the current two.
template <bool ROT, bool INB>
__device__ __forceinline__ void vh_range(const VhArgs &a, const float *px, const float *py, const float *pz, int64_t start,
                                         int64_t end, int lane, unsigned int *h, unsigned int *mask, VhAcc &s, float phi_scale,
                                         float cos_scale, float hp, float hc)
{
    int64_t n0 = start;
    if ((start & 3) == 0) {
        const int64_t nvec = (end - start) >> 2;
        float4 XA[2], YA[2], ZA[2], XB[2], YB[2], ZB[2];
#define SR_VH_LOAD2(G, X, Y, Z)                                                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                          \
            int64_t q = (G) + lane + 64 * j;                                                     \
            q = q < nvec ? q : nvec - 1;                                                         \
            const int64_t n = start + (q << 2);                                                  \
            X[j] = *reinterpret_cast<const float4 *>(px + n);                                    \
            Y[j] = *reinterpret_cast<const float4 *>(py + n);                                    \
            Z[j] = *reinterpret_cast<const float4 *>(pz + n);                                    \
        }
#define SR_VH_COMP2(G, X, Y, Z)                                                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                          \
            const int64_t q = (G) + lane + 64 * j;                                               \
            if (q < nvec) {                                                                      \
                const unsigned int i0 = (unsigned int)(q << 2);                                  \
                vh_sample<ROT, INB>(a, X[j].x, Y[j].x, Z[j].x, i0 + 0, h, mask, s, phi_scale, cos_scale, hp, hc); \
                vh_sample<ROT, INB>(a, X[j].y, Y[j].y, Z[j].y, i0 + 1, h, mask, s, phi_scale, cos_scale, hp, hc); \
                vh_sample<ROT, INB>(a, X[j].z, Y[j].z, Z[j].z, i0 + 2, h, mask, s, phi_scale, cos_scale, hp, hc); \
                vh_sample<ROT, INB>(a, X[j].w, Y[j].w, Z[j].w, i0 + 3, h, mask, s, phi_scale, cos_scale, hp, hc); \
            }                                                                                    \
        }
        if (nvec > 0) {
            SR_VH_LOAD2(0, XA, YA, ZA)
            for (int64_t g0 = 0; g0 < nvec; g0 += 256) {
                SR_VH_LOAD2(g0 + 128, XB, YB, ZB)
                SR_VH_COMP2(g0, XA, YA, ZA)
                SR_VH_LOAD2(g0 + 256, XA, YA, ZA)
                SR_VH_COMP2(g0 + 128, XB, YB, ZB)
            }
        }
#undef SR_VH_LOAD2
#undef SR_VH_COMP2
        n0 = start + (nvec << 2);
    }
    for (int64_t n = n0 + lane; n < end; n += 64)
        vh_sample<ROT, INB>(a, px[n], py[n], pz[n], (unsigned int)(n - start), h, mask, s, phi_scale, cos_scale, hp, hc);
}

// grid = (groups of 4 ranges, vectors): a workgroup keeps one vector's histogram in LDS; each of its 4 waves streams one
// range by itself (no barrier, no cross-wave reduction while streaming: a wave reduces its own nine sums with DPP and lane 0
// stores them), then the workgroup classifies the parked samples of all four ranges together and flushes the histogram.
constexpr int kRangesPerWG = 4;
constexpr int kListCap = 4096;       // parked samples a workgroup classifies densely; beyond that (pathological input:
                                     // every sample undecided) the collecting thread classifies them itself

__global__ __launch_bounds__(256, 4) void k_vechist(VhArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *edges = reinterpret_cast<double *>(smem);                 // nphi+1 + ncos+1
    const int ne = a.nphi + 1 + a.ncos + 1;
    unsigned int *h = reinterpret_cast<unsigned int *>(edges + ne);   // nphi*ncos
    const int nbins = a.nphi * a.ncos;
    unsigned int *mask = h + ((nbins + 3) & ~3);                      // 4 x kMaxRange bits: undecided samples per wave / range
    unsigned int *qcount = mask + kRangesPerWG * (kMaxRange / 32);    // their number (+ pad) ...
    unsigned short *qlist = reinterpret_cast<unsigned short *>(qcount + 4);   // ... and (wave << 13 | position in the range)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t v = blockIdx.y;

    for (int i = tid; i < ne; i += 256) edges[i] = a.edges ? a.edges[i] : a.edges_inline[i];
    for (int i = tid; i < nbins; i += 256) h[i] = 0u;
    for (int i = tid; i < kRangesPerWG * (kMaxRange / 32); i += 256) mask[i] = 0u;
    if (tid == 0) *qcount = 0u;
    __syncthreads();
    const double *ephi = edges, *ecos = edges + a.nphi + 1;
    // the float32 estimate assumes the uniform numpy.linspace edges of calculate-Ct-from-traj.py:618
    const float phi_scale = (float)((double)a.nphi / (ephi[a.nphi] - ephi[0]));
    const float cos_scale = (float)((double)a.ncos / (ecos[a.ncos] - ecos[0]));
    const float hp = 0.5f - fmaxf(kEdgeGuard, kPhiGuardRad * phi_scale), hc = 0.5f - fmaxf(kEdgeGuard, kCosGuard * cos_scale);
    const float *px = a.soa + (v * 3) * a.Npad;
    const float *py = px + a.Npad;
    const float *pz = py + a.Npad;

    const int rid = blockIdx.x * kRangesPerWG + wave;                 // this wave's range
    int64_t start = 0, end = 0;
    if (rid < a.nranges) {
        bool in_block;
        if (rid < a.nB * a.m) {
            const int b = rid / a.m, i = rid - b * a.m;
            start = (int64_t)b * a.Fb + (int64_t)i * a.sub;
            end = min(start + a.sub, (int64_t)(b + 1) * a.Fb);
            in_block = true;
        } else {                                                      // frames behind the last full S2 block
            start = (int64_t)a.nB * a.Fb + (int64_t)(rid - a.nB * a.m) * a.sub;
            end = min(start + a.sub, a.N);
            in_block = false;
        }
        unsigned int *wmask = mask + wave * (kMaxRange / 32);
        VhAcc s = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (a.rotate) {
            if (in_block) vh_range<true, true>(a, px, py, pz, start, end, lane, h, wmask, s, phi_scale, cos_scale, hp, hc);
            else vh_range<true, false>(a, px, py, pz, start, end, lane, h, wmask, s, phi_scale, cos_scale, hp, hc);
        } else {
            if (in_block) vh_range<false, true>(a, px, py, pz, start, end, lane, h, wmask, s, phi_scale, cos_scale, hp, hc);
            else vh_range<false, false>(a, px, py, pz, start, end, lane, h, wmask, s, phi_scale, cos_scale, hp, hc);
        }
        // the wave's nine sums (fixed order: DPP butterfly), stored by lane 0..8
        double vals[9] = {s.sx, s.sy, s.sz, s.oxx, s.oyy, s.ozz, s.oxy, s.oxz, s.oyz};
        double mine = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double t = sr_wave_sum_f64(vals[k]);
            if (lane == k) mine = t;
        }
        if (lane < 9) a.partials[(v * a.nranges + rid) * 9 + lane] = mine;
    }
    __syncthreads();                                                  // every sample of the four ranges is counted or marked
    // collect the undecided samples (four mask words per thread), then classify them densely, one per thread
    for (int w = tid; w < kRangesPerWG * (kMaxRange / 32); w += 256) {
        unsigned int bits = mask[w];
        const int wv = w / (kMaxRange / 32), w0 = w - wv * (kMaxRange / 32);
        while (bits) {
            const int bpos = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const unsigned int slot = atomicAdd(qcount, 1u);
            const unsigned int pos = (unsigned int)(w0 * 32 + bpos);
            if (slot < (unsigned int)kListCap) {
                qlist[slot] = (unsigned short)((wv << 13) | pos);
            } else {
                const int r2 = blockIdx.x * kRangesPerWG + wv;
                const int64_t st2 = r2 < a.nB * a.m ? (int64_t)(r2 / a.m) * a.Fb + (int64_t)(r2 % a.m) * a.sub
                                                     : (int64_t)a.nB * a.Fb + (int64_t)(r2 - a.nB * a.m) * a.sub;
                vh_exact(a, px[st2 + pos], py[st2 + pos], pz[st2 + pos], ephi, ecos, h);
            }
        }
    }
    __syncthreads();
    {
        const unsigned int nq = min(*qcount, (unsigned int)kListCap);
        for (unsigned int i = tid; i < nq; i += 256) {
            const unsigned int e = qlist[i];
            const int wv = (int)(e >> 13);
            const int r2 = blockIdx.x * kRangesPerWG + wv;
            const int64_t st2 = r2 < a.nB * a.m ? (int64_t)(r2 / a.m) * a.Fb + (int64_t)(r2 % a.m) * a.sub
                                                 : (int64_t)a.nB * a.Fb + (int64_t)(r2 - a.nB * a.m) * a.sub;
            const int64_t n = st2 + (e & 8191u);
            vh_exact(a, px[n], py[n], pz[n], ephi, ecos, h);
        }
    }
    __syncthreads();
    unsigned int *gh = a.hist_u32 + v * nbins;
    for (int i = tid; i < nbins; i += 256) {
        const unsigned int c = h[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

// histogram counters -> float64; per-range sums combined in fixed order, then rotated: sum R v = R sum v and
// sum (R v)(R v)^T = R (sum v v^T) R^T with R the rotation matrix of the unit quaternion (identity without rotation)
struct VhRot {
    double R[3][3];
};

// per-range sums combined in fixed order, then rotated (see above)
__device__ __forceinline__ void vh_finalize_sums(const double *__restrict__ partials, int64_t v, int64_t nV, int nranges, int nB,
                                                 int m, const VhRot &rot, double *__restrict__ vecsum,
                                                 double *__restrict__ outer, int tid)
{
#pragma clang fp contract(off)
    const double *p = partials + v * nranges * 9;
    if (vecsum && tid == 0) {
        double s[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < nranges; ++r)
            for (int k = 0; k < 3; ++k) s[k] += p[r * 9 + k];
        for (int i = 0; i < 3; ++i) vecsum[v * 3 + i] = (rot.R[i][0] * s[0] + rot.R[i][1] * s[1]) + rot.R[i][2] * s[2];
    }
    if (outer) {
        for (int b = tid; b < nB; b += 256) {
            double q[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < m; ++j)
                for (int k = 0; k < 6; ++k) q[k] += p[(b * m + j) * 9 + 3 + k];
            const double M[3][3] = {{q[0], q[3], q[4]}, {q[3], q[1], q[5]}, {q[4], q[5], q[2]}};
            double T[3][3], O[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) T[i][j] = (rot.R[i][0] * M[0][j] + rot.R[i][1] * M[1][j]) + rot.R[i][2] * M[2][j];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) O[i][j] = (T[i][0] * rot.R[j][0] + T[i][1] * rot.R[j][1]) + T[i][2] * rot.R[j][2];
            double *o = outer + ((int64_t)b * nV + v) * 6;
            o[0] = O[0][0]; o[1] = O[1][1]; o[2] = O[2][2]; o[3] = O[0][1]; o[4] = O[0][2]; o[5] = O[1][2];
        }
    }
}

__global__ __launch_bounds__(256) void k_vechist_finalize(const unsigned int *__restrict__ hist_u32,
                                                          const double *__restrict__ partials, int64_t nV, int nbins,
                                                          int nranges, int nB, int m, VhRot rot, double *__restrict__ hist,
                                                          double *__restrict__ vecsum, double *__restrict__ outer)
{
    const int64_t v = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < nbins; i += 256) hist[v * nbins + i] = (double)hist_u32[v * nbins + i];
    vh_finalize_sums(partials, v, nV, nranges, nB, m, rot, vecsum, outer, tid);
}

// ------------------------------------------------------------------------------------------
// kernel 0 + kernel 2 in ONE pass over the frame-major input: per-vector planes (what k_pack_soa writes) AND the rotated
// Lambert histogram, mean-vector and S2 sums (what k_vechist computes from the planes).  The planes are then only read by
// kernel 1: 615 MB of HBM reads and one launch per batch less.
//
// A workgroup owns kPhVec = 8 consecutive vectors and one frame range (<= kMaxRange frames, inside one S2 block): eight
// histograms live in LDS as packed 16-bit counters (8 x 2592 x 2 B = 41 KB), so a range's counts need no global atomics
// at all -- they leave as one coalesced store of the packed words and k_packhist_finalize adds the ranges up (+128 MB of
// traffic against 615 MB saved; device-scope atomics run at the memory side, ~2e10/s for scattered words: the ~2e6 non-
// empty (range, vector, bin) cells would cost more than the pass itself).
// Reads: a lane loads one sample (12 bytes, x y z of one vector in one frame); a wave covers 8 frames x 8 vectors = eight
// 96-byte runs.  Four neighbouring vector groups share the three 128-byte lines of a frame row; their workgroups are
// dealt to the SAME XCD back to back (blockIdx -> item mapping below; blocks b and b + 8 share an XCD), so the lines
// are fetched from HBM once and served from that XCD's L2 to the other three.  Classification happens on the sample while
// it is in registers (the fast float32 path of k_vechist, undecided samples parked in a bit mask and classified
// exactly afterwards from the input array); the samples then go through the wave's own LDS tile (8 vectors x 3 components x
// 64 frames) and out to the planes as 16-byte stores.  No workgroup barrier while streaming: every wave runs its own
// software-pipelined loop (first version: one 256-frame tile per workgroup, two barriers per tile, loads not overlapped
// with anything -- 0.55 ms against 0.23 + 0.18 ms for the two kernels it replaces).
// ------------------------------------------------------------------------------------------
constexpr int kPhVec = 8;
constexpr int kPhTile = 64;               // frames per wave tile
constexpr int kPhRow = kPhTile + 4;        // floats per tile row: the 64 lanes of a column write hit 32 different banks

struct PhRange {
    long long start;                       // first frame
    int len;                               // frames (1 .. kMaxRange)
    int rid;                               // histogram range index (layout of VhArgs::partials), -1 = pack only
    int inb;                               // 1: inside an S2 block (all nine sums), 0: tail (vector sums only)
    int pad;
};

struct PhArgs {
    VhArgs vh;                             // rotation, edges, nphi / ncos (soa / N / ranges of VhArgs unused)
    const float *vecs;                     // (Ntot, Vtot, 3)
    int64_t Ntot, Vtot, v0, nV;
    float *soa;                            // (nV, 3, Npad)
    int64_t Npad;
    const PhRange *ranges;
    int nitems, ngroups, nhist, nbw;       // nitems = ranges x (ngroups / 4); ngroups = vector groups, padded to 4
    unsigned int *hist_part;               // (nhist, nV, nbw) packed 16-bit counter pairs
};

__global__ __launch_bounds__(256, 2) void k_pack_hist(PhArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const VhArgs &vh = a.vh;
    const int ne = vh.nphi + 1 + vh.ncos + 1;
    double *edges = reinterpret_cast<double *>(smem);
    double *wsum = edges + ((ne + 1) & ~1);                                         // 4 waves x 8 vectors x 9 sums
    unsigned int *h = reinterpret_cast<unsigned int *>(wsum + 4 * kPhVec * 9);       // kPhVec x nbw4 packed counters
    const int nbw4 = (a.nbw + 3) & ~3;
    unsigned int *mask = h + kPhVec * nbw4;                                          // kPhVec x kMaxRange bits
    float *tile = reinterpret_cast<float *>(mask + kPhVec * (kMaxRange / 32));       // per wave: 3 kPhVec rows of kPhRow floats
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // item of this workgroup: the four vector groups of a quad (same frame range) sit in consecutive slots of one XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int it = (slot >> 2) * 8 + xcd;
    if (it >= a.nitems) return;
    const int nquads = a.ngroups >> 2;
    const int ridx = it / nquads, grp = (it - ridx * nquads) * 4 + (slot & 3);
    const int64_t vb = (int64_t)grp * kPhVec;                     // first vector of the group, relative to v0
    if (vb >= a.nV) return;
    const int nvec = (int)min((int64_t)kPhVec, a.nV - vb);
    const PhRange rg = a.ranges[ridx];
    const bool do_hist = rg.rid >= 0;

    for (int i = tid; i < ne; i += 256) edges[i] = vh.edges ? vh.edges[i] : vh.edges_inline[i];
    for (int i = tid; i < kPhVec * nbw4; i += 256) h[i] = 0u;
    for (int i = tid; i < kPhVec * (kMaxRange / 32); i += 256) mask[i] = 0u;
    __syncthreads();
    const double *ephi = edges, *ecos = edges + vh.nphi + 1;
    const float phi_scale = (float)((double)vh.nphi / (ephi[vh.nphi] - ephi[0]));
    const float cos_scale = (float)((double)vh.ncos / (ecos[vh.ncos] - ecos[0]));
    const float hp = 0.5f - fmaxf(kEdgeGuard, kPhiGuardRad * phi_scale), hc = 0.5f - fmaxf(kEdgeGuard, kCosGuard * cos_scale);

    // Every WAVE streams its own 64-frame tiles (tile w, w + 4, ... of the range) without a workgroup barrier: loads of the
    // next tile in flight (24 registers) while the current one is classified, staged through the wave's private LDS tile
    // (8 vectors x 3 components x 64 frames) and written to the planes.  LDS operations of one wave complete in order, so
    // the staging needs fences for the compiler only.
    const int vl = lane & 7, fl = lane >> 3;                      // this lane's vector of the group and frame of a pass
    const bool vok = vl < nvec;
    const float *src = a.vecs + (a.v0 + vb + (vok ? vl : 0)) * 3;
    unsigned int *hv = h + vl * nbw4, *mv = mask + vl * (kMaxRange / 32);
    float *wtile = tile + wave * (3 * kPhVec * kPhRow);
    float *trow = wtile + (vl * 3) * kPhRow;
    VhAcc s = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t end = rg.start + rg.len;
    const bool aligned = ((rg.start | a.Npad) & 3) == 0;          // 16-byte plane stores
    const int ntiles = (rg.len + kPhTile - 1) / kPhTile;
    float XA[8], YA[8], ZA[8], XB[8], YB[8], ZB[8];
    // all 24 loads of a tile (unconditional, from a clamped frame: a conditional load is a branch per sample)
#define SR_PH_LOAD(T, X, Y, Z)                                                                   \
    {                                                                                            \
        const int64_t t0_ = rg.start + (int64_t)(T) * kPhTile;                                   \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                          \
            const int64_t fr = t0_ + p * 8 + fl;                                                 \
            const bool ok = vok && fr < end && fr < a.Ntot;                                      \
            const float *q = src + (ok ? fr : 0) * a.Vtot * 3;                                   \
            const float x = q[0], y = q[1], z = q[2];                                            \
            X[p] = ok ? x : 0.f; Y[p] = ok ? y : 0.f; Z[p] = ok ? z : 0.f;                       \
        }                                                                                        \
    }
#define SR_PH_TILE(T, X, Y, Z)                                                                   \
    {                                                                                            \
        const int64_t t0 = rg.start + (int64_t)(T) * kPhTile;                                    \
        const int tl = (int)min((int64_t)kPhTile, end - t0);                                     \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                          \
            const int o = p * 8 + fl;                                                            \
            const bool ok = vok && o < tl && t0 + o < a.Ntot;                                    \
            if (do_hist && ok) {                                                                 \
                const unsigned int idx = (unsigned int)(t0 - rg.start) + (unsigned int)o;        \
                if (vh.rotate) {                                                                 \
                    if (rg.inb) vh_sample<true, true, true>(vh, X[p], Y[p], Z[p], idx, hv, mv, s, phi_scale, cos_scale, hp, hc);  \
                    else vh_sample<true, false, true>(vh, X[p], Y[p], Z[p], idx, hv, mv, s, phi_scale, cos_scale, hp, hc);       \
                } else {                                                                         \
                    if (rg.inb) vh_sample<false, true, true>(vh, X[p], Y[p], Z[p], idx, hv, mv, s, phi_scale, cos_scale, hp, hc); \
                    else vh_sample<false, false, true>(vh, X[p], Y[p], Z[p], idx, hv, mv, s, phi_scale, cos_scale, hp, hc);      \
                }                                                                                \
            }                                                                                    \
            trow[o] = X[p];                                                                      \
            trow[kPhRow + o] = Y[p];                                                             \
            trow[2 * kPhRow + o] = Z[p];                                                         \
        }                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                   \
        __builtin_amdgcn_wave_barrier();                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                   \
        /* planes: row k = (vector, component) of the group, 16 groups of four frames per row */ \
        _Pragma("unroll") for (int j = 0; j < 3 * kPhVec * (kPhTile / 4) / 64; ++j) {            \
            const int idx = lane + 64 * j;                                                       \
            const int k = idx >> 4, q4 = (idx & 15) * 4;                                         \
            if (k < nvec * 3 && q4 < tl) {                                                       \
                const float4 val = *reinterpret_cast<const float4 *>(wtile + k * kPhRow + q4);  \
                float *dst = a.soa + (vb * 3 + k) * a.Npad + t0 + q4;                            \
                if (aligned && q4 + 3 < tl) {                                                    \
                    *reinterpret_cast<float4 *>(dst) = val;                                      \
                } else {                                                                         \
                    const float e[4] = {val.x, val.y, val.z, val.w};                             \
                    for (int u = 0; u < 4 && q4 + u < tl; ++u) dst[u] = e[u];                    \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                   \
        __builtin_amdgcn_wave_barrier();                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                   \
    }
    if (wave < ntiles) {
        SR_PH_LOAD(wave, XA, YA, ZA)
        for (int t = wave; t < ntiles; t += 8) {
            if (t + 4 < ntiles) SR_PH_LOAD(t + 4, XB, YB, ZB)
            SR_PH_TILE(t, XA, YA, ZA)
            if (t + 4 < ntiles) {
                if (t + 8 < ntiles) SR_PH_LOAD(t + 8, XA, YA, ZA)
                SR_PH_TILE(t + 4, XB, YB, ZB)
            }
        }
    }
#undef SR_PH_LOAD
#undef SR_PH_TILE
    __syncthreads();
    if (!do_hist) return;

    // the nine sums: lanes lane, lane ^ 8, ^ 16, ^ 32 hold the same vector; then the four waves in fixed order
    {
        double vals[9] = {s.sx, s.sy, s.sz, s.oxx, s.oyy, s.ozz, s.oxy, s.oxz, s.oyz};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            double t = vals[k];
            t += __shfl_xor(t, 8, 64);
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (lane < kPhVec) wsum[(wave * kPhVec + lane) * 9 + k] = t;
        }
    }
    __syncthreads();                                              // also: every sample of the range is counted or marked
    if (tid < kPhVec * 9) {
        const int v8 = tid / 9, k = tid - v8 * 9;
        if (v8 < nvec) {
            const double t = ((wsum[(0 * kPhVec + v8) * 9 + k] + wsum[(1 * kPhVec + v8) * 9 + k]) + wsum[(2 * kPhVec + v8) * 9 + k]) +
                             wsum[(3 * kPhVec + v8) * 9 + k];
            vh.partials[((vb + v8) * vh.nranges + rg.rid) * 9 + k] = t;
        }
    }
    // undecided samples: exact classification from the input array
    {
        const int wpr = (rg.len + 31) >> 5;                       // mask words per vector in use
        for (int wi = tid; wi < nvec * wpr; wi += 256) {
            const int v8 = wi / wpr, w0 = wi - v8 * wpr;
            unsigned int bits = mask[v8 * (kMaxRange / 32) + w0];
            while (bits) {
                const int bpos = __ffs((int)bits) - 1;
                bits &= bits - 1u;
                const int64_t fr = rg.start + w0 * 32 + bpos;
                const float *q = a.vecs + (fr * a.Vtot + a.v0 + vb + v8) * 3;
                vh_exact<true>(vh, q[0], q[1], q[2], ephi, ecos, h + v8 * nbw4);
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < nvec * a.nbw; idx += 256) {
        const int v8 = idx / a.nbw, i = idx - v8 * a.nbw;
        a.hist_part[((int64_t)rg.rid * a.nV + vb + v8) * a.nbw + i] = h[v8 * nbw4 + i];
    }
}

// histogram = sum over the ranges of the packed 16-bit counters; sums as in k_vechist_finalize
__global__ __launch_bounds__(256) void k_packhist_finalize(const unsigned int *__restrict__ hist_part,
                                                           const double *__restrict__ partials, int64_t nV, int nbins, int nbw,
                                                           int nranges, int nB, int m, VhRot rot, double *__restrict__ hist,
                                                           double *__restrict__ vecsum, double *__restrict__ outer)
{
    const int64_t v = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < nbw; i += 256) {
        unsigned int lo = 0u, hi = 0u;
        for (int r = 0; r < nranges; ++r) {
            const unsigned int w = hist_part[((int64_t)r * nV + v) * nbw + i];
            lo += w & 0xFFFFu;
            hi += w >> 16;
        }
        hist[v * nbins + 2 * i] = (double)lo;
        if (2 * i + 1 < nbins) hist[v * nbins + 2 * i + 1] = (double)hi;
    }
    vh_finalize_sums(partials, v, nV, nranges, nB, m, rot, vecsum, outer, tid);
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
    // ranges: every S2 block is cut into m ranges of `sub` frames (a multiple of 4, at most kMaxRange: the LDS mask and
    // list of undecided samples are sized for it); enough of them that the grid fills the chip (>= ~1024 workgroups of
    // kRangesPerWG ranges each), each at least 1024 frames; the frames behind the last full block form further ranges
    int64_t want = (1024 * kRangesPerWG + nV - 1) / nV;
    int64_t per_block = (want + a.nB - 1) / a.nB;
    if (per_block < 1) per_block = 1;
    const int64_t maxm = (a.Fb + 1023) / 1024, minm = (a.Fb + kMaxRange - 1) / kMaxRange;
    if (per_block > maxm) per_block = maxm;
    if (per_block < minm) per_block = minm;
    a.m = (int)per_block;
    a.sub = sr_round_up((a.Fb + a.m - 1) / a.m, 4);
    if (a.sub > kMaxRange) a.sub = kMaxRange;
    a.m = (int)((a.Fb + a.sub - 1) / a.sub);
    const int64_t tail = N - (int64_t)a.nB * a.Fb;
    a.nranges = a.nB * a.m + (int)((tail + a.sub - 1) / a.sub);
    a.nphi = nphi; a.ncos = ncos;
    a.rotate = q_host ? 1 : 0;
    a.qw = 1; a.qx = a.qy = a.qz = 0;
    if (q_host) {
        double qn[4];
        normalise_q(q_host, qn);
        a.qw = qn[0]; a.qx = qn[1]; a.qy = qn[2]; a.qz = qn[3];
    }
    VhRot rot;
    {
        const double w = a.qw, x = a.qx, y = a.qy, z = a.qz;
        const double Rm[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                 {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                 {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                rot.R[i][j] = a.rotate ? Rm[i][j] : (i == j ? 1.0 : 0.0);
                a.rm[i * 3 + j] = (float)rot.R[i][j];
            }
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
    const size_t lds = (size_t)ne * sizeof(double) + (size_t)((nbins + 3) & ~3) * sizeof(unsigned int) +
                       (kRangesPerWG * (kMaxRange / 32) + 4) * sizeof(unsigned int) + (size_t)kListCap * sizeof(unsigned short);
    if (int rc = sr_grant_lds(ctx, SR_K_VECHIST, reinterpret_cast<const void *>(&k_vechist), lds)) return rc;
    hipLaunchKernelGGL(k_vechist, dim3((unsigned)((a.nranges + kRangesPerWG - 1) / kRangesPerWG), (unsigned)nV), dim3(256), lds, ctx->stream, a);
    SR_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_vechist_finalize, dim3((unsigned)nV), dim3(256), 0, ctx->stream, h32, partials, nV, nbins,
                       a.nranges, a.nB, a.m, rot, hist, vecsum, outer);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_pack_hist_f32_dev(sr_ctx *ctx, const float *vecs, int64_t Ntot, int64_t Vtot, int64_t v0, int64_t nV, float *soa,
                         int64_t Npad, int64_t N_hist, const int64_t *chunk_start_host, int64_t R, int64_t block_len,
                         const double *q_host, const double *edges_phi_host, int nphi, const double *edges_cos_host, int ncos,
                         double *hist, double *vecsum, double *outer)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && soa && hist && edges_phi_host && edges_cos_host, -2, "sr_pack_hist_f32_dev: null pointer");
    SR_REQUIRE(Ntot > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3,
               "sr_pack_hist_f32_dev: bad shape Ntot=%lld Vtot=%lld v0=%lld nV=%lld", (long long)Ntot, (long long)Vtot,
               (long long)v0, (long long)nV);
    SR_REQUIRE(Npad >= Ntot && Npad % 4 == 0, -3, "sr_pack_hist_f32_dev: Npad=%lld must be >= Ntot and a multiple of 4",
               (long long)Npad);
    SR_REQUIRE(nphi >= 1 && ncos >= 1 && nphi + ncos + 2 <= kInlineEdges, -6,
               "sr_pack_hist_f32_dev: %d x %d bins: use sr_pack_soa_f32_dev + sr_rotate_hist_f32_dev", nphi, ncos);
    const int nbins = nphi * ncos, nbw = (nbins + 1) / 2, nbw4 = (nbw + 3) & ~3;
    const int ne = nphi + 1 + ncos + 1;
    const size_t lds = (size_t)((ne + 1) & ~1) * sizeof(double) + (size_t)4 * kPhVec * 9 * sizeof(double) +
                       (size_t)kPhVec * nbw4 * sizeof(unsigned int) + (size_t)kPhVec * (kMaxRange / 32) * sizeof(unsigned int) +
                       (size_t)4 * 3 * kPhVec * kPhRow * sizeof(float);
    SR_REQUIRE(lds <= sr_lds_limit(ctx) / 2, -6,
               "sr_pack_hist_f32_dev: %d x %d bins need %zu B of LDS per workgroup (two must share a CU): use sr_pack_soa_f32_dev + "
               "sr_rotate_hist_f32_dev", nphi, ncos, lds);
    // ---- blocks (S2 blocks = the chunks of kernel 1 when chunk starts are given) and their frame ranges ----
    int64_t Fb, nB;
    if (chunk_start_host) {
        SR_REQUIRE(R >= 1 && block_len >= 1, -3, "sr_pack_hist_f32_dev: chunk starts need R >= 1 and block_len = frames per chunk");
        Fb = block_len; nB = R;
        for (int64_t r = 0; r < R; ++r)
            SR_REQUIRE(chunk_start_host[r] >= 0 && chunk_start_host[r] + Fb <= Ntot && (r == 0 || chunk_start_host[r] >= chunk_start_host[r - 1] + Fb),
                       -3, "sr_pack_hist_f32_dev: chunk %lld start %lld out of range or overlapping", (long long)r, (long long)chunk_start_host[r]);
    } else {
        SR_REQUIRE(N_hist >= 1 && N_hist <= Ntot, -3, "sr_pack_hist_f32_dev: N_hist=%lld out of range", (long long)N_hist);
        Fb = (block_len > 0 && block_len <= N_hist) ? block_len : N_hist;
        nB = N_hist / Fb;
    }
    const int64_t ngroups = sr_round_up((nV + kPhVec - 1) / kPhVec, 4);
    // ranges per block: enough workgroups to fill the chip (>= ~1024), each at least 1024 and at most kMaxRange frames
    int64_t m = (1024 + nB * ngroups - 1) / (nB * ngroups);
    const int64_t maxm = (Fb + 1023) / 1024, minm = (Fb + kMaxRange - 1) / kMaxRange;
    if (m > maxm) m = maxm;
    if (m < minm) m = minm;
    int64_t sub = sr_round_up((Fb + m - 1) / m, 4);
    if (sub > kMaxRange) sub = kMaxRange;
    m = (Fb + sub - 1) / sub;
    const int64_t tail = chunk_start_host ? 0 : N_hist - nB * Fb;
    const int64_t ntail = (tail + sub - 1) / sub;
    const int64_t nhist = nB * m + ntail;
    // table: histogram ranges first (their index is the slot of VhArgs::partials), then whatever else must reach the planes
    const int64_t psub = 4096;                // frames per pack-only range
    const size_t cap = (size_t)(nhist + Npad / psub + 2 * nB + 8);
    PhRange *tab = (PhRange *)calloc(cap, sizeof(PhRange));
    SR_REQUIRE(tab != nullptr, -5, "sr_pack_hist_f32_dev: out of host memory");
    size_t nr = 0;
    int64_t covered = 0;                      // frames [0, covered) are in the table
    auto pack_only = [&](int64_t from, int64_t to) {
        for (int64_t f = from; f < to && nr < cap; f += psub) {
            tab[nr].start = f; tab[nr].len = (int)(to - f < psub ? to - f : psub); tab[nr].rid = -1; tab[nr].inb = 0; ++nr;
        }
    };
    size_t nh = 0;
    {
        // first pass: histogram ranges in slot order; second pass appends the gaps
        for (int64_t b = 0; b < nB; ++b) {
            const int64_t bs = chunk_start_host ? chunk_start_host[b] : b * Fb;
            for (int64_t i = 0; i < m; ++i) {
                const int64_t st = bs + i * sub, en = st + sub < bs + Fb ? st + sub : bs + Fb;
                tab[nr].start = st; tab[nr].len = (int)(en - st); tab[nr].rid = (int)(b * m + i); tab[nr].inb = 1; ++nr;
            }
        }
        for (int64_t i = 0; i < ntail; ++i) {
            const int64_t st = nB * Fb + i * sub, en = st + sub < N_hist ? st + sub : N_hist;
            tab[nr].start = st; tab[nr].len = (int)(en - st); tab[nr].rid = (int)(nB * m + i); tab[nr].inb = 0; ++nr;
        }
        nh = nr;
        for (int64_t b = 0; b < nB; ++b) {
            const int64_t bs = chunk_start_host ? chunk_start_host[b] : b * Fb;
            if (bs > covered) pack_only(covered, bs);
            covered = bs + Fb;
        }
        if (!chunk_start_host) covered = N_hist;
        if (covered < Npad) pack_only(covered, Npad);
    }
    if (nr >= cap || (int64_t)nh != nhist) {
        free(tab);
        sr_set_error("sr_pack_hist_f32_dev: internal range table error");
        return -9;
    }
    const size_t tab_bytes = nr * sizeof(PhRange);
    PhRange *tab_d = (PhRange *)sr_workspace(ctx, SR_WS_TAB, tab_bytes);
    unsigned int *hpart = (unsigned int *)sr_workspace(ctx, SR_WS_HPART, (size_t)nhist * nV * nbw * sizeof(unsigned int));
    double *partials = (double *)sr_workspace(ctx, SR_WS_OUT3, (size_t)nV * nhist * 9 * sizeof(double));
    if (!tab_d || !hpart || !partials) { free(tab); return -5; }
    if (ctx->tab_shadow_bytes != tab_bytes || memcmp(ctx->tab_shadow, tab, tab_bytes) != 0) {
        // new geometry: synchronous upload (rare: a pipeline repeats the same table batch after batch)
        hipError_t e1 = hipStreamSynchronize(ctx->stream);
        hipError_t e2 = hipMemcpy(tab_d, tab, tab_bytes, hipMemcpyHostToDevice);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            free(tab);
            sr_set_error("sr_pack_hist_f32_dev: range table upload failed");
            return -100;
        }
        free(ctx->tab_shadow);
        ctx->tab_shadow = tab;
        ctx->tab_shadow_bytes = tab_bytes;
    } else {
        free(tab);
    }
    PhArgs a;
    memset(&a, 0, sizeof(a));
    VhArgs &vh = a.vh;
    vh.nphi = nphi; vh.ncos = ncos; vh.nranges = (int)nhist; vh.nB = (int)nB; vh.m = (int)m;
    vh.rotate = q_host ? 1 : 0;
    vh.qw = 1; vh.qx = vh.qy = vh.qz = 0;
    if (q_host) {
        double qn[4];
        normalise_q(q_host, qn);
        vh.qw = qn[0]; vh.qx = qn[1]; vh.qy = qn[2]; vh.qz = qn[3];
    }
    VhRot rot;
    {
        const double w = vh.qw, x = vh.qx, y = vh.qy, z = vh.qz;
        const double Rm[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                 {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                 {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                rot.R[i][j] = vh.rotate ? Rm[i][j] : (i == j ? 1.0 : 0.0);
                vh.rm[i * 3 + j] = (float)rot.R[i][j];
            }
    }
    for (int i = 0; i <= nphi; ++i) vh.edges_inline[i] = edges_phi_host[i];
    for (int i = 0; i <= ncos; ++i) vh.edges_inline[nphi + 1 + i] = edges_cos_host[i];
    vh.edges = nullptr;
    vh.partials = partials;
    a.vecs = vecs; a.Ntot = Ntot; a.Vtot = Vtot; a.v0 = v0; a.nV = nV; a.soa = soa; a.Npad = Npad;
    a.ranges = tab_d; a.ngroups = (int)ngroups; a.nitems = (int)(nr * (size_t)(ngroups / 4)); a.nhist = (int)nhist; a.nbw = nbw;
    a.hist_part = hpart;
    if (int rc = sr_grant_lds(ctx, SR_K_PACKHIST, reinterpret_cast<const void *>(&k_pack_hist), lds)) return rc;
    const int64_t nblocks = sr_round_up(a.nitems, 8) * 4;
    SR_REQUIRE(nblocks < ((int64_t)1 << 31), -3, "sr_pack_hist_f32_dev: too many workgroups");
    hipLaunchKernelGGL(k_pack_hist, dim3((unsigned)nblocks), dim3(256), lds, ctx->stream, a);
    SR_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_packhist_finalize, dim3((unsigned)nV), dim3(256), 0, ctx->stream, hpart, partials, nV, nbins, nbw,
                       (int)nhist, (int)nB, (int)m, rot, hist, vecsum, outer);
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
