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

// exact (reference-order, float64) classification of one sample
__device__ __forceinline__ void vh_exact(const VhArgs &a, float xf, float yf, float zf, const double *ephi, const double *ecos,
                                         unsigned int *h)
{
    const double x = (double)xf, y = (double)yf, z = (double)zf;
    double rx = x, ry = y, rz = z, phi, c;
    if (a.rotate) rotate_q(a.qw, a.qx, a.qy, a.qz, x, y, z, rx, ry, rz);
    exact_phi_cos(rx, ry, rz, phi, c);
    const int kp = np_bin(ephi, a.nphi, phi), kc = np_bin(ecos, a.ncos, c);
    if (kp >= 0 && kc >= 0) atomicAdd(&h[kp * a.ncos + kc], 1u);
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
template <bool ROT, bool INB>
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
    unsigned int *addr = sure ? h + bin : mask + (idx >> 5);
    atomicAdd(addr, sure ? 1u : (1u << (idx & 31)));
}

// The samples of one range, by ONE WAVE, through the fast classification (sums into s, decided samples into h, the others
// marked in the wave's mask).  16-byte loads of 4 consecutive frames, software-pipelined: a lane owns groups lane + 64 k of
// the range and always has the loads of the NEXT two groups (6 x 16 B) in flight while it classifies the current two.
// 16 bytes of a plane, non-temporal: the histogram is the last reader of the planes (alone 0.171 against 0.177 ms)
__device__ __forceinline__ float4 vh_load4(const float *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

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
            X[j] = vh_load4(px + n);                                                             \
            Y[j] = vh_load4(py + n);                                                             \
            Z[j] = vh_load4(pz + n);                                                             \
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

__global__ __launch_bounds__(256) void k_vechist_finalize(const unsigned int *__restrict__ hist_u32,
                                                          const double *__restrict__ partials, int64_t nV, int nbins,
                                                          int nranges, int nB, int m, VhRot rot, double *__restrict__ hist,
                                                          double *__restrict__ vecsum, double *__restrict__ outer)
{
#pragma clang fp contract(off)
    const int64_t v = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < nbins; i += 256) hist[v * nbins + i] = (double)hist_u32[v * nbins + i];
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

int sr_rotate_hist_f32(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV, const double *q,
                       const double *edges_phi, int nphi, const double *edges_cos, int ncos, double *hist,
                       double *vecsum, double *outer, int64_t block_len)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && hist, -2, "sr_rotate_hist_f32: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3, "sr_rotate_hist_f32: bad shape");
    sr_vectors *h = sr_vectors_create(ctx, nV, N);            // the rank's columns only (sr_vectors.hip)
    if (!h) return -5;
    int rc = sr_vectors_append_f32(ctx, h, vecs, N, Vtot, v0);
    if (!rc) rc = sr_vectors_hist_f32(ctx, h, N, q, edges_phi, nphi, edges_cos, ncos, hist, vecsum, outer, block_len);
    sr_vectors_destroy(ctx, h);
    return rc;
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
