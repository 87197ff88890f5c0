// sr_fit.hip -- kernel 3b: multi-exponential C(t) model, residual/Jacobian evaluation and a
// batched bounded non-linear least-squares solver that runs entirely on the device.
//
// Reference call site: conduct_curve_fitting, fitting_Ct_functions.py:322-324
//     curve_fit(curvefit_exponential, DeltaT, Decay, sigma=dDecay, p0=..., bounds=(0, [1..,tauMax..,1]))
// The arithmetic of that call lives in a third-party dependency that is not part of the reference
// tree: scipy (requirements.txt:2 "scipy>=0.17.1", unpinned; the build container has 1.15.3).
// With bounds, curve_fit -> least_squares(method='trf', jac='2-point', x_scale=1, ftol=xtol=gtol=1e-8,
// max_nfev=100*n, tr_solver='exact').  k_trf restates that published algorithm (Branch, Coleman & Li
// 1999 trust-region-reflective with More's 1977 trust-region sub-problem iteration, as implemented in
// scipy/optimize/_lsq/trf.py::trf_bounds and _lsq/common.py, and the forward-difference Jacobian of
// scipy/optimize/_numdiff.py::approx_derivative) so that the GPU follows the same iteration path:
// same step sizes h = sqrt(eps)*sign(x)*max(1,|x|) adjusted to the bounds, same Coleman-Li scaling,
// same reflective / Cauchy step selection, same radius update and termination tests.  The only
// structural difference: scipy factorises the augmented Jacobian with an SVD, here the n x n normal
// matrix B = (J d)^T (J d) + diag_h is Cholesky-factorised for every Levenberg parameter alpha that
// More's iteration visits -- phi(alpha) and phi'(alpha) are the same functions, evaluated differently.
//
// Mapping: one workgroup of W = 4 waves (one per SIMD of a CU) per residue.  The L data points are spread over the
// W*64 threads for model evaluation and the J^T J / J^T f reductions (wave shuffles, then a fixed-order
// combine through LDS); the n <= 11 dimensional algebra is workgroup-uniform, executed redundantly by
// every thread and held in registers (the kernel is templated on n so every small array is statically
// indexed).  The critical path of a fit is serial (up to 100 n dependent iterations), so the kernel is
// latency-bound by design: the W waves exist to shorten each iteration, not to raise throughput.
#include "sr_internal.h"

namespace {

constexpr double kEPS = 2.220446049250313e-16;
constexpr int kNmax = 11;

struct FitArgs {
    const double *t, *y, *sigma;      // (nRes, L); sigma may be null
    const double *p0;                 // (nRes, N)
    const unsigned char *skip;        // (nRes) or null: 1 = leave this residue alone
    double tau_max;
    int nRes, L, max_nfev, jac_mode;  // jac_mode 0: 2-point finite differences (scipy default), 1: analytic
    double ftol, xtol, gtol;
    double *popt, *pcov, *chisq;      // (nRes,N), (nRes,N,N), (nRes)
    int *status, *nfev;               // (nRes)
    double *fws;                      // (nRes, 2, L) residual work space
};

__device__ __forceinline__ double wsum(double v) { return sr_wave_sum_f64(v); }

// ---- model: curvefit_exponential, fitting_Ct_functions.py:419-427 -----------------------------
template <int N>
struct Model {
    static constexpr int K = N / 2;
    static constexpr bool kFreeS2 = (N % 2) == 1;

    __device__ static __forceinline__ double S2_of(const double *x)
    {
#pragma clang fp contract(off)
        if (kFreeS2) return x[N - 1];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += x[k];
        return 1.0 - s;
    }
    // e[k] = exp(-1.0*t/tau_k)
    __device__ static __forceinline__ void exps(const double *x, double t, double *e)
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < K; ++k) e[k] = exp((-1.0 * t) / x[K + k]);
    }
    __device__ static __forceinline__ double value(const double *x, const double *e)
    {
#pragma clang fp contract(off)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += x[k] * e[k];
        return S2_of(x) + s;
    }
};

// ---- small dense helpers on packed symmetric matrices (i >= j stored at i*(i+1)/2 + j) ----------
__host__ __device__ constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <int N>
__device__ __forceinline__ double dotN(const double *a, const double *b)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i] * b[i];
    return s;
}
template <int N>
__device__ __forceinline__ double normN(const double *a) { return sqrt(dotN<N>(a, a)); }

// s^T B s
template <int N>
__device__ __forceinline__ double quadN(const double *B, const double *s)
{
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) r += B[tri(i, j)] * s[j];
        q += s[i] * r;
    }
    return q;
}
// u^T B s
template <int N>
__device__ __forceinline__ double bilinN(const double *B, const double *u, const double *s)
{
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) r += B[tri(i, j)] * s[j];
        q += u[i] * r;
    }
    return q;
}

// Cholesky of (B + alpha I): Lf holds the strict lower triangle of L, inv[i] = 1/L_ii (the factor is only
// ever used through multiplications by these reciprocals: one rsqrt per column instead of a sqrt and
// n divisions -- float64 sqrt/div are ~15-instruction dependent chains and dominated the serial part).
// Returns false when a pivot is not positive; lmin2 = smallest squared diagonal of L.
template <int N>
__device__ __forceinline__ bool cholN(const double *B, double alpha, double *Lf, double *inv, double &lmin2)
{
    bool ok = true;
    lmin2 = 1e300;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = B[tri(i, j)] + (i == j ? alpha : 0.0);
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Lf[tri(i, k)] * Lf[tri(j, k)];
            if (i == j) {
                if (!(s > 0.0)) { ok = false; s = 1.0; }
                lmin2 = fmin(lmin2, s);
                inv[i] = rsqrt(s);
            } else {
                Lf[tri(i, j)] = s * inv[j];
            }
        }
    }
    return ok;
}
// solve L z = b (forward)
template <int N>
__device__ __forceinline__ void fwdN(const double *Lf, const double *inv, const double *b, double *z)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Lf[tri(i, k)] * z[k];
        z[i] = s * inv[i];
    }
}
// solve L^T p = z (backward)
template <int N>
__device__ __forceinline__ void bwdN(const double *Lf, const double *inv, const double *z, double *p)
{
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = z[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= Lf[tri(k, i)] * p[k];
        p[i] = s * inv[i];
    }
}

// phi(alpha) = ||p(alpha)|| - Delta, phi' = -(p^T (B+alpha I)^-1 p)/||p||, p = -(B+alpha I)^-1 g.
// (common.py:phi_and_derivative expressed through the Cholesky factor: with L L^T = B + alpha I,
//  z = L^-1 g, p = -L^-T z, q = L^-1 p:  p^T (B+alpha I)^-1 p = ||q||^2.)
template <int N>
__device__ __forceinline__ void phi_from_factor(const double *Lf, const double *inv, const double *g, double Delta,
                                                double *p, double &phi, double &phi_prime)
{
    double z[N], q[N];
    fwdN<N>(Lf, inv, g, z);
    bwdN<N>(Lf, inv, z, p);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = -p[i];
    fwdN<N>(Lf, inv, p, q);
    const double pn = normN<N>(p);
    phi = pn - Delta;
    phi_prime = -dotN<N>(q, q) / pn;
}

// common.py:solve_lsq_trust_region
template <int N>
__device__ void solve_tr(const double *B, const double *g, int m, double Delta, double &alpha, double *p)
{
    // rank test of scipy: s_min > EPS*m*s_max on the singular values of the augmented Jacobian.  Here:
    // B is accepted as full rank when its Cholesky factorisation succeeds with pivots above the
    // equivalent threshold (EPS*m)^2 * max diag.
    double Lf[N * (N + 1) / 2], inv[N], lmin2;
    bool full_rank = cholN<N>(B, 0.0, Lf, inv, lmin2);
    if (full_rank) {
        double dmax = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) dmax = fmax(dmax, B[tri(i, i)]);
        const double thr = kEPS * (double)m;
        if (!(lmin2 > thr * thr * dmax)) full_rank = false;
    }
    double alpha_upper = normN<N>(g) / Delta;
    double alpha_lower = 0.0;
    if (full_rank) {
        double phi, phip;
        phi_from_factor<N>(Lf, inv, g, Delta, p, phi, phip);     // p = Gauss-Newton step
        if (phi <= 0.0) { alpha = 0.0; return; }                 // norm(p) <= Delta
        alpha_lower = -phi / phip;
    }
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double phi, phip;
        cholN<N>(B, alpha, Lf, inv, lmin2);
        phi_from_factor<N>(Lf, inv, g, Delta, p, phi, phip);
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / phip;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    {
        double z[N];
        cholN<N>(B, alpha, Lf, inv, lmin2);
        fwdN<N>(Lf, inv, g, z);
        bwdN<N>(Lf, inv, z, p);
    }
    const double sc = -Delta / normN<N>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] *= sc;
}

// common.py:step_size_to_bound (returns min step; hits[i] = sign(s_i) where the minimum is attained)
template <int N>
__device__ __forceinline__ double step_to_bound(const double *x, const double *s, const double *lb, const double *ub,
                                                int *hits)
{
    double steps[N], mn = INFINITY;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        steps[i] = INFINITY;
        if (s[i] != 0.0) steps[i] = fmax((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]);
        mn = fmin(mn, steps[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) hits[i] = (steps[i] == mn) ? (s[i] > 0 ? 1 : (s[i] < 0 ? -1 : 0)) : 0;
    return mn;
}

// common.py:minimize_quadratic_1d
__device__ __forceinline__ void min_quad_1d(double a, double b, double lo, double hi, double c, double &t, double &y)
{
    t = lo;
    y = lo * (a * lo + b) + c;
    const double yh = hi * (a * hi + b) + c;
    if (yh < y) { y = yh; t = hi; }
    if (a != 0) {
        const double ex = -0.5 * b / a;
        if (lo < ex && ex < hi) {
            const double ye = ex * (a * ex + b) + c;
            if (ye < y) { y = ye; t = ex; }
        }
    }
}

// trf.py:select_step.  B = J_h^T J_h + diag_h, so every quadratic form of the original is a form in B.
template <int N>
__device__ void select_step(const double *x, const double *B, const double *g_h, double *p, double *p_h,
                            const double *d, double Delta, const double *lb, const double *ub, double theta,
                            double *step, double *step_h, double &predicted)
{
    bool inb = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double xn = x[i] + p[i];
        inb = inb && (xn >= lb[i]) && (xn <= ub[i]);
    }
    if (inb) {
        const double pv = 0.5 * quadN<N>(B, p_h) + dotN<N>(g_h, p_h);
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = p[i]; step_h[i] = p_h[i]; }
        predicted = -pv;
        return;
    }
    int hits[N];
    const double p_stride = step_to_bound<N>(x, p, lb, ub, hits);
    double r_h[N], r[N], xb[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r_h[i] = hits[i] != 0 ? -p_h[i] : p_h[i];
        r[i] = d[i] * r_h[i];
        p[i] *= p_stride;
        p_h[i] *= p_stride;
        xb[i] = x[i] + p[i];
    }
    // intersect_trust_region(p_h, r_h, Delta): positive root
    double to_tr;
    {
        const double a = dotN<N>(r_h, r_h), b = dotN<N>(p_h, r_h), c = dotN<N>(p_h, p_h) - Delta * Delta;
        const double dd = sqrt(b * b - a * c);
        const double q = -(b + copysign(dd, b));
        const double t1 = q / a, t2 = c / q;
        to_tr = t1 < t2 ? t2 : t1;
    }
    int hits2[N];
    const double to_bound = step_to_bound<N>(xb, r, lb, ub, hits2);
    double r_stride = fmin(to_bound, to_tr), r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = (1 - theta) * p_stride / r_stride;
        r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        // build_quadratic_1d(J_h, g_h, r_h, s0=p_h, diag=diag_h)
        const double a = 0.5 * quadN<N>(B, r_h);
        const double b = dotN<N>(g_h, r_h) + bilinN<N>(B, p_h, r_h);
        const double c = 0.5 * quadN<N>(B, p_h) + dotN<N>(g_h, p_h);
        min_quad_1d(a, b, r_stride_l, r_stride_u, c, r_stride, r_value);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            r_h[i] = r_h[i] * r_stride + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = INFINITY;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { p[i] *= theta; p_h[i] *= theta; }
    const double p_value = 0.5 * quadN<N>(B, p_h) + dotN<N>(g_h, p_h);

    double ag_h[N], ag[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    const double to_tr2 = Delta / normN<N>(ag_h);
    int hits3[N];
    const double to_bound2 = step_to_bound<N>(x, ag, lb, ub, hits3);
    double ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
    double ag_value;
    {
        const double a = 0.5 * quadN<N>(B, ag_h);
        const double b = dotN<N>(g_h, ag_h);
        min_quad_1d(a, b, 0.0, ag_stride, 0.0, ag_stride, ag_value);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }

    if (p_value < r_value && p_value < ag_value) {
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = p[i]; step_h[i] = p_h[i]; }
        predicted = -p_value;
    } else if (r_value < p_value && r_value < ag_value) {
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = r[i]; step_h[i] = r_h[i]; }
        predicted = -r_value;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = ag[i]; step_h[i] = ag_h[i]; }
        predicted = -ag_value;
    }
}

// common.py:make_strictly_feasible
template <int N>
__device__ __forceinline__ void strictly_feasible(double *x, const double *lb, const double *ub, double rstep)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double lower_dist = x[i] - lb[i], upper_dist = ub[i] - x[i];
        bool lo_act, up_act;
        if (rstep == 0) {
            lo_act = x[i] <= lb[i];
            up_act = x[i] >= ub[i];
        } else {
            const double lt = rstep * fmax(1.0, fabs(lb[i])), ut = rstep * fmax(1.0, fabs(ub[i]));
            lo_act = lower_dist <= fmin(upper_dist, lt);
            up_act = upper_dist <= fmin(lower_dist, ut);
        }
        double xn = x[i];
        if (rstep == 0) {
            if (lo_act) xn = nextafter(lb[i], ub[i]);
            if (up_act) xn = nextafter(ub[i], lb[i]);
        } else {
            if (lo_act) xn = lb[i] + rstep * fmax(1.0, fabs(lb[i]));
            if (up_act) xn = ub[i] - rstep * fmax(1.0, fabs(ub[i]));
        }
        if (xn < lb[i] || xn > ub[i]) xn = 0.5 * (lb[i] + ub[i]);
        x[i] = xn;
    }
}

template <int N, int W>
struct Trf {
    static constexpr int K = N / 2;
    static constexpr int NT = N * (N + 1) / 2;
    static constexpr int NTH = W * 64;
    using M = Model<N>;

    const double *t, *y, *sg;
    int L, tid;
    double *red;          // LDS: W x (NT + N + 2) partials

    // sum of v over the workgroup, identical in every thread (fixed combination order)
    template <int CNT>
    __device__ __forceinline__ void block_sums(double *vals) const
    {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const double w = wsum(vals[k]);
            if (lane == 0) red[wave * (NT + N + 2) + k] = w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            double acc = red[k];
#pragma unroll
            for (int w = 1; w < W; ++w) acc += red[w * (NT + N + 2) + k];
            vals[k] = acc;
        }
        __syncthreads();
    }

    __device__ __forceinline__ double weight(int l) const { return sg ? 1.0 / sg[l] : 1.0; }

    // residuals f = w*(model - y) into out; returns cost = 0.5 f.f ; finite=false when any f is not finite
    __device__ double eval_f(const double *x, double *out, bool &finite) const
    {
        double acc = 0.0, bad = 0.0;
        for (int l = tid; l < L; l += NTH) {
            double e[K > 0 ? K : 1];
            M::exps(x, t[l], e);
            double f;
            {
#pragma clang fp contract(off)
                f = weight(l) * (M::value(x, e) - y[l]);
            }
            out[l] = f;
            if (!isfinite(f)) bad = 1.0; else acc += f * f;
        }
        double v[2] = {acc, bad};
        block_sums<2>(v);
        finite = v[1] == 0.0;
        return 0.5 * v[0];
    }

    // J^T J (packed) and J^T f for the residual vector f0 stored in fbuf
    __device__ void eval_jac(const double *x, const double *lb, const double *ub, const double *fbuf, int mode,
                             double *A, double *g) const
    {
        double h[N], dx[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            // _numdiff.py:_compute_absolute_step + _adjust_scheme_to_bounds('1-sided', num_steps=1)
            const double sgn = x[i] >= 0 ? 1.0 : -1.0;
            double hi = 1.4901161193847656e-08 * sgn * fmax(1.0, fabs(x[i]));
            const double ld = x[i] - lb[i], ud = ub[i] - x[i];
            const double xp = x[i] + hi;
            const bool violated = (xp < lb[i]) || (xp > ub[i]);
            const bool fitting = fabs(hi) <= fmax(ld, ud);
            if (violated && fitting) hi = -hi;
            if (!fitting) hi = (ud >= ld) ? ud : -ld;
            h[i] = hi;
            dx[i] = (x[i] + hi) - x[i];
        }
        double Aacc[NT], gacc[N];
#pragma unroll
        for (int i = 0; i < NT; ++i) Aacc[i] = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) gacc[i] = 0.0;
        for (int l = tid; l < L; l += NTH) {
            const double tl = t[l], w = weight(l), f0 = fbuf[l];
            double e[K > 0 ? K : 1], Jr[N];
            M::exps(x, tl, e);
            if (mode == 0) {
                const double yl = y[l];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    // model at x + h_i e_i, recomputed the way fun(x + h) evaluates it
                    double xi[N], ei[K > 0 ? K : 1];
#pragma unroll
                    for (int j = 0; j < N; ++j) xi[j] = x[j];
                    xi[i] = x[i] + h[i];
#pragma unroll
                    for (int k = 0; k < K; ++k) ei[k] = e[k];
                    if (i >= K && i < 2 * K) {
#pragma clang fp contract(off)
                        ei[i - K] = exp((-1.0 * tl) / xi[i]);
                    }
                    double fi;
                    {
#pragma clang fp contract(off)
                        fi = w * (M::value(xi, ei) - yl);
                        Jr[i] = (fi - f0) / dx[i];
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    Jr[k] = w * (e[k] - (M::kFreeS2 ? 0.0 : 1.0));
                    Jr[K + k] = w * (x[k] * e[k] * tl / (x[K + k] * x[K + k]));
                }
                if (M::kFreeS2) Jr[N - 1] = w;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                gacc[i] += Jr[i] * f0;
#pragma unroll
                for (int j = 0; j <= i; ++j) Aacc[tri(i, j)] += Jr[i] * Jr[j];
            }
        }
        double all[NT + N];
#pragma unroll
        for (int i = 0; i < NT; ++i) all[i] = Aacc[i];
#pragma unroll
        for (int i = 0; i < N; ++i) all[NT + i] = gacc[i];
        block_sums<NT + N>(all);
#pragma unroll
        for (int i = 0; i < NT; ++i) A[i] = all[i];
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = all[NT + i];
    }
};

template <int N, int W>
__global__ __launch_bounds__(W * 64) void k_trf(FitArgs a)
{
    constexpr int K = N / 2;
    constexpr int NT = N * (N + 1) / 2;
    __shared__ double red_lds[W * (NT + N + 2)];
    const int res = blockIdx.x;
    if (a.skip && a.skip[res]) return;
    const int tid = threadIdx.x;
    Trf<N, W> T;
    T.L = a.L;
    T.tid = tid;
    T.red = red_lds;
    T.t = a.t + (int64_t)res * a.L;
    T.y = a.y + (int64_t)res * a.L;
    T.sg = a.sigma ? a.sigma + (int64_t)res * a.L : nullptr;
    double *fcur = a.fws + (int64_t)res * 2 * a.L, *fnew = fcur + a.L;
    const int m = a.L;

    double x[N], lb[N], ub[N];
    bool inb = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i] = a.p0[(int64_t)res * N + i];
        lb[i] = 0.0;
        ub[i] = (i >= K && i < 2 * K) ? a.tau_max : 1.0;
        inb = inb && (x[i] >= lb[i]) && (x[i] <= ub[i]);
    }
    int status = -99, nfev = 0;
    double cost = INFINITY;
    double A[NT], g[N];
#pragma unroll
    for (int i = 0; i < NT; ++i) A[i] = 0.0;
    bool have_fit = false;

    if (!inb) {
        status = -2;              // least_squares: "Initial guess is outside of provided bounds" (ValueError)
    } else {
        strictly_feasible<N>(x, lb, ub, 1e-10);
        bool finite;
        cost = T.eval_f(x, fcur, finite);
        nfev = 1;
        if (!finite) {
            status = -3;          // "Residuals are not finite in the initial point"
        } else {
            have_fit = true;
            T.eval_jac(x, lb, ub, fcur, a.jac_mode, A, g);
            const int max_nfev = a.max_nfev > 0 ? a.max_nfev : 100 * N;
            double v[N], dv[N];
            // CL_scaling_vector
#pragma unroll
            for (int i = 0; i < N; ++i) {
                v[i] = 1.0; dv[i] = 0.0;
                if (g[i] < 0) { v[i] = ub[i] - x[i]; dv[i] = -1.0; }
                if (g[i] > 0) { v[i] = x[i] - lb[i]; dv[i] = 1.0; }
            }
            double Delta;
            {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) { const double q = x[i] / sqrt(v[i]); s += q * q; }
                Delta = sqrt(s);
                if (Delta == 0) Delta = 1.0;
            }
            double alpha = 0.0;
            int term = 0;      // 0 = none
            bool term_set = false;
            while (true) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    v[i] = 1.0; dv[i] = 0.0;
                    if (g[i] < 0) { v[i] = ub[i] - x[i]; dv[i] = -1.0; }
                    if (g[i] > 0) { v[i] = x[i] - lb[i]; dv[i] = 1.0; }
                }
                double g_norm = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
                if (g_norm < a.gtol) { term = 1; term_set = true; }
                if (term_set || nfev == max_nfev) break;

                double d[N], g_h[N], B[NT];
#pragma unroll
                for (int i = 0; i < N; ++i) { d[i] = sqrt(v[i]); g_h[i] = d[i] * g[i]; }
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j)
                        B[tri(i, j)] = (A[tri(i, j)] * d[i]) * d[j] + (i == j ? g[i] * dv[i] : 0.0);
                const double theta = fmax(0.995, 1 - g_norm);

                double actual_reduction = -1.0, cost_new = cost;
                double xn[N];
                while (actual_reduction <= 0 && nfev < max_nfev) {
                    double p_h[N], p[N], step[N], step_h[N], predicted;
                    solve_tr<N>(B, g_h, m, Delta, alpha, p_h);
#pragma unroll
                    for (int i = 0; i < N; ++i) p[i] = d[i] * p_h[i];
                    select_step<N>(x, B, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h, predicted);
#pragma unroll
                    for (int i = 0; i < N; ++i) xn[i] = x[i] + step[i];
                    strictly_feasible<N>(xn, lb, ub, 0.0);
                    bool finite2;
                    cost_new = T.eval_f(xn, fnew, finite2);
                    nfev += 1;
                    const double step_h_norm = normN<N>(step_h);
                    if (!finite2) {
                        Delta = 0.25 * step_h_norm;
                        continue;
                    }
                    actual_reduction = cost - cost_new;
                    // update_tr_radius
                    double ratio;
                    if (predicted > 0) ratio = actual_reduction / predicted;
                    else if (predicted == 0 && actual_reduction == 0) ratio = 1;
                    else ratio = 0;
                    double Delta_new = Delta;
                    if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
                    else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
                    const double step_norm = normN<N>(step);
                    // check_termination
                    const bool ft = (actual_reduction < a.ftol * cost) && (ratio > 0.25);
                    const bool xt = step_norm < a.xtol * (a.xtol + normN<N>(x));
                    if (ft && xt) { term = 4; term_set = true; }
                    else if (ft) { term = 2; term_set = true; }
                    else if (xt) { term = 3; term_set = true; }
                    if (term_set) break;
                    alpha *= Delta / Delta_new;
                    Delta = Delta_new;
                }
                if (actual_reduction > 0) {
#pragma unroll
                    for (int i = 0; i < N; ++i) x[i] = xn[i];
                    double *tmp = fcur; fcur = fnew; fnew = tmp;
                    cost = cost_new;
                    T.eval_jac(x, lb, ub, fcur, a.jac_mode, A, g);
                }
            }
            status = term_set ? term : 0;
        }
    }

    // ---- outputs: popt, pcov = (J^T J)^-1 * 2 cost / (m - n)  (curve_fit, _minpack_py.py:1040-1055), chi ----
    double pc[NT];
    bool cov_ok = false;
    if (have_fit && m > N) {
        double Lf[NT], inv[N], lmin2;
        cov_ok = cholN<N>(A, 0.0, Lf, inv, lmin2);
        if (cov_ok) {
            // conditioning guard equivalent to scipy's singular-value cut eps*max(m,n)*s_max
            double dmax = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) dmax = fmax(dmax, A[tri(i, i)]);
            const double thr = kEPS * (double)m;
            if (!(lmin2 > thr * thr * dmax)) cov_ok = false;
        }
        if (cov_ok) {
            const double s_sq = 2.0 * cost / (double)(m - N);
#pragma unroll
            for (int c = 0; c < N; ++c) {
                double e[N], z[N], col[N];
#pragma unroll
                for (int i = 0; i < N; ++i) e[i] = (i == c) ? 1.0 : 0.0;
                fwdN<N>(Lf, inv, e, z);
                bwdN<N>(Lf, inv, z, col);
#pragma unroll
                for (int i = c; i < N; ++i) pc[tri(i, c)] = col[i] * s_sq;
            }
        }
    }
    double chi = INFINITY;
    if (have_fit) {
        // calc_chiSq, fitting_Ct_functions.py:272-276: mean((model - y)^2 / sigma)
        double acc[1] = {0.0};
        for (int l = tid; l < a.L; l += W * 64) {
            double e[K > 0 ? K : 1];
            Model<N>::exps(x, T.t[l], e);
            const double r = Model<N>::value(x, e) - T.y[l];
            acc[0] += T.sg ? (r * r) / T.sg[l] : r * r;
        }
        T.template block_sums<1>(acc);
        chi = acc[0] / (double)a.L;
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) a.popt[(int64_t)res * N + i] = x[i];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                a.pcov[((int64_t)res * N + i) * N + j] = cov_ok ? pc[tri(i, j)] : INFINITY;
        a.chisq[res] = chi;
        a.status[res] = status;
        a.nfev[res] = nfev;
    }
}

// residual + analytic Jacobian for arbitrary parameter sets
__global__ __launch_bounds__(256) void k_resjac(const double *__restrict__ t, const double *__restrict__ y,
                                                const double *__restrict__ sigma, const double *__restrict__ params,
                                                int L, int P, double *__restrict__ resid, double *__restrict__ jac)
{
#pragma clang fp contract(off)
    const int res = blockIdx.x;
    const int K = P / 2;
    const bool freeS2 = (P % 2) == 1;
    const double *x = params + (int64_t)res * P;
    double S2;
    if (freeS2) S2 = x[P - 1];
    else {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += x[k];
        S2 = 1.0 - s;
    }
    for (int l = threadIdx.x; l < L; l += 256) {
        const int64_t o = (int64_t)res * L + l;
        const double tl = t[o], w = sigma ? 1.0 / sigma[o] : 1.0;
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double e = exp((-1.0 * tl) / x[K + k]);
            s += x[k] * e;
            if (jac) {
                jac[o * P + k] = w * (e - (freeS2 ? 0.0 : 1.0));
                jac[o * P + K + k] = w * (x[k] * e * tl / (x[K + k] * x[K + k]));
            }
        }
        if (jac && freeS2) jac[o * P + P - 1] = w;
        resid[o] = w * ((S2 + s) - y[o]);
    }
}

template <int N>
int launch_trf(sr_ctx *ctx, const FitArgs &a)
{
    // 4 waves = one per SIMD of a CU: the redundant serial algebra is VALU-issue bound, a second wave per
    // SIMD doubles its time (measured at n = 9: 68 / 41 / 28 / 51 us per iteration for W = 1 / 2 / 4 / 8)
    constexpr int W = 4;
    hipLaunchKernelGGL((k_trf<N, W>), dim3((unsigned)a.nRes), dim3(W * 64), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}

int dispatch_trf(sr_ctx *ctx, int P, const FitArgs &a)
{
    switch (P) {
        case 2: return launch_trf<2>(ctx, a);
        case 3: return launch_trf<3>(ctx, a);
        case 4: return launch_trf<4>(ctx, a);
        case 5: return launch_trf<5>(ctx, a);
        case 6: return launch_trf<6>(ctx, a);
        case 7: return launch_trf<7>(ctx, a);
        case 8: return launch_trf<8>(ctx, a);
        case 9: return launch_trf<9>(ctx, a);
        case 10: return launch_trf<10>(ctx, a);
        case 11: return launch_trf<11>(ctx, a);
        default:
            sr_set_error("sr_expfit_lm_f64: P=%d parameters not supported (2..%d)", P, kNmax);
            return -3;
    }
}

}  // namespace

extern "C" {

int sr_expfit_lm_f64_dev(sr_ctx *ctx, const double *t, const double *C, const double *sigma, int nRes, int L, int P,
                         const double *p0, double tau_max, int max_iter, const unsigned char *skip, double *work,
                         double *popt, double *pcov, double *chisq, int *status, int *n_iter)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && p0 && popt && pcov && chisq && status && n_iter, -2, "sr_expfit_lm_f64_dev: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && P >= 2 && P <= kNmax, -3, "sr_expfit_lm_f64_dev: bad sizes nRes=%d L=%d P=%d", nRes, L, P);
    double *fws = work ? work : (double *)sr_workspace(ctx, SR_WS_FIT, (size_t)nRes * 2 * L * sizeof(double));
    if (!fws) return -5;
    FitArgs a;
    a.t = t; a.y = C; a.sigma = sigma; a.p0 = p0; a.skip = skip; a.tau_max = tau_max;
    a.nRes = nRes; a.L = L; a.max_nfev = max_iter; a.jac_mode = 0;
    a.ftol = a.xtol = a.gtol = 1e-8;
    a.popt = popt; a.pcov = pcov; a.chisq = chisq; a.status = status; a.nfev = n_iter; a.fws = fws;
    if (max_iter < 0) { a.max_nfev = -max_iter; a.jac_mode = 1; }     // negative: analytic Jacobian variant
    return dispatch_trf(ctx, P, a);
}

int sr_expfit_lm_f64(sr_ctx *ctx, const double *t, const double *C, const double *sigma, int nRes, int L, int P,
                     const double *p0, double tau_max, int max_iter, double *popt, double *pcov, double *chisq,
                     int *status, int *n_iter)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && p0 && popt && pcov && chisq && status && n_iter, -2, "sr_expfit_lm_f64: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && P >= 2 && P <= kNmax, -3, "sr_expfit_lm_f64: bad sizes nRes=%d L=%d P=%d", nRes, L, P);
    const size_t nL = (size_t)nRes * L, nP = (size_t)nRes * P;
    double *in = (double *)sr_workspace(ctx, SR_WS_IN0, (3 * nL + nP) * sizeof(double));
    double *out = (double *)sr_workspace(ctx, SR_WS_OUT0, (nP + nP * P + nRes) * sizeof(double));
    int *iout = (int *)sr_workspace(ctx, SR_WS_OUT1, (size_t)nRes * 2 * sizeof(int));
    if (!in || !out || !iout) return -5;
    double *t_d = in, *y_d = in + nL, *s_d = in + 2 * nL, *p0_d = in + 3 * nL;
    SR_HIP(hipMemcpyAsync(t_d, t, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(y_d, C, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (sigma) SR_HIP(hipMemcpyAsync(s_d, sigma, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(p0_d, p0, nP * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    double *popt_d = out, *pcov_d = out + nP, *chi_d = pcov_d + nP * P;
    int rc = sr_expfit_lm_f64_dev(ctx, t_d, y_d, sigma ? s_d : nullptr, nRes, L, P, p0_d, tau_max, max_iter, nullptr, nullptr, popt_d, pcov_d,
                                  chi_d, iout, iout + nRes);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(popt, popt_d, nP * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(pcov, pcov_d, nP * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(chisq, chi_d, (size_t)nRes * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(status, iout, (size_t)nRes * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(n_iter, iout + nRes, (size_t)nRes * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_expfit_resjac_f64(sr_ctx *ctx, const double *t, const double *C, const double *sigma, const double *params,
                         int nRes, int L, int P, double *resid, double *jac)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && params && resid, -2, "sr_expfit_resjac_f64: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && P >= 1 && P <= 64, -3, "sr_expfit_resjac_f64: bad sizes");
    const size_t nL = (size_t)nRes * L, nP = (size_t)nRes * P;
    double *in = (double *)sr_workspace(ctx, SR_WS_IN0, (3 * nL + nP) * sizeof(double));
    double *r_d = (double *)sr_workspace(ctx, SR_WS_OUT0, nL * sizeof(double));
    double *j_d = jac ? (double *)sr_workspace(ctx, SR_WS_OUT1, nL * P * sizeof(double)) : nullptr;
    if (!in || !r_d || (jac && !j_d)) return -5;
    double *t_d = in, *y_d = in + nL, *s_d = in + 2 * nL, *p_d = in + 3 * nL;
    SR_HIP(hipMemcpyAsync(t_d, t, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(y_d, C, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (sigma) SR_HIP(hipMemcpyAsync(s_d, sigma, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(p_d, params, nP * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_resjac, dim3((unsigned)nRes), dim3(256), 0, ctx->stream, t_d, y_d, sigma ? s_d : nullptr, p_d, L, P,
                       r_d, j_d);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(resid, r_d, nL * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (jac) SR_HIP(hipMemcpyAsync(jac, j_d, nL * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
