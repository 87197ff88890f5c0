/*
 * ct_palmer_oracle.c -- plain-C restatement of the C(t) hot loop.  TEST INFRASTRUCTURE ONLY:
 * built into oracle/libsr_oracle.so and loaded by tests/ and bench.py's cpu_baseline leg; the
 * product (spinrelax_amd/) never links or loads it.
 *
 * Restates calculate_Ct_Palmer, /root/reference/calculate-Ct-from-traj.py:200-238:
 *   for delta in 1..F/2:
 *       tmp[r,j,v] = -0.5 + 1.5*(u[r,j,v,:].u[r,j+delta,v,:])^2          (:225)
 *       p[r,v]     = sum_j tmp[r,j,v] / (F-delta)                         (:226)
 *       Ct[delta-1,v]  = mean_r p[r,v]                                    (:227)
 *       dCt[delta-1,v] = std_r(p[r,v], ddof=0) / (sqrt(R) - 1)            (:228)
 * evaluated in float64 on the float32 input (the parity definition of SURVEY.md section 8(c)).
 * Parity of this file is pinned in tests/test_oracle_golden.py against tests/golden/ct_*.npz,
 * whose expected outputs were produced by the imported reference function itself.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* vecs: [R][F][V][3] float32, C-contiguous.  Ct, dCt: [L][V] float64.  p (optional): [R][L][V]. */
int sr_oracle_ct_palmer_f64(const float *vecs, int64_t R, int64_t F, int64_t V,
                            double *Ct, double *dCt, double *p_out)
{
    const int64_t L = F / 2;
    if (R < 1 || F < 2 || V < 1) return -1;
    double *p = p_out ? p_out : (double *)malloc(sizeof(double) * (size_t)(R * L * V));
    if (!p) return -2;

#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int64_t r = 0; r < R; ++r) {
        for (int64_t v = 0; v < V; ++v) {
            const float *u = vecs + (r * F * V + v) * 3;
            const int64_t st = V * 3;
            for (int64_t d = 1; d <= L; ++d) {
                double acc = 0.0;
                for (int64_t j = 0; j + d < F; ++j) {
                    const float *a = u + j * st, *b = u + (j + d) * st;
                    double x = (double)a[0] * (double)b[0] + (double)a[1] * (double)b[1]
                             + (double)a[2] * (double)b[2];
                    acc += -0.5 + 1.5 * (x * x);
                }
                p[(r * L + (d - 1)) * V + v] = acc / (double)(F - d);
            }
        }
    }

    const double denom = sqrt((double)R) - 1.0;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < L * V; ++k) {
        double m = 0.0;
        for (int64_t r = 0; r < R; ++r) m += p[r * L * V + k];
        m /= (double)R;
        double s = 0.0;
        for (int64_t r = 0; r < R; ++r) { double e = p[r * L * V + k] - m; s += e * e; }
        Ct[k] = m;
        dCt[k] = sqrt(s / (double)R) / denom;     /* R == 1 -> 0/0 = NaN, as numpy gives */
    }
    if (!p_out) free(p);
    return 0;
}

/* The reference's own arithmetic type: float32 throughout, one pass per lag over the whole
 * (R, F-delta, V) block exactly like the numpy einsum formulation (streams both operand views
 * from memory once per lag).  Used only as a timed CPU baseline ("port"), single- or multi-thread. */
int sr_oracle_ct_palmer_f32_stream(const float *vecs, int64_t R, int64_t F, int64_t V,
                                   float *Ct, float *dCt)
{
    const int64_t L = F / 2;
    if (R < 1 || F < 2 || V < 1) return -1;
    const float denom = sqrtf((float)R) - 1.0f;
#pragma omp parallel
    {
        float *p = (float *)malloc(sizeof(float) * (size_t)(R * V));
#pragma omp for schedule(dynamic, 1)
        for (int64_t d = 1; d <= L; ++d) {
            for (int64_t r = 0; r < R; ++r) {
                float *pr = p + r * V;
                for (int64_t v = 0; v < V; ++v) pr[v] = 0.0f;
                for (int64_t j = 0; j + d < F; ++j) {
                    const float *a = vecs + ((r * F + j) * V) * 3;
                    const float *b = vecs + ((r * F + j + d) * V) * 3;
                    for (int64_t v = 0; v < V; ++v) {
                        float x = a[3 * v] * b[3 * v] + a[3 * v + 1] * b[3 * v + 1] + a[3 * v + 2] * b[3 * v + 2];
                        pr[v] += -0.5f + 1.5f * (x * x);
                    }
                }
                for (int64_t v = 0; v < V; ++v) pr[v] /= (float)(F - d);
            }
            for (int64_t v = 0; v < V; ++v) {
                float m = 0.0f;
                for (int64_t r = 0; r < R; ++r) m += p[r * V + v];
                m /= (float)R;
                float s = 0.0f;
                for (int64_t r = 0; r < R; ++r) { float e = p[r * V + v] - m; s += e * e; }
                Ct[(d - 1) * V + v] = m;
                dCt[(d - 1) * V + v] = sqrtf(s / (float)R) / denom;
            }
        }
        free(p);
    }
    return 0;
}
