// Developer probe (round 4): what would an EIGHT-wave transform cost?  3 072-point complex float64 FFT per workgroup of 512 threads,
// 6 x 8 x 8 x 8 (thread-held 6-point step, then three radix-8 steps in place in LDS on 384 of the 512 threads), 7 transforms per
// series with a |X|^2 accumulation standing in for the spectrum step, 12 288 series -- the shape of k_ct_rfft's cfg3 launch.
// Synthetic input, no global traffic in the loop: this measures the transform structure only (instruction stream per wave,
// LDS exchanges, barriers, four waves per SIMD at <= 128 VGPRs).   hipcc --offload-arch=gfx950 -O3 -o fft8_probe fft8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct alignas(16) cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx mul_mi(cplx a) { return {a.im, -a.re}; }                                   // * (-i)
// 8-point DFT in registers (decimation in frequency, outputs in natural order)
__device__ __forceinline__ void fft8(cplx *v)
{
    constexpr double s = 0.70710678118654752440;
    cplx a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    cplx b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
    b1 = {s * (b1.re + b1.im), s * (b1.im - b1.re)};          // * w8^1
    b2 = mul_mi(b2);                                            // * w8^2
    b3 = {s * (b3.im - b3.re), -s * (b3.re + b3.im)};         // * w8^3
    cplx c0 = cadd(a0, a2), c1 = cadd(a1, a3), c2 = csub(a0, a2), c3 = mul_mi(csub(a1, a3));
    cplx d0 = cadd(b0, b2), d1 = cadd(b1, b3), d2 = csub(b0, b2), d3 = mul_mi(csub(b1, b3));
    v[0] = cadd(c0, c1); v[4] = csub(c0, c1); v[2] = cadd(c2, c3); v[6] = csub(c2, c3);
    v[1] = cadd(d0, d1); v[5] = csub(d0, d1); v[3] = cadd(d2, d3); v[7] = csub(d2, d3);
}
__device__ __forceinline__ void dft3(cplx x0, cplx x1, cplx x2, cplx &y0, cplx &y1, cplx &y2)
{
    constexpr double h = 0.86602540378443864676;
    const cplx t = cadd(x1, x2), d = csub(x1, x2);
    const cplx m = {x0.re - 0.5 * t.re, x0.im - 0.5 * t.im};
    const cplx r = {h * d.im, -h * d.re};
    y0 = cadd(x0, t); y1 = cadd(m, r); y2 = csub(m, r);
}
constexpr int P = 576;                      // row pitch of the image (slots of 16 B): element a + 8 b + 64 c at slot a + 9 b + 72 c
__device__ __forceinline__ int slot(int n2) { return (n2 & 7) + 9 * ((n2 >> 3) & 7) + 72 * (n2 >> 6); }
constexpr int TW64 = 6 * P, TW512 = TW64 + 64, LDS_SLOTS = TW512 + 512;

__global__ __launch_bounds__(512, 2) void k_probe(double *out, int ntrans, double eps)
{
    extern __shared__ cplx lds[];
    const int tid = threadIdx.x;
    {
        double sn, cs;
        sincospi(-2.0 * ((tid >> 3) + 8 * (tid & 7)) / 512.0, &sn, &cs);          // w_512^(kc + 8 kb) at [kb + 8 kc]: lane-contiguous in step 4
        lds[TW512 + tid] = {cs, sn};
        if (tid < 64) { sincospi(-2.0 * tid / 64.0, &sn, &cs); lds[TW64 + tid] = {cs, sn}; }
    }
    cplx base;
    { double sn, cs; sincospi(-2.0 * tid / 3072.0, &sn, &cs); base = {cs, sn}; }
    cplx x[6];
#pragma unroll
    for (int n1 = 0; n1 < 6; ++n1) x[n1] = {1e-3 * (tid + 512 * n1 + blockIdx.x % 7), 0.25 + 1e-4 * n1};
    double acc = 0.0;
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < ntrans; ++c) {
        // step 1: 6-point transform over n1 (6 = 2 x 3), twiddle w_3072^(tid k1), image[k1][tid]
        cplx e0, e1, e2, o0, o1, o2;
        dft3(x[0], x[2], x[4], e0, e1, e2);
        dft3(x[1], x[3], x[5], o0, o1, o2);
        constexpr double h = 0.86602540378443864676;
        o1 = cmul(o1, cplx{0.5, -h});                          // w6^1
        o2 = cmul(o2, cplx{-0.5, -h});                         // w6^2
        cplx y[6] = {cadd(e0, o0), cadd(e1, o1), cadd(e2, o2), csub(e0, o0), csub(e1, o1), csub(e2, o2)};
        cplx pw = base;
#pragma unroll
        for (int k1 = 1; k1 < 6; ++k1) { y[k1] = cmul(y[k1], pw); if (k1 < 5) pw = cmul(pw, base); }
#pragma unroll
        for (int k1 = 0; k1 < 6; ++k1) lds[k1 * P + slot(tid)] = y[k1];
        __syncthreads();
        const int k1 = tid >> 6, j = tid & 63;
        const bool act = tid < 384;
        cplx v[8];
        if (act) {                                              // step 2: over c, element a + 8 b + 64 c, (a, b) = j
            cplx *b = lds + k1 * P + (j & 7) + 9 * (j >> 3);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = b[72 * q];
            fft8(v);
#pragma unroll
            for (int q = 0; q < 8; ++q) b[72 * q] = v[q];
        }
        __syncthreads();
        if (act) {                                              // step 3: over b, (a, kc) = (j & 7, j >> 3), twiddle w_64^(b kc)
            const int a = j & 7, kc = j >> 3;
            cplx *b = lds + k1 * P + a + 72 * kc;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = b[9 * q];
            const cplx w = lds[TW64 + kc];
            cplx pw = w;
#pragma unroll
            for (int q = 1; q < 8; ++q) { v[q] = cmul(v[q], pw); if (q < 7) pw = cmul(pw, w); }
            fft8(v);
#pragma unroll
            for (int q = 0; q < 8; ++q) b[9 * q] = v[q];
        }
        __syncthreads();
        if (act) {                                              // step 4: over a, (kb, kc) = (j & 7, j >> 3), twiddle w_512^(a (kc + 8 kb))
            const int kb = j & 7, kc = j >> 3;
            const cplx *b = lds + k1 * P + 9 * kb + 72 * kc;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = b[q];
            const cplx w = lds[TW512 + j];
            cplx pw = w;
#pragma unroll
            for (int q = 1; q < 8; ++q) { v[q] = cmul(v[q], pw); if (q < 7) pw = cmul(pw, w); }
            fft8(v);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fma(v[q].re, v[q].re, fma(v[q].im, v[q].im, acc));
        }
        __syncthreads();
#pragma unroll
        for (int n1 = 0; n1 < 6; ++n1) x[n1].re += eps * acc;
    }
    out[(size_t)blockIdx.x * 512 + tid] = acc;
}

int main()
{
    const int series = 12288, ntrans = 7;
    double *out;
    CK(hipMalloc((void **)&out, (size_t)series * 512 * sizeof(double)));
    const size_t lds = (size_t)LDS_SLOTS * sizeof(cplx);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe, dim3(series), dim3(512), lds, 0, out, ntrans, 1e-30);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eight-wave transform probe: %d series x %d transforms of 3072 points: %.4f ms (LDS %zu B per workgroup)\n", series, ntrans, ms, lds);
    }
    // correctness of one transform against a direct DFT (first series, ntrans = 1 would need the spectrum; here: checksum only)
    std::vector<double> h(512);
    CK(hipMemcpy(h.data(), out, 512 * sizeof(double), hipMemcpyDeviceToHost));
    double s = 0; for (double v : h) s += v;
    printf("checksum %.6e\n", s);
    return 0;
}
