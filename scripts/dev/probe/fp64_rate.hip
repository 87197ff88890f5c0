// Developer probe: issue cost of v_add_f64 / v_mul_f64 / v_fma_f64 against v_fma_f32 on gfx950, in shader cycles per wave-instruction
// (s_memtime stamps around a loop of independent instructions), with one and with two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_rate fp64_rate.hip && ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ void k(double seed, int iters, long long *cyc, double *sink)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const double b = 1.0000001, c = 1e-9;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const float fb = 1.0000001f, fc = 1e-9f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define EIGHT(INS, A, B, C) \
        asm volatile(INS : "+v"(A##0) : "v"(B), "v"(C)); asm volatile(INS : "+v"(A##1) : "v"(B), "v"(C)); \
        asm volatile(INS : "+v"(A##2) : "v"(B), "v"(C)); asm volatile(INS : "+v"(A##3) : "v"(B), "v"(C)); \
        asm volatile(INS : "+v"(A##4) : "v"(B), "v"(C)); asm volatile(INS : "+v"(A##5) : "v"(B), "v"(C)); \
        asm volatile(INS : "+v"(A##6) : "v"(B), "v"(C)); asm volatile(INS : "+v"(A##7) : "v"(B), "v"(C));
        if (OP == 0) { EIGHT("v_add_f64 %0, %0, %2", a, b, c) EIGHT("v_add_f64 %0, %0, %2", a, b, c) }
        if (OP == 1) { EIGHT("v_mul_f64 %0, %0, %1", a, b, c) EIGHT("v_mul_f64 %0, %0, %1", a, b, c) }
        if (OP == 2) { EIGHT("v_fma_f64 %0, %0, %1, %2", a, b, c) EIGHT("v_fma_f64 %0, %0, %1, %2", a, b, c) }
        if (OP == 3) { EIGHT("v_fma_f32 %0, %0, %1, %2", f, fb, fc) EIGHT("v_fma_f32 %0, %0, %1, %2", f, fb, fc) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

int main()
{
    const int iters = 20000;
    const char *names[4] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_fma_f32"};
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * wps, threads = 256;          // 4 waves per workgroup = one per SIMD; wps workgroups per CU
        long long *cyc; double *sink;
        CK(hipMalloc((void **)&cyc, blocks * sizeof(long long)));
        CK(hipMalloc((void **)&sink, (size_t)blocks * threads * sizeof(double)));
        for (int op = 0; op < 4; ++op) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, 1.0, iters, cyc, sink);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, 1.0, iters, cyc, sink);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, 1.0, iters, cyc, sink);
                if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, 1.0, iters, cyc, sink);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(blocks);
            CK(hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double per = (double)h[blocks / 2] / ((double)iters * 16.0);
            printf("%d wave(s) per SIMD  %-10s  %.2f shader cycles per wave-instruction and wave (median workgroup), %.2f per SIMD issue slot; kernel %.3f ms\n",
                   wps, names[op], per, per / wps, ms);
        }
        CK(hipFree(cyc)); CK(hipFree(sink));
    }
    return 0;
}
