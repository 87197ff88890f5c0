// Developer probe: what a SIMD of gfx950 issues per cycle in float32 -- plain (v_fma_f32, v_add_f32) against packed
// (v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32) instructions, 16 independent chains per wave, 1 .. 4 waves per SIMD, every SIMD of
// the chip busy.  Reports shader cycles (s_memtime) per wave-instruction and wave, the same per SIMD, the clock the chip held
// (s_memtime against s_memrealtime, 100 MHz) and the resulting float operations per cycle and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float seed, int iters, long long *cyc, long long *rt, float *sink)
{
    v2f a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (v2f){seed + i, seed - i};
    const v2f b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 6) {      // the mix of a butterfly: 2 plain adds + 1 plain fma per ...
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.x), "v"(c.x));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    const int iters = 20000;
    const char *names[7] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32 op_sel", "add+fma pair"};
    const double flop[7] = {2, 1, 4, 2, 2, 4, 3}, ins[7] = {1, 1, 1, 1, 1, 1, 2};
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps, threads = 256;          // 4 waves per workgroup = one per SIMD; wps workgroups per CU
        long long *cyc, *rt; float *sink;
        CK(hipMalloc((void **)&cyc, blocks * sizeof(long long)));
        CK(hipMalloc((void **)&rt, blocks * sizeof(long long)));
        CK(hipMalloc((void **)&sink, (size_t)blocks * threads * sizeof(float)));
        for (int op = 0; op < 7; ++op) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
#define L(N) if (op == N) hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(threads), 0, 0, 1.0f, iters, cyc, rt, sink);
                L(0) L(1) L(2) L(3) L(4) L(5) L(6)
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(blocks), hr(blocks);
            CK(hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), rt, blocks * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end()); std::sort(hr.begin(), hr.end());
            const double n = (double)iters * 16.0 * ins[op];
            const double per = (double)h[blocks / 2] / n;
            const double ghz = (double)h[blocks / 2] / ((double)hr[blocks / 2] * 10.0);     // realtime ticks are 10 ns
            printf("%d waves/SIMD  %-20s %.2f cycles per instruction and wave, %.2f per SIMD; clock %.2f GHz; %.1f flop/cycle/SIMD; kernel %.3f ms -> %.1f TFLOP/s\n",
                   wps, names[op], per, per / wps, ghz, 64.0 * flop[op] / ins[op] / (per / wps), ms,
                   1024.0 * wps * n / ins[op] * 64.0 * flop[op] / (ms * 1e-3) * 1e-12);
        }
        CK(hipFree(cyc)); CK(hipFree(rt)); CK(hipFree(sink));
    }
    return 0;
}
