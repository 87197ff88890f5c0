// Developer probe: can a kernel release a stream that waits with hipStreamWaitValue32 on signal memory?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void writer(uint32_t *flag, uint32_t v, long long spin, long long *t_out)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) {
        __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        t_out[0] = wall_clock64();
    }
    t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) t_out[1] = wall_clock64();
}
__global__ void reader(long long *t_out) { if (threadIdx.x == 0) t_out[2] = wall_clock64(); }
int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint32_t *flag = nullptr;
    CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
    *flag = 0;
    long long *t = nullptr;
    CK(hipHostMalloc((void **)&t, 4 * sizeof(long long)));
    hipStream_t a, b;
    CK(hipStreamCreate(&a));
    CK(hipStreamCreate(&b));
    for (int rep = 1; rep <= 3; ++rep) {
        t[0] = t[1] = t[2] = 0;
        CK(hipStreamWaitValue32(b, flag, (uint32_t)rep, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(reader, dim3(1), dim3(64), 0, b, t);
        hipLaunchKernelGGL(writer, dim3(4), dim3(64), 0, a, flag, (uint32_t)rep, 100000LL * 10, t);   // wall clock 100 MHz: 10 ms halves
        CK(hipStreamWriteValue32(a, flag, (uint32_t)rep, 0));                                           // fallback release
        CK(hipDeviceSynchronize());
        printf("rep %d: flag written at 0, writer ended at %+.3f ms, reader ran at %+.3f ms\n", rep, (t[1] - t[0]) / 1e5, (t[2] - t[0]) / 1e5);
    }
    return 0;
}
