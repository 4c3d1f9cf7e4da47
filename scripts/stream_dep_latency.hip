// Cross-queue dependency latency on one GPU: kernel A on stream s, kernel B on stream s2 must start after A.
// Variants: (0) same stream, (1) hipEventRecord + hipStreamWaitEvent, (2) hipStreamWriteValue32 + hipStreamWaitValue32.
// Kernels stamp s_memrealtime (100 MHz) at their end (A) / start (B).  hipcc --offload-arch=gfx950 -O2 -o /tmp/sdl scripts/stream_dep_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void kA(unsigned long long *t, int spin) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
}
__global__ void kB(unsigned long long *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t s, s2;
    CK(hipStreamCreate(&s)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned long long *t; CK(hipHostMalloc((void **)&t, 64, hipHostMallocMapped | hipHostMallocCoherent));
    uint32_t *flag; CK(hipMalloc((void **)&flag, 64)); CK(hipMemset(flag, 0, 64));
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue = %d\n", can);
    for (int variant = 0; variant < 3; ++variant) {
        if (variant == 2 && !can) continue;
        std::vector<double> gaps;
        for (int it = 0; it < 60; ++it) {
            t[0] = t[1] = 0;
            hipLaunchKernelGGL(kA, dim3(64), dim3(256), 0, s, t, 2000);      // ~20 us
            if (variant == 0) {
                hipLaunchKernelGGL(kB, dim3(64), dim3(256), 0, s, t);
            } else if (variant == 1) {
                CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(s2, ev, 0));
                hipLaunchKernelGGL(kB, dim3(64), dim3(256), 0, s2, t);
            } else {
                CK(hipStreamWriteValue32(s, flag, it + 1, 0));
                CK(hipStreamWaitValue32(s2, flag, it + 1, hipStreamWaitValueEq, 0xffffffff));
                hipLaunchKernelGGL(kB, dim3(64), dim3(256), 0, s2, t);
            }
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            if (it >= 10) gaps.push_back((double)(long long)(t[1] - t[0]) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        printf("variant %d: A end -> B start  median %.2f us  min %.2f  max %.2f\n", variant, gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
    return 0;
}
