// Can small, latency-critical launches on a HIGH-priority stream run under a chip-filling GEMM-like kernel of another stream?
// A: 8192 workgroups of 256 threads, 36 KB LDS, ~190 VGPRs (two per CU), ~50 us each.  B: 40 one-workgroup kernels (90 KB LDS,
// ~5 us) back to back.  Prints B's total time alone, under A on a default-priority stream, under A on a high-priority stream.
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/prio_probe scripts/prio_stream_probe.hip && /tmp/prio_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void kA(double *out, int spin) {
    __shared__ double sm[4608];
    double a[80];
    for (int i = 0; i < 80; ++i) a[i] = threadIdx.x + i;
    for (int it = 0; it < spin; ++it)
#pragma unroll
        for (int i = 0; i < 80; ++i) a[i] = a[i] * 1.0000001 + 1e-9;
    double s = 0;
    for (int i = 0; i < 80; ++i) s += a[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[1] + sm[2];
}
__global__ __launch_bounds__(256) void kB(double *out, int spin) {
    extern __shared__ double dyn[];
    double x = threadIdx.x;
    for (int it = 0; it < spin; ++it) x = x * 1.0000001 + 1e-9;
    dyn[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = dyn[1];
}
static float runB(hipStream_t sb, double *d, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, sb);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kB, dim3(1), dim3(256), 90 * 1024, sb, d, 600);
    hipEventRecord(e1, sb);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double *d; hipMalloc(&d, 65536 * 8 + 64);
    hipFuncSetAttribute((const void *)kB, hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t sa, sb0, sbh;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb0, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&sbh, hipStreamNonBlocking, hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    // calibrate A
    hipEvent_t a0, a1; hipEventCreate(&a0); hipEventCreate(&a1);
    hipLaunchKernelGGL(kA, dim3(512), dim3(256), 0, sa, d + 8, 150);
    hipEventRecord(a0, sa);
    hipLaunchKernelGGL(kA, dim3(65536), dim3(256), 0, sa, d + 8, 150);
    hipEventRecord(a1, sa);
    hipEventSynchronize(a1);
    float msA; hipEventElapsedTime(&msA, a0, a1);
    printf("A alone (65536 workgroups): %.3f ms = %.1f us per round of 512\n", msA, msA * 1e3 / 128);
    runB(sb0, d, 4);
    printf("B alone (40 kernels): %.3f ms\n", runB(sb0, d, 40));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kA, dim3(65536), dim3(256), 0, sa, d + 8, 150);
        float t0 = runB(sb0, d, 40);
        hipStreamSynchronize(sa);
        hipLaunchKernelGGL(kA, dim3(65536), dim3(256), 0, sa, d + 8, 150);
        float t1 = runB(sbh, d, 40);
        hipStreamSynchronize(sa);
        printf("B under A: default priority %.3f ms, high priority %.3f ms\n", t0, t1);
    }
    return 0;
}
