// FP64 MFMA issue-rate microbenchmark (v_mfma_f64_16x16x4_f64): back-to-back MFMAs on independent
// accumulators, every SIMD of every CU busy.  Prints sustained TFLOP/s; used to pin the "peak" of bench.py's
// roofline (vendor sheet: 78.6 TFLOP/s FP64 matrix on MI355X).   hipcc --offload-arch=gfx950 -O3 -o mfma_peak ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int blocks = 256 * 4, threads = 256, iters = 20000;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.000001, 0.999999);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * (threads / 64) * iters * 8.0 * (16.0 * 16 * 4 * 2);
        printf("rep %d: %.3f ms  %.2f TFLOP/s FP64 MFMA\n", rep, ms, flops / ms * 1e-9);
    }
    return 0;
}
