// Micro-benchmark: cost of dependent v_mfma_f64_16x16x4 chains with one wave per SIMD (k_potrf_reg's regime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain scripts/mfma_f64_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double *out, unsigned long long *cyc, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 - a;
    d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 4 dependent MFMAs, VGPR accumulators
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t"
                         "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                         : "+v"(c0) : "v"(a), "v"(b));
        } else if (MODE == 1) {   // 4 dependent MFMAs, AGPR accumulators
            asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\t"
                         "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]"
                         :: "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        } else if (MODE == 2) {   // 8 MFMAs, two interleaved chains, AGPR
            asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n\t"
                         "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n\t"
                         "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n\t"
                         "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]"
                         :: "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12",
                            "a13", "a14", "a15");
        } else if (MODE == 3) {   // 4 independent MFMAs (4 tiles), AGPR
            asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n\t"
                         "v_mfma_f64_16x16x4_f64 a[16:23], %0, %1, a[16:23]\n\tv_mfma_f64_16x16x4_f64 a[24:31], %0, %1, a[24:31]"
                         :: "v"(a), "v"(b) : "a0", "a8", "a16", "a24", "a31");
        } else if (MODE == 4) {   // 136 dependent DPP FMAs ~ one forward substitution
            double x = a;
#pragma unroll
            for (int r = 0; r < 34; ++r)
                asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf"
                             : "+v"(x) : "v"(b), "v"(a));
            c1[0] += x;
        } else if (MODE == 5) {   // 136 independent-ish DPP FMAs over 16 accumulators
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = a + r;
#pragma unroll
            for (int r = 0; r < 136; ++r)
                asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[r & 15]) : "v"(b), "v"(a));
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[0] += x[r];
        } else if (MODE == 6) {   // 136 plain FMAs over 16 accumulators
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = a + r;
#pragma unroll
            for (int r = 0; r < 136; ++r)
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[r & 15]) : "v"(b), "v"(a));
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[0] += x[r];
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[0];
}

int main() {
    double *out; unsigned long long *cyc, h[8] = {0};
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 64);
    const int iters = 1000;
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char *names[] = {"4 dependent MFMA (VGPR acc)", "4 dependent MFMA (AGPR acc)", "2x4 interleaved MFMA (AGPR)",
                           "4 independent MFMA (AGPR)", "136 dependent DPP fmac", "136 DPP fmac over 16 acc",
                           "136 plain fmac over 16 acc"};
    for (int m = 0; m < 7; ++m) printf("%-32s %8.1f s_memtime ticks / iteration\n", names[m], (double)h[m] / iters);
    return 0;
}
