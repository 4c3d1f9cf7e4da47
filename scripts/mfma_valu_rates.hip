// Issue rates behind DESIGN.md section 4 ("what bounds the GEMM-shaped stages"), measured with s_memtime (shader cycles):
//  (1) v_mfma_f64_16x16x4_f64 from 1..4 waves per SIMD, 1 / 2 / 8 independent accumulators: cycles per MFMA per SIMD
//  (2) v_fma_f64 from 1..4 waves per SIMD: cycles per instruction per SIMD
//  (3) one wave of back-to-back MFMAs next to one wave of v_fma_f64 (or v_fma_f32) on every SIMD: do they overlap?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_rates scripts/mfma_valu_rates.hip && ./mfma_valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k_mfma(double *out, unsigned long long *st, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(&st[0], m1 - m0);      // the slowest wave: the pipe serves the oldest wave first
}

__global__ __launch_bounds__(1024) void k_fma(double *out, unsigned long long *st, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(&st[0], m1 - m0);
}

// mode bit0: waves 0-3 run MFMA f64; bit1: waves 4-7 run v_fma_f64; bit2: waves 4-7 run v_fma_f32 instead
__global__ __launch_bounds__(512) void k_mix(double *out, unsigned long long *st, int iters, double a0, double b0, int mode) {
    int wave = threadIdx.x >> 6;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    unsigned long long m0 = __builtin_amdgcn_s_memtime();
    double s = 0;
    if (wave < 4) {
        if (mode & 1) {
            d4 acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (mode & 2) {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    } else if (mode & 4) {
        float x[8], fa = (float)a, fb = (float)b;
        for (int i = 0; i < 8; ++i) x[i] = fa + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], fb, fa);
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    unsigned long long m1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(&st[wave < 4 ? 0 : 1], m1 - m0);
}

int main() {
    int iters = 2000;
    double *d; unsigned long long *st, h[2];
    hipMalloc(&d, sizeof(double) * 256 * 1024);
    hipMalloc(&st, sizeof(h));
    auto run = [&](auto launch) {
        for (int rep = 0; rep < 2; ++rep) { hipMemset(st, 0, sizeof(h)); launch(); hipDeviceSynchronize(); }
        hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    };
    for (int nacc : {1, 2, 8})
        for (int wps = 1; wps <= 4; ++wps) {
            run([&] {
                if (nacc == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(256 * wps), 0, 0, d, st, iters * 8, 1.000001, 0.999999);
                if (nacc == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(256 * wps), 0, 0, d, st, iters * 4, 1.000001, 0.999999);
                if (nacc == 8) hipLaunchKernelGGL(k_mfma<8>, dim3(256), dim3(256 * wps), 0, 0, d, st, iters, 1.000001, 0.999999);
            });
            printf("v_mfma_f64_16x16x4, %d independent accumulator(s), %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD\n", nacc, wps,
                   (double)h[0] / (iters * 8.0) / wps);
        }
    for (int wps = 1; wps <= 4; ++wps) {
        run([&] { hipLaunchKernelGGL(k_fma, dim3(256), dim3(256 * wps), 0, 0, d, st, iters, 1.000001, 0.999999); });
        printf("v_fma_f64, %d wave(s)/SIMD: %.2f cycles per instruction per SIMD\n", wps, (double)h[0] / (iters * 64.0) / wps);
    }
    const char *names[] = {"", "MFMA wave alone", "v_fma_f64 wave alone", "MFMA wave + v_fma_f64 wave", "v_fma_f32 wave alone", "MFMA wave + v_fma_f32 wave"};
    for (int mode : {1, 2, 3, 4, 5}) {
        run([&] { hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, d, st, iters, 1.000001, 0.999999, mode); });
        printf("%-28s", names[mode]);
        if (mode & 1) printf("  %.1f cycles per MFMA", (double)h[0] / (iters * 4.0));
        if (mode & 6) printf("  %.2f cycles per fma instruction", (double)h[1] / (iters * 64.0));
        printf("\n");
    }
    return 0;
}
