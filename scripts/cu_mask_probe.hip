// Does a CU-masked stream confine a kernel's workgroups (hipExtStreamCreateWithCUMask)?  Prints the distinct (XCC, SE, CU) a
// 512-workgroup launch ran on, for an unmasked stream and for masks of 8 / 248 CUs, and the time of a busy kernel on each.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe scripts/cu_mask_probe.hip && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff) | (x < 0 ? 1u << 31 : 0);
}
static void run(const char *name, hipStream_t s, unsigned *d, int n) {
    std::vector<unsigned> h(n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, s, d, 2000);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, s, d, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    for (unsigned v : h) cus.insert(((v >> 16) & 0xf) << 12 | ((v >> 8) & 0xf) << 4 | ((v >> 13) & 0x7) << 8);   // xcc | cu_id(bits 8-11) | se_id(bits 13-15)
    printf("%-22s distinct (xcc, se, cu): %3zu   busy kernel %.3f ms\n", name, cus.size(), ms);
}
int main() {
    const int n = 2048;
    unsigned *d; hipMalloc(&d, n * 4);
    hipStream_t s0, s8, s248;
    hipStreamCreate(&s0);
    uint32_t m8[8] = {0}, m248[8];
    for (int i = 0; i < 8; ++i) { m8[i] = 1u; m248[i] = ~1u; }      // bit 0 of every 32-bit word / everything else
    hipError_t r1 = hipExtStreamCreateWithCUMask(&s8, 8, m8), r2 = hipExtStreamCreateWithCUMask(&s248, 8, m248);
    printf("create masked streams: %s / %s\n", hipGetErrorString(r1), hipGetErrorString(r2));
    run("unmasked", s0, d, n);
    if (r1 == hipSuccess) run("mask: 8 bits", s8, d, n);
    if (r2 == hipSuccess) run("mask: 248 bits", s248, d, n);
    return 0;
}
