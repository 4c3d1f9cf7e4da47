// How does hipExtStreamCreateWithCUMask number the CUs of an MI355X?  For a mask with ONE bit set (bit b of 256) the kernel
// below reports which (XCC, SE, CU) its 2048 workgroups ran on; then the times of a busy kernel under masks of 8 / 32 / 248 bits.
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/cu_mask_probe scripts/cu_mask_probe.hip && /tmp/cu_mask_probe
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
static std::set<unsigned> where(hipStream_t s, unsigned *d, int n, float *ms_out, int spin) {
    std::vector<unsigned> h(n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, s, d, spin);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    if (ms_out) hipEventElapsedTime(ms_out, e0, e1);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    for (unsigned v : h) cus.insert((((v >> 16) & 0xf) << 8) | (((v >> 13) & 0x7) << 4) | ((v >> 8) & 0xf));   // xcc | se | cu
    hipEventDestroy(e0); hipEventDestroy(e1);
    return cus;
}
int main() {
    const int n = 2048;
    unsigned *d; hipMalloc(&d, n * 4);
    for (int b = 0; b < 256; b += (b < 40 ? 1 : 37)) {
        uint32_t m[8] = {0};
        m[b >> 5] = 1u << (b & 31);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("bit %d: create failed\n", b); continue; }
        std::set<unsigned> c = where(s, d, n, nullptr, 200);
        printf("bit %3d -> %zu CUs:", b, c.size());
        int cnt = 0;
        for (unsigned v : c) if (cnt++ < 10) printf(" (x%u s%u c%u)", v >> 8, (v >> 4) & 7, v & 15);
        printf("\n");
        hipStreamDestroy(s);
    }
    struct { const char *name; int nbits; } tests[] = {{"8 bits (0..7)", 8}, {"32 bits (0..31)", 32}, {"248 bits (8..255)", -248}, {"256 bits", 256}};
    for (auto &t : tests) {
        uint32_t m[8] = {0};
        for (int b = 0; b < 256; ++b) { bool on = t.nbits > 0 ? b < t.nbits : b >= 8; if (on) m[b >> 5] |= 1u << (b & 31); }
        hipStream_t s; hipExtStreamCreateWithCUMask(&s, 8, m);
        float ms; where(s, d, n, nullptr, 2000);
        std::set<unsigned> c = where(s, d, n, &ms, 20000);
        printf("%-18s %3zu CUs, busy kernel %.3f ms\n", t.name, c.size(), ms);
        hipStreamDestroy(s);
    }
    return 0;
}
