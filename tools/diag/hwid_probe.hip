// Diagnostic: where do the workgroups of a 4096 x 64-lane launch land (XCC, SE, CU, SIMD), and is it stable?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(64) void probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    lds[threadIdx.x] = a;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    const int G = 4096;
    unsigned* d; hipMalloc(&d, G * 8);
    std::vector<unsigned> h(G * 2), prev;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(64), 36000, 0, d, 20000);  // 36 KB LDS: 4 workgroups per CU... see below
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_simd;
        for (int b = 0; b < G; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_simd[(xcc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd]++;
        }
        std::map<int, int> hist;
        for (auto& kv : per_simd) hist[kv.second]++;
        printf("rep %d: %zu distinct SIMDs; waves-per-SIMD histogram:", rep, per_simd.size());
        for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
        int same = 0;
        if (!prev.empty()) for (int b = 0; b < G; ++b) same += (prev[2 * b] >> 4 & 0xFFF) == (h[2 * b] >> 4 & 0xFFF) && prev[2 * b + 1] == h[2 * b + 1];
        printf("; same placement as previous launch: %d of %d\n", same, G);
        if (rep == 0) for (int b = 0; b < 24; ++b) printf("  block %d: xcc %u se %u sh %u cu %u simd %u wave %u\n", b, h[2*b+1] & 0xF, (h[2*b] >> 13) & 7, (h[2*b] >> 12) & 1, (h[2*b] >> 8) & 0xF, (h[2*b] >> 4) & 3, h[2*b] & 0xF);
        prev = h;
    }
    return 0;
}
