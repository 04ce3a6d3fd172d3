// Diagnostic: shader clock seen by s_memtime vs the 100 MHz s_memrealtime, under a sustained VALU load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long* out, int iters, float seed) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float a = seed + threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a == 12345.678f) out[2] = 1;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    std::vector<double> ghz;
    for (int rep = 0; rep < 400; ++rep) {
        hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, 0, d, 20000, 1.0f);
        unsigned long long h[2];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        ghz.push_back((double)h[0] / ((double)h[1] * 10.0));  // memrealtime: 100 MHz -> 10 ns per tick
        if (rep % 50 == 0 || rep == 399) printf("rep %d: memtime %llu ticks, realtime %llu ticks -> %.3f GHz\n", rep, h[0], h[1], ghz.back());
    }
    return 0;
}
