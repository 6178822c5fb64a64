// Effective shader clock during a short one-wave-per-CU kernel (like the PVGO solver levels).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* out, double* sink, int iters) {
    long long w0 = wall_clock64(), c0 = clock64();
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001;
#pragma unroll 1
    for (int i = 0; i < iters / 32; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) a = fma(a, b, 1e-9);      // dependent fp64 FMA chain, 32 per branch
    }
    long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = w1 - w0; out[2 * blockIdx.x + 1] = c1 - c0; }
    sink[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
    long long* d; double* s;
    hipMalloc(&d, 4096 * 16); hipMalloc(&s, 4096 * 64 * 8);
    int wfreq = 0; hipDeviceGetAttribute(&wfreq, hipDeviceAttributeWallClockRate, 0);
    int sclk = 0; hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, max shader clock %d kHz\n", wfreq, sclk);
    for (int iters : {1000, 10000, 100000}) for (int grid : {1, 256, 834}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d, s, iters);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double us = h[0] / (wfreq * 1e-3);
        printf("iters %6d grid %4d: %.1f us, %lld shader cycles -> %.0f MHz, %.2f cycles per dependent fma\n", iters, grid, us, h[1], h[1] / us, (double)h[1] / iters);
    }
    return 0;
}
