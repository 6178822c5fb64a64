// Launch cost of (nearly) empty kernels as a function of workgroup size, dynamic LDS and grid: back-to-back dependent launches on one
// stream between two events.  hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost  (scripts/probes/, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Big { double pad[40]; };

__global__ void k_empty(double* p, Big b) {
    extern __shared__ double lds[];
    if (threadIdx.x == 0 && blockIdx.x == 0 && b.pad[0] == 12345.0) { lds[0] = 1.0; p[0] = lds[0]; }
}

static float run(int grid, int block, size_t lds, int n) {
    double* p;
    hipMalloc(&p, 64);
    Big b{};
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, 0, p, b);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, 0, p, b);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(p);
    return ms * 1e3f / n;
}

int main() {
    const int n = 2000;
    printf("grid  block  lds_KB   us per launch (back to back, same stream)\n");
    const int cfg[][3] = {{1, 64, 0}, {256, 64, 0}, {248, 640, 0}, {248, 640, 64}, {248, 640, 140}, {248, 192, 34}, {1024, 128, 27}, {248, 256, 140}, {2048, 256, 0}, {8960, 256, 73}};
    for (auto& c : cfg) printf("%5d %5d %6d   %.2f\n", c[0], c[1], c[2], run(c[0], c[1], (size_t)c[2] * 1024, n));
    return 0;
}
