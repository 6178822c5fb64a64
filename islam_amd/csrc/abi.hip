// libislam_hip.so: error buffer and ABI version.
#include "common.h"

namespace islam {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace islam

extern "C" const char* islam_last_error(void) { return islam::err_buf(); }
extern "C" int islam_abi_version(void) { return 1; }

// ---- measurement aid (bench.py): the shader clock the chip sustains RIGHT NOW.  One wavefront runs a dependent fp64 FMA chain and
// reads both counters around it: wall_clock64() ticks at the constant hipDeviceAttributeWallClockRate, clock64() in shader cycles.
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(long long* out, int iters) {
    const long long w0 = wall_clock64(), c0 = clock64();
    double a = threadIdx.x * 1e-3 + 1.0;
    const double b = 1.0000001;
#pragma unroll 1
    for (int i = 0; i < iters / 32; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) a = fma(a, b, 1e-9);
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = a > 0.0; }
}
}  // namespace

extern "C" int islam_clock_probe(long long* out3, int iters, void* stream) {
    if (!out3 || iters < 32) return islam::fail(ISLAM_EARG, "islam_clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, islam::as_stream(stream), out3, iters);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}
// ---- measurement aid (bench.py, roofline.latency_model): what ONE kernel in a chain of dependent launches costs on this platform whatever
// it does -- `n` launches of a kernel that does nothing, back to back on `stream` between one pair of events.  Synchronises the stream.
namespace {
struct ProbePad { double pad[40]; };                     // (a kernel-argument block the size the LM kernels pass)
__global__ void launch_cost_kernel(double* p, ProbePad b) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && b.pad[0] == 12345.0) p[0] = 1.0;
}
}  // namespace
extern "C" int islam_launch_cost_probe(int grid, int block, int n, float* us_per_launch, void* stream) {
    if (grid < 1 || block < 1 || block > 1024 || n < 1 || !us_per_launch) return islam::fail(ISLAM_EARG, "islam_launch_cost_probe: bad argument");
    hipStream_t s = islam::as_stream(stream);
    double* p = nullptr;
    ISLAM_HIP_CHECK(hipMalloc(&p, 64));
    hipEvent_t e0, e1;
    ISLAM_HIP_CHECK(hipEventCreate(&e0));
    ISLAM_HIP_CHECK(hipEventCreate(&e1));
    ProbePad b{};
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(launch_cost_kernel, dim3(grid), dim3(block), 0, s, p, b);
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    ISLAM_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(launch_cost_kernel, dim3(grid), dim3(block), 0, s, p, b);
    ISLAM_HIP_CHECK(hipEventRecord(e1, s));
    ISLAM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    ISLAM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(p);
    *us_per_launch = ms * 1e3f / n;
    return ISLAM_OK;
}
extern "C" int islam_wall_clock_khz(int device) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess) return -1;
    return khz;
}
