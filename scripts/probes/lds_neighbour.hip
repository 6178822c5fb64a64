// Does a workgroup with MORE THAN 64 KB of LDS disturb a small kernel that runs beside it on the same CUs?
// scripts/debug/coherence_ops.py: with conv_nhwc_kernel (73 KB of LDS per workgroup) looping on a side stream, islam_scale_ls on the main
// stream returned a wrong mask in half of its launches (16 consecutive pixels = lanes 48..63 of one wave), with conv3x3_mfma_kernel
// (44 KB) never.  This probe takes torch, the library and the graph out of the picture:
//   aggressor<>: 256 threads, `lds_bytes` of dynamic LDS, every thread writes and reads its stripe for ~20 us
//   victim:      out[i] = f(in[i]) with per-lane global loads only (optionally a small static LDS array used at the very end)
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/lds_neighbour.hip -o scripts/probes/lds_neighbour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void aggressor(float* sink, int lds_floats, int rounds) {
    extern __shared__ float lds[];
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < lds_floats; i += 256) lds[i] = (float)(i + r);
        __syncthreads();
        for (int i = threadIdx.x; i < lds_floats; i += 256) acc += lds[lds_floats - 1 - i];
        __syncthreads();
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ a, const float* __restrict__ b, const unsigned char* __restrict__ e,
                                              unsigned char* __restrict__ out, double* __restrict__ partial, int n) {
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float x = a[i], y = b[i];
        bool m = (x * x + y * y) > 0.f;
        m = m && e[i] != 0;
        out[i] = m ? 1 : 0;
        if (m) s += x;
    }
    if (USE_LDS) {
        __shared__ double red[256];
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) { double t = 0; for (int k = 0; k < 256; ++k) t += red[k]; partial[blockIdx.x] = t; }
    } else if (s == 1.2345) partial[blockIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 73, use_lds = argc > 2 ? atoi(argv[2]) : 1, iters = argc > 3 ? atoi(argv[3]) : 2000;
    const int n = 8 * 112 * 160;
    std::vector<float> ha(n), hb(n); std::vector<unsigned char> he(n), want(n), got(n);
    srand(1);
    for (int i = 0; i < n; ++i) { ha[i] = rand() / (float)RAND_MAX - 0.5f; hb[i] = rand() / (float)RAND_MAX - 0.5f; he[i] = rand() & 1; want[i] = he[i]; }
    float *a, *b, *sink; unsigned char *e, *out; double* partial;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&e, n)); CK(hipMalloc(&out, n)); CK(hipMalloc(&partial, 4096 * 8)); CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(e, he.data(), n, hipMemcpyHostToDevice));
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const size_t lds = (size_t)lds_kb * 1024;
    CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int bad_launches = 0; long long bad_bytes = 0;
    for (int it = 0; it < iters; ++it) {
        if (lds_kb > 0) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), lds, sa, sink, (int)(lds / 4), 6);
        for (int k = 0; k < 4; ++k) {
            CK(hipMemsetAsync(out, 0xff, n, sv));
            if (use_lds) hipLaunchKernelGGL(victim<true>, dim3(16 * 8), dim3(256), 0, sv, a, b, e, out, partial, n);
            else hipLaunchKernelGGL(victim<false>, dim3(16 * 8), dim3(256), 0, sv, a, b, e, out, partial, n);
            CK(hipMemcpyAsync(got.data(), out, n, hipMemcpyDeviceToHost, sv));
            CK(hipStreamSynchronize(sv));
            int nb = 0, first = -1;
            for (int i = 0; i < n; ++i) if (got[i] != want[i]) { ++nb; if (first < 0) first = i; }
            if (nb) { ++bad_launches; bad_bytes += nb; if (bad_launches <= 5) printf("  launch %d.%d: %d wrong bytes, first at %d (lane %d of its wave), got %d want %d\n", it, k, nb, first, first & 63, got[first], want[first]); }
        }
        CK(hipStreamSynchronize(sa));
    }
    printf("aggressor LDS %d KB, victim %s LDS: %d of %d victim launches wrong, %lld bytes\n", lds_kb, use_lds ? "with" : "without", bad_launches, 4 * iters, bad_bytes);
    return 0;
}
