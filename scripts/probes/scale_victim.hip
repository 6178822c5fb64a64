// Debug twin of scale_partial_kernel (csrc/scale_ls.hip): the same loop, and every lane also stores WHAT IT READ (flow x, flow y, edge byte,
// disparity) -- so that a launch with a wrong mask (scripts/debug/coherence_ops.py) shows which load came back wrong.
// build: hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared scripts/probes/scale_victim.hip -o islam_amd/lib/libislam_probe_scale.so
#include <hip/hip_runtime.h>
#include <cstdint>
constexpr int NS = 18, NBLK = 16;
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ disp, const float* __restrict__ flow, const uint8_t* __restrict__ edge,
                                                     const float* __restrict__ disp_th, uint8_t* __restrict__ mask, float* __restrict__ seen,
                                                     double* __restrict__ partial, int H, int W, int variant) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const float dth = disp_th[b];
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
    const float* dp = disp + (size_t)b * HW;
    const float* fxp = flow + (size_t)b * 2 * HW;
    const float* fyp = fxp + HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += NBLK * 256) {
        const int v = i / W, u = i - v * W;
        const float uf = (float)u, vf = (float)v;
        const float d = dp[i], flx = fxp[i], fly = fyp[i];
        const float fu = flx + uf, fv = fly + vf;
        const bool inside = fu >= 0.f && fu <= (float)W && fv >= 0.f && fv <= (float)H;
        bool m = inside && (sqrtf(flx * flx + fly * fly) > 0.f);
        unsigned char eb = 7;
        if (variant == 0) { if (edge) { eb = edge[(size_t)b * HW + i]; m = m && (eb != 0); } }        // as the product: the edge byte is loaded under a branch
        else { eb = edge[(size_t)b * HW + i]; m = m && (eb != 0); }                                 // unconditional load
        const float du = -d + uf;
        const bool dm = du >= 0.f && du <= (float)W && d >= dth;
        m = m && dm;
        mask[(size_t)b * HW + i] = m ? 1 : 0;
        float* sp = seen + 4 * ((size_t)b * HW + i);
        sp[0] = flx; sp[1] = fly; sp[2] = (float)eb; sp[3] = d;
        if (m) {
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[k] += (double)(flx * (k + 1)) + fly;
        }
    }
    __shared__ double red[4][NS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) partial[((size_t)b * NBLK + blockIdx.x) * NS + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
extern "C" int victim_launch(const float* disp, const float* flow, const uint8_t* edge, const float* disp_th, uint8_t* mask, float* seen, double* partial,
                             int B, int H, int W, int variant, void* stream) {
    hipLaunchKernelGGL(victim_kernel, dim3(NBLK, B), dim3(256), 0, (hipStream_t)stream, disp, flow, edge, disp_th, mask, seen, partial, H, W, variant);
    return (int)hipGetLastError();
}
