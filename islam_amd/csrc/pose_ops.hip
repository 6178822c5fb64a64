// Elementwise tail of the trainable pose head's convolutions (reference Network/VOFlowNet.py:42-157: conv + bias + ReLU, and
// the residual blocks' conv + bias + shortcut + ReLU), fp32 channels-last, forward AND backward in one launch each.
//
// PyTorch runs a convolution with bias on ROCm as MIOpen's kernel + a broadcast add, then the ReLU (a clamp), then -- in the
// residual blocks -- the shortcut add and another clamp; backward: threshold_backward + a per-channel sum for the bias gradient.
// On the pose head's small late layers (256 channels at 4x5 and 2x3 pixels) each of those is a 4-7 us launch: ~300 of the ~650
// launches of a forward + backward.  Here:
//   forward   y = act(x + bias[c] (+ res))                              one launch  (was 2-4)
//   backward  gx = gy * (y > 0);  gbias[c] = sum over pixels of gx      one launch  (was 2-3): per-workgroup partial sums in a fixed
//             order, folded in workgroup order by the workgroup that draws the last ticket -- deterministic
// The convolution itself (and its data / weight gradients) stays on MIOpen with the pinned solution set.
#include <hip/hip_runtime.h>

#include "common.h"

using namespace islam;

namespace {

__global__ __launch_bounds__(256) void bias_act_f32_kernel(const float4* __restrict__ x, const float* __restrict__ bias,
                                                           const float4* __restrict__ res, float4* __restrict__ y, long long total4, int C4,
                                                           int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    const float4 b = reinterpret_cast<const float4*>(bias)[c4];
    float4 v = x[i];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;                       // (conv + bias) first, like the convolution's own bias add
    if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    y[i] = v;
}

constexpr int BW_ROWS = 128;           // pixels per workgroup (a thread walks BW_ROWS / (256 / C4) of them)

// workgroup w: pixels [w * BW_ROWS, ...); thread t: channel group t % C4, pixel offset t / C4 (C4 <= 64 -> >= 4 pixel lanes)
__global__ __launch_bounds__(256) void bias_act_bwd_f32_kernel(const float4* __restrict__ gy, const float4* __restrict__ y,
                                                               float4* __restrict__ gx, float* __restrict__ gbias,
                                                               float* __restrict__ partial, unsigned* __restrict__ ticket, long long npix,
                                                               int C4, int relu) {
    __shared__ float4 red[256];
    __shared__ int last;
    const int t = threadIdx.x, c4 = t % C4, po = t / C4, lanes = 256 / C4;
    const long long p0 = (long long)blockIdx.x * BW_ROWS;
    float4 acc{0.f, 0.f, 0.f, 0.f};
    if (po < lanes) {
        for (long long p = p0 + po; p < p0 + BW_ROWS && p < npix; p += lanes) {
            const long long i = p * C4 + c4;
            float4 g = gy[i];
            if (relu) {
                const float4 o = y[i];
                g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
            }
            gx[i] = g;
            acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        }
    }
    red[t] = acc;
    __syncthreads();
    if (t < C4) {                                                          // pixel lanes of this workgroup, in order
        float4 s = red[t];
        for (int l = 1; l < lanes; ++l) { const float4 v = red[l * C4 + t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * C4 + t] = s;
    }
    // the workgroup that draws the last ticket folds the partial sums in workgroup order (release / acquire at agent scope around
    // the ticket: the partial sums are a few KB)
    __threadfence();
    __syncthreads();
    if (t == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // all 256 threads: thread (c4, po) sums the workgroups po, po + lanes, ... (loads in flight together), then the pixel lanes are
    // added in order as above -- a fixed order whichever workgroup folds
    {
        float4 s{0.f, 0.f, 0.f, 0.f};
        if (po < lanes) {
            for (unsigned w = po; w < gridDim.x; w += lanes) {
                const float4 v = reinterpret_cast<const float4*>(partial)[(size_t)w * C4 + c4];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        red[t] = s;
    }
    __syncthreads();
    if (t < C4) {
        float4 s = red[t];
        for (int l = 1; l < lanes; ++l) { const float4 v = red[l * C4 + t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        reinterpret_cast<float4*>(gbias)[t] = s;
    }
    if (t == 0) *ticket = 0u;
}

}  // namespace

extern "C" {

int islam_bias_act_f32_nhwc(const float* x, const float* bias, const float* res, float* y, long long pixels, int C, int relu, void* stream) {
    if (pixels < 1 || C < 4 || (C & 3)) return fail(ISLAM_EARG, "islam_bias_act_f32_nhwc: C=%d must be a multiple of 4", C);
    const long long total4 = pixels * (C / 4);
    hipLaunchKernelGGL(bias_act_f32_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), (const float4*)x, bias,
                       (const float4*)res, (float4*)y, total4, C / 4, relu);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// floats of scratch islam_bias_act_bwd_f32_nhwc needs for `pixels` pixels of C channels (+ one zero-initialised ticket word, which
// the launch leaves at zero)
long long islam_bias_act_bwd_scratch_floats(long long pixels, int C) { return ((pixels + BW_ROWS - 1) / BW_ROWS) * (long long)C; }

int islam_bias_act_bwd_f32_nhwc(const float* gy, const float* y, float* gx, float* gbias, float* scratch, unsigned* ticket, long long pixels,
                                int C, int relu, void* stream) {
    if (pixels < 1 || C < 4 || (C & 3) || C > 256) return fail(ISLAM_EARG, "islam_bias_act_bwd_f32_nhwc: C=%d must be a multiple of 4, <= 256", C);
    const unsigned nwg = (unsigned)((pixels + BW_ROWS - 1) / BW_ROWS);
    hipLaunchKernelGGL(bias_act_bwd_f32_kernel, dim3(nwg), dim3(256), 0, as_stream(stream), (const float4*)gy, (const float4*)y, (float4*)gx,
                       gbias, scratch, ticket, pixels, C / 4, relu);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // extern "C"
