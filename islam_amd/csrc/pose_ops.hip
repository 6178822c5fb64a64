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

constexpr int BW_PER_THREAD = 4;       // pixels per thread: all of a thread's loads are requested before the first one is used
// pixels per workgroup: 256 / (C / 4) pixel lanes x BW_PER_THREAD (C = 32: 128 pixels, C = 256: 16 -- the late layers' 48-160 pixels
// spread over several workgroups instead of one thread walking 32 pixels, one dependent round trip each)
__host__ __device__ inline int bw_rows(int C4) { return (256 / C4) * BW_PER_THREAD; }

// workgroup w: pixels [w * rows, ...); thread t: channel group t % C4, pixel offset t / C4 (C4 <= 64 -> >= 4 pixel lanes)
__global__ __launch_bounds__(256) void bias_act_bwd_f32_kernel(const float4* __restrict__ gy, const float4* __restrict__ y,
                                                               float4* __restrict__ gx, float* __restrict__ gbias,
                                                               float* __restrict__ partial, unsigned* __restrict__ ticket, long long npix,
                                                               int C4, int relu) {
    __shared__ float4 red[256];
    __shared__ int last;
    const int t = threadIdx.x, c4 = t % C4, po = t / C4, lanes = 256 / C4;
    const long long p0 = (long long)blockIdx.x * bw_rows(C4);
    float4 acc{0.f, 0.f, 0.f, 0.f};
    {
        float4 g[BW_PER_THREAD], o[BW_PER_THREAD];
        bool ok[BW_PER_THREAD];
#pragma unroll
        for (int j = 0; j < BW_PER_THREAD; ++j) {
            const long long p = p0 + po + (long long)j * lanes;
            ok[j] = po < lanes && p < npix;
            const long long i = ok[j] ? p * C4 + c4 : 0;
            g[j] = gy[i];
            o[j] = relu ? y[i] : float4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int j = 0; j < BW_PER_THREAD; ++j) {
            if (!ok[j]) continue;
            float4 v = g[j];
            v.x = o[j].x > 0.f ? v.x : 0.f; v.y = o[j].y > 0.f ? v.y : 0.f; v.z = o[j].z > 0.f ? v.z : 0.f; v.w = o[j].w > 0.f ? v.w : 0.f;
            gx[(p0 + po + (long long)j * lanes) * C4 + c4] = v;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[t] = acc;
    __syncthreads();
    if (t < C4) {                                                          // pixel lanes of this workgroup, in order
        float4 s = red[t];
        for (int l = 1; l < lanes; ++l) { const float4 v = red[l * C4 + t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        if (gridDim.x == 1) { reinterpret_cast<float4*>(gbias)[t] = s; return; }      // one workgroup: its sums are the result
        // write-through (agent-coherent) stores + completion wait instead of a release fence: an agent-scope fence on MI355X is a
        // write-back walk of the XCD's whole L2 (several microseconds; two of them made this 2-us kernel a 14-us one)
        float* pp = partial + ((size_t)blockIdx.x * C4 + t) * 4;
        __hip_atomic_store(pp, s.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, s.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 2, s.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 3, s.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (gridDim.x == 1) return;
    // the workgroup that draws the last ticket folds the partial sums in workgroup order
    __syncthreads();
    if (t == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    // all 256 threads: thread (c4, po) sums the workgroups po, po + lanes, ... (loads in flight together), then the pixel lanes are
    // added in order as above -- a fixed order whichever workgroup folds
    {
        float4 s{0.f, 0.f, 0.f, 0.f};
        if (po < lanes) {
            for (unsigned w = po; w < gridDim.x; w += lanes) {                 // (agent-coherent loads: the other workgroups' stores)
                const float* q = partial + ((size_t)w * C4 + c4) * 4;
                s.x += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s.y += __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s.z += __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s.w += __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
long long islam_bias_act_bwd_scratch_floats(long long pixels, int C) {
    const int rows = bw_rows(C > 3 ? C / 4 : 1);
    return ((pixels + rows - 1) / rows) * (long long)C;
}

int islam_bias_act_bwd_f32_nhwc(const float* gy, const float* y, float* gx, float* gbias, float* scratch, unsigned* ticket, long long pixels,
                                int C, int relu, void* stream) {
    if (pixels < 1 || C < 4 || (C & 3) || C > 256) return fail(ISLAM_EARG, "islam_bias_act_bwd_f32_nhwc: C=%d must be a multiple of 4, <= 256", C);
    const int rows = bw_rows(C / 4);
    const unsigned nwg = (unsigned)((pixels + rows - 1) / rows);
    hipLaunchKernelGGL(bias_act_bwd_f32_kernel, dim3(nwg), dim3(256), 0, as_stream(stream), (const float4*)gy, (const float4*)y, (float4*)gx,
                       gbias, scratch, ticket, pixels, C / 4, relu);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // extern "C"
