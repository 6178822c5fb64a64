// One level of PWC-Net's feature pyramid in ONE launch (frozen flow net, bf16 operands / fp32 accumulation like islam_conv3x3_mfma).
//
// Reference: Network/PWC/PWCNet.py:78-95 (conv1a, conv1aa, conv1b ... = conv(k=3, stride 2) + two conv(k=3, stride 1), each
// Conv2d + LeakyReLU(0.1), :16-20) and their use at :240-251.  At the two large levels (16 channels at 1/2 resolution, 32 at 1/4) the three
// layers are memory traffic, not arithmetic: launched one by one they move the level's fp32 activation through HBM five times
// (0.33 + 0.19 ms of the 4.3 ms flow forward at B = 8, half of it MIOpen's stride-2 kernel + a bias launch + a LeakyReLU launch).
// Here a workgroup produces a TH x TW tile of the level's output from the (2(TH+4)+1) x (2(TW+4)+1) source patch it depends on:
//   source patch  fp32 NCHW -> bf16 [y][x][SC] in LDS (zero outside the image = the first layer's padding)
//   layer A       stride 2, the (TH+4) x (TW+4) pixels layer B will read  -> bf16 [pixel][C] in LDS
//   layer B       stride 1, (TH+2) x (TW+2) pixels                         -> bf16 [pixel][C] in LDS (over the source patch)
//   layer C       stride 1, TH x TW pixels                                 -> fp32 NCHW in HBM
// Pixels of an intermediate tile that lie outside the image are stored as zero (the next layer's padding).  The halo is recomputed
// per tile (layer A 1.4-1.9x, layer B 1.2-1.4x) -- matrix-core work that costs less than one pass through HBM.
//
// GEMM view per layer: M = output channels (A operand: weights, held in registers for the layer), N = 16 pixels of the destination
// tile (B operand: im2col gathered from LDS, 16 bytes per lane), K = 9 taps x source channels in steps of 32 on
// v_mfma_f32_16x16x32_bf16.  K index = tap * SC + channel; a lane's 8 consecutive K values are 8 channels of one tap (SC = 16, 32) or
// the 4 channels of two taps (SC = 4: the 3-channel image, padded).  K is padded to a multiple of 32 with zero weights (the padded
// taps re-read tap 8).  Rounding points are those of the launch-per-layer path: activations are rounded to bf16 (nearest even)
// once, after bias + LeakyReLU in fp32.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "../../include/islam_hip.h"
#include "common.h"

namespace {

using namespace islam;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2bf(float f) {              // round to nearest even (finite inputs)
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

template <int SC> constexpr int k_steps() { return (9 * SC + 31) / 32; }

// per-lane LDS offsets (bf16 elements, relative to the destination pixel's source pixel) of the K steps of one layer
template <int SC, int NS>
__device__ __forceinline__ void tap_offsets(int kb, int srcW, int (&off)[NS], int (&off2)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int q = s * 4 + kb;
        if constexpr (SC == 4) {
            const int ta = min(2 * q, 8), tb = min(2 * q + 1, 8);
            off[s] = ((ta / 3) * srcW + ta % 3) * 4;
            off2[s] = ((tb / 3) * srcW + tb % 3) * 4;
        } else {
            constexpr int KPT = SC / 8;
            const int tap = min(q / KPT, 8);
            off[s] = ((tap / 3) * srcW + tap % 3) * SC + (q % KPT) * 8;
            off2[s] = 0;
        }
    }
}

template <int SC, int STRIDE, int NS, int MB, int NW, class Emit>
__device__ __forceinline__ void conv_layer(const unsigned short* __restrict__ src, int srcW, int dstW, int dstN,
                                           const unsigned short* __restrict__ wp, int wave, int lane, Emit emit) {
    const int kb = lane >> 4, li = lane & 15;
    bf16x8 wf[NS][MB];                                           // the layer's weights: row = output channel, 8 K values per lane and step
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) wf[s][mb] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(mb * 16 + li) * (NS * 32) + s * 32 + kb * 8);
    int off[NS], off2[NS];
    tap_offsets<SC, NS>(kb, srcW, off, off2);
    for (int g = wave; g * 16 < dstN; g += NW) {
        const int p = g * 16 + li, pc = min(p, dstN - 1), py = pc / dstW, px = pc - py * dstW;
        const unsigned short* base = src + (size_t)((STRIDE * py) * srcW + STRIDE * px) * SC;
        f32x4 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 bf[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (SC == 4) {
                const uint2 a = *reinterpret_cast<const uint2*>(base + off[s]), b = *reinterpret_cast<const uint2*>(base + off2[s]);
                const uint4 v = make_uint4(a.x, a.y, b.x, b.y);
                bf[s] = *reinterpret_cast<const bf16x8*>(&v);
            } else {
                bf[s] = *reinterpret_cast<const bf16x8*>(base + off[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][mb], bf[s], acc[mb], 0, 0, 0);
        if (p < dstN) emit(p, py, px, kb, acc);
    }
}

template <int CIN, int SC, int C, int TH, int TW, int THREADS>
__global__ __launch_bounds__(THREADS) void pyr_level_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wA,
                                                            const float* __restrict__ bA, const unsigned short* __restrict__ wB,
                                                            const float* __restrict__ bB, const unsigned short* __restrict__ wC,
                                                            const float* __restrict__ bC, float* __restrict__ y, int H, int W, int Ho, int Wo,
                                                            float slope, int tiles_x, int tiles_per_img, int bh, long long s_img, long long s_half) {
    constexpr int R1 = TH + 4, W1 = TW + 4, R2 = TH + 2, W2 = TW + 2, R0 = 2 * R1 + 1, W0 = 2 * W1 + 1;
    constexpr int NSA = k_steps<SC>(), NSB = k_steps<C>(), MB = C / 16, NW = THREADS / 64;
    constexpr int T0 = ((R0 * W0 * SC > R2 * W2 * C ? R0 * W0 * SC : R2 * W2 * C) + 7) / 8 * 8;
    extern __shared__ __align__(16) unsigned short lds[];
    unsigned short* t0 = lds;                  // source patch [R0][W0][SC]; later layer B's output [R2*W2][C]
    unsigned short* t1 = lds + T0;             // layer A's output [R1*W1][C]
    const int b = blockIdx.x / tiles_per_img, tile = blockIdx.x - b * tiles_per_img;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int sy0 = 2 * (oy0 - 2) - 1, sx0 = 2 * (ox0 - 2) - 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- source patch: thread = (channel pair, pixel), consecutive threads along x; eight requests in flight per thread
    {
        constexpr int NPIX = R0 * W0, ITEMS = NPIX * (SC / 2), UN = 8;
        // image b of the launch = image b % bh of half b / bh of the source (islam_flow_pyramid_level_pair: the two frames of a (B, 2 CIN, H, W)
        // pair tensor as a batch of 2 B images, no concatenated copy); plain form: bh = B, s_half = 0
        const float* xb = x + (size_t)(b % bh) * s_img + (size_t)(b / bh) * s_half;
        for (int i0 = threadIdx.x; i0 < ITEMS; i0 += THREADS * UN) {
            float v0[UN], v1[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int it = i0 + u * THREADS, itc = min(it, ITEMS - 1);
                const int cp = itc / NPIX, pix = itc - cp * NPIX, sy = pix / W0, sx = pix - sy * W0;
                const int gy = sy0 + sy, gx = sx0 + sx, c0 = 2 * cp;
                const bool in = it < ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const size_t o = ((size_t)min(c0, CIN - 1) * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1);
                const float a = xb[o], c = xb[o + ((c0 + 1 < CIN) ? (size_t)H * W : (size_t)0)];
                v0[u] = (in && c0 < CIN) ? a : 0.0f;
                v1[u] = (in && c0 + 1 < CIN) ? c : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int it = i0 + u * THREADS;
                if (it < ITEMS) {
                    const int cp = it / NPIX, pix = it - cp * NPIX;
                    *reinterpret_cast<unsigned*>(t0 + (size_t)pix * SC + 2 * cp) = f2bf(v0[u]) | (f2bf(v1[u]) << 16);
                }
            }
        }
    }
    __syncthreads();

    auto act = [&](float v, float bias) { v += bias; return v > 0.0f ? v : v * slope; };
    // intermediate tiles: bias + LeakyReLU in fp32, one rounding to bf16, zero outside the image (the next layer's padding)
    auto to_lds = [&](unsigned short* dst, const float* __restrict__ bias, int gy0, int gx0) {
        return [=](int p, int py, int px, int kb, const f32x4 (&acc)[MB]) {
            const int gy = gy0 + py, gx = gx0 + px;
            const bool in = gy >= 0 && gy < Ho && gx >= 0 && gx < Wo;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int c = mb * 16 + 4 * kb;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
                uint2 o = make_uint2(0u, 0u);
                if (in) {
                    o.x = f2bf(act(acc[mb][0], bv[0])) | (f2bf(act(acc[mb][1], bv[1])) << 16);
                    o.y = f2bf(act(acc[mb][2], bv[2])) | (f2bf(act(acc[mb][3], bv[3])) << 16);
                }
                *reinterpret_cast<uint2*>(dst + (size_t)p * C + c) = o;
            }
        };
    };
    conv_layer<SC, 2, NSA, MB, NW>(t0, W0, W1, R1 * W1, wA, wave, lane, to_lds(t1, bA, oy0 - 2, ox0 - 2));
    __syncthreads();                                           // t1 complete; the source patch is dead
    conv_layer<C, 1, NSB, MB, NW>(t1, W1, W2, R2 * W2, wB, wave, lane, to_lds(t0, bB, oy0 - 1, ox0 - 1));
    __syncthreads();
    float* yb = y + (size_t)b * C * Ho * Wo;
    conv_layer<C, 1, NSB, MB, NW>(t0, W2, TW, TH * TW, wC, wave, lane, [&](int p, int py, int px, int kb, const f32x4 (&acc)[MB]) {
        const int gy = oy0 + py, gx = ox0 + px;
        if (gy >= Ho || gx >= Wo) return;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int c = mb * 16 + 4 * kb;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bC + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) yb[((size_t)(c + i) * Ho + gy) * Wo + gx] = act(acc[mb][i], bv[i]);
        }
    });
}

template <int CIN, int SC, int C, int TH, int TW, int THREADS>
int launch_level(const float* x, const unsigned short* wA, const float* bA, const unsigned short* wB, const float* bB, const unsigned short* wC,
                 const float* bC, float* y, int B, int H, int W, float slope, hipStream_t s, bool pair = false) {
    constexpr int R1 = TH + 4, W1 = TW + 4, R2 = TH + 2, W2 = TW + 2, R0 = 2 * R1 + 1, W0 = 2 * W1 + 1;
    constexpr int T0 = ((R0 * W0 * SC > R2 * W2 * C ? R0 * W0 * SC : R2 * W2 * C) + 7) / 8 * 8;
    constexpr size_t lds = ((size_t)T0 + (size_t)R1 * W1 * C) * sizeof(unsigned short);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)pyr_level_kernel<CIN, SC, C, TH, TW, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((pyr_level_kernel<CIN, SC, C, TH, TW, THREADS>), dim3((unsigned)(tiles_x * tiles_y * B)), dim3(THREADS), lds, s, x, wA, bA, wB, bB,
                       wC, bC, y, H, W, Ho, Wo, slope, tiles_x, tiles_x * tiles_y, pair ? B / 2 : B, (long long)(pair ? 2 : 1) * CIN * H * W,
                       pair ? (long long)CIN * H * W : 0ll);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int source_channels(int Cin) { return Cin <= 4 ? 4 : Cin; }

}  // namespace

extern "C" {

// bf16 elements of one packed 3x3 weight: [Cout][ceil(9 * SC / 32) * 32], K index = (ky * 3 + kx) * SC + c, SC = 4 for Cin <= 4
size_t islam_pyramid_packed_elems(int Cin, int Cout) {
    const int SC = source_channels(Cin);
    return (size_t)Cout * ((9 * SC + 31) / 32 * 32);
}

int islam_flow_pyramid_level(const float* x, const uint16_t* wA, const float* bA, const uint16_t* wB, const float* bB, const uint16_t* wC,
                             const float* bC, float* y, int B, int Cin, int H, int W, int C, float slope, void* stream) {
    if (B < 1 || H < 2 || W < 2) return fail(ISLAM_EARG, "islam_flow_pyramid_level: bad shape B=%d H=%d W=%d", B, H, W);
    if ((size_t)B * std::max(Cin, C) * H * W >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_flow_pyramid_level: tensor too large for 32-bit offsets");
    hipStream_t s = as_stream(stream);
    static const int variant = [] { const char* e = std::getenv("ISLAM_PYR_TILE"); return e ? std::atoi(e) : 0; }();   // (A/B runs)
    if (Cin == 3 && C == 16) {
        if (variant == 1) return launch_level<3, 4, 16, 8, 32, 256>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        if (variant == 2) return launch_level<3, 4, 16, 16, 32, 512>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        if (variant == 3) return launch_level<3, 4, 16, 16, 16, 256>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        if (variant == 4) return launch_level<3, 4, 16, 32, 32, 512>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        return launch_level<3, 4, 16, 16, 32, 256>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
    }
    if (Cin == 16 && C == 32) {
        if (variant == 1) return launch_level<16, 16, 32, 8, 32, 256>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        if (variant == 2) return launch_level<16, 16, 32, 8, 32, 512>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        if (variant == 3) return launch_level<16, 16, 32, 16, 16, 256>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        if (variant == 4) return launch_level<16, 16, 32, 16, 16, 512>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
        return launch_level<16, 16, 32, 8, 16, 256>(x, wA, bA, wB, bB, wC, bC, y, B, H, W, slope, s);
    }
    return fail(ISLAM_EARG, "islam_flow_pyramid_level: (Cin, C) = (%d, %d); built for PWC-Net's levels 1 and 2: (3, 16), (16, 32)", Cin, C);
}

// Level 1 on the two frames of a pair tensor: x (B, 6, H, W) fp32 = [frame 1 | frame 2] along the channels (what PWCDCNet.forward is
// called with, PWCNet.py:224-226); the result is (2 B, 16, H/2, W/2) with the B first-frame images first -- the batch
// torch.cat((x[:, :3], x[:, 3:]), 0) would give, without that copy.
int islam_flow_pyramid_level_pair(const float* x, const uint16_t* wA, const float* bA, const uint16_t* wB, const float* bB, const uint16_t* wC,
                                  const float* bC, float* y, int B, int H, int W, float slope, void* stream) {
    if (B < 1 || H < 2 || W < 2) return fail(ISLAM_EARG, "islam_flow_pyramid_level_pair: bad shape B=%d H=%d W=%d", B, H, W);
    if ((size_t)B * 2 * 16 * H * W >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_flow_pyramid_level_pair: tensor too large for 32-bit offsets");
    return launch_level<3, 4, 16, 16, 32, 256>(x, wA, bA, wB, bB, wC, bC, y, 2 * B, H, W, slope, as_stream(stream), true);
}

}  // extern "C"
