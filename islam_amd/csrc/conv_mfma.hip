// conv_mfma.hip -- 3x3 convolution of the frozen PWC-Net flow network as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces, for the frozen (inference-only) flow network, what the reference runs through cuDNN
// (Network/PWC/PWCNet.py:16-20 `conv()` = Conv2d(k=3, padding=dilation) + LeakyReLU(0.1), :208-292 the network):
//   y[b, coff+n, ho, wo] = act( bias[n] + sum_{c,r,s} w[n,c,r,s] * x[b, c, ho*S + r*D - D, wo*S + s*D - D] )
// x, y stay fp32 NCHW (the layout of the correlation / warp kernels and of torch.cat); operands are rounded to bf16 when
// they are staged in LDS and accumulated in fp32 (v_mfma_f32_32x32x16_bf16) -- BASELINE config 2 "bf16 nets".  Bias and
// LeakyReLU are fused; `xoff`/`xtot` and `coff`/`ytot` let a layer read and write channel slices of the DenseNet-style
// concatenation buffer directly (no torch.cat).
//
// GEMM view per image: M = Cout (A operand: weights), N = pixels (B operand: im2col of x), K = 9*Cin; computing it this way
// round makes the accumulator's lane index the pixel x, so output stores are 128-byte rows of one channel.
// Workgroup: 256 threads, tile 32 x 8 output pixels x TN output channels; wave w owns pixel rows 2w, 2w+1.  K is walked in
// chunks of 16 input channels: the halo tile of the chunk ([y][x][16 ch] bf16, 48-byte pixel stride: conflict-free 16-byte
// operand reads) and the nine weight taps are staged in LDS once and feed 9 * 2 * TN/32 MFMAs per wave; the next chunk's
// global loads are in flight (registers) while the current chunk is multiplied.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <algorithm>

#include "../../include/islam_hip.h"
#include "common.h"

// scripts/conv_probe.sh builds experiment variants of this file (never the product library): ISLAM_CONV_PROBE=1 skips the multiply
// phase of conv3x3_mfma_kernel, =2 skips fetch + staging
#ifndef ISLAM_CONV_PROBE
#define ISLAM_CONV_PROBE 0
#endif

namespace {

using namespace islam;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TW = 32, TH = 8, KC = 16, PS = 24;     // PS: bf16 elements per LDS pixel / weight row (16 + 8 pad = 48 bytes)
constexpr int THREADS = 256;
constexpr int MAX_IN_PER_THREAD = 48;                // halo-tile (pixel, channel-pair) items a thread prefetches per chunk (large dilations / stride 2)

// two floats -> packed bf16 pair (a in the low half), round-to-nearest-even as torch's .to(bfloat16): one v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}

template <int TN, int NPRE>
__global__ __launch_bounds__(THREADS) void conv3x3_mfma_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp,
                                                                const float* __restrict__ bias, float* __restrict__ y, int Cin,
                                                                int CinP, int H, int W, int Cout, int CoutP, int Ho, int Wo,
                                                                int S, int D, int coff, int ytot, float slope, int tiles_x, int xoff,
                                                                int xtot) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int IH = (TH - 1) * S + 2 * D + 1, IW = (TW - 1) * S + 2 * D + 1;
    unsigned short* lin = lds;                               // [IH][IW][PS]
    unsigned short* lw = lds + (size_t)IH * IW * PS;         // [9][TN][PS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int wo0 = tx * TW, ho0 = ty * TH, n0 = blockIdx.y * TN, b = blockIdx.z;
    const float* xb = x + ((size_t)b * xtot + xoff) * H * W;
    const int gx0 = wo0 * S - D, gy0 = ho0 * S - D;
    const int npix = IH * IW;
    const int nitems = npix * (KC / 2);                      // (channel pair, pixel) items of one chunk, pixel fastest
    constexpr int NT = TN / 32;
    f32x16 acc[NT][2];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][p][i] = 0.0f;

    float pre0[NPRE], pre1[NPRE];                            // fully unrolled below: stays in registers
    uint4 prew[(9 * TN * 2 + THREADS - 1) / THREADS];
    constexpr int NW = (9 * TN * 2 + THREADS - 1) / THREADS;
    // The (pixel, channel pair) items a thread stages are the same for every chunk: decode them once.  Everything in the
    // channel loop is branch-free: an item outside the image reads a valid address and is zeroed by a select, a thread
    // without an item writes to a dummy LDS slot behind the two tiles.
    int goff[NPRE];                                          // offset inside the chunk's first channel plane, -1 outside the image
    int loff[NPRE];                                          // LDS element offset
    const size_t plane = (size_t)H * W;
    const int dummy = (IH * IW + 9 * TN) * PS;               // 16 bytes of scratch behind lin and lw
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int it = tid + k * THREADS;
        goff[k] = -1;
        loff[k] = dummy;
        if (it < nitems) {
            const int cp = it / npix, pix = it - cp * npix;
            const int yy = pix / IW, xx = pix - yy * IW;
            const int gy = gy0 + yy, gx = gx0 + xx;
            loff[k] = pix * PS + 2 * cp;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) goff[k] = (2 * cp * H + gy) * W + gx;
        }
    }
    int woff[NW], wlds[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int it = tid + k * THREADS;                    // 16-byte vector index: ((tap*TN + n)*2 + half)
        woff[k] = 0;
        wlds[k] = dummy;
        if (it < 9 * TN * 2) {
            const int half = it & 1, row = it >> 1, tap = row / TN, n = row - tap * TN;
            woff[k] = (tap * CoutP + n0 + n) * CinP + 8 * half;
            wlds[k] = row * PS + 8 * half;
        }
    }

    // Raw loads only: nothing here consumes a loaded value (the zero padding is applied when the chunk is staged), so the
    // loads stay in flight behind the MFMAs of the current chunk.  A last chunk with fewer than 16 channels left reads the
    // LAST 16 channels of the tensor instead (all exist); ops.pack_conv3x3_weight lays that chunk's weights out to match,
    // with zeros for the channels the previous chunk has already covered.  Cin < 16: channels past the end are read from
    // the last existing channel (finite data) and meet zero weights.
    auto fetch = [&](int c0) {
        const int rem = Cin - c0;
        if (rem >= KC || Cin >= KC) {                        // (uniform)
            const float* xc = xb + (size_t)min(c0, Cin - KC) * plane;
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const float* p = xc + max(goff[k], 0);       // outside the image: any valid address, zeroed when staged
                pre0[k] = p[0];
                pre1[k] = p[plane];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int c = 2 * ((tid + k * THREADS) / npix);
                const float* p = xb + (goff[k] < 0 ? (size_t)0 : goff[k] - (size_t)c * plane);
                pre0[k] = p[(size_t)min(c, rem - 1) * plane];
                pre1[k] = p[(size_t)min(c + 1, rem - 1) * plane];
            }
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const uint4* q = reinterpret_cast<const uint4*>(wp + (size_t)woff[k] + c0);
            prew[k] = make_uint4(q->x, q->y, q->z, q->w);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            *reinterpret_cast<unsigned*>(lin + loff[k]) = pack_bf16(pre0[k], pre1[k]) & ~(unsigned)(goff[k] >> 31);   // zero padding
#pragma unroll
        for (int k = 0; k < NW; ++k) *reinterpret_cast<uint4*>(lw + wlds[k]) = prew[k];
    };

    const int kg = lane >> 5, li = lane & 31;
    // operand addresses of tap (r, s): B = halo pixel rows of this wave's two output rows, A = the tap's weight rows
    const unsigned short* bbase = lin + ((size_t)(2 * wave * S) * IW + li * S) * PS + 8 * kg;
    const unsigned short* abase = lw + (size_t)li * PS + 8 * kg;
    auto load_tap = [&](int tap, bf16x8 (&bf)[2], bf16x8 (&af)[NT]) {
        const int r = tap / 3, s = tap - r * 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[p] = *reinterpret_cast<const bf16x8*>(bbase + ((size_t)(p * S + r * D) * IW + s * D) * PS);
#pragma unroll
        for (int a = 0; a < NT; ++a) af[a] = *reinterpret_cast<const bf16x8*>(abase + ((size_t)tap * TN + a * 32) * PS);
    };
    fetch(0);
    for (int c0 = 0; c0 < CinP; c0 += KC) {
        __syncthreads();                                     // the previous chunk's operand reads are done
#if ISLAM_CONV_PROBE != 2
        stage();
#endif
        __syncthreads();
#if ISLAM_CONV_PROBE != 2
        if (c0 + KC < CinP) fetch(c0 + KC);                  // in flight while this chunk is multiplied
#endif
#if ISLAM_CONV_PROBE == 1
        continue;
#endif
        // software pipeline over the nine taps: the operands of tap t+1 are requested before the MFMAs of tap t issue
        bf16x8 bq[2][2], aq[2][NT];
        load_tap(0, bq[0], aq[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) load_tap(tap + 1, bq[cur ^ 1], aq[cur ^ 1]);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    acc[a][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[cur][a], bq[cur][p], acc[a][p], 0, 0, 0);
        }
    }
    // epilogue: D row (channel) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), D col (pixel x) = lane&31
    float* yb = y + ((size_t)b * ytot + coff) * Ho * Wo;
    const int wo = wo0 + li;
    // the lane's bias values, all requested before the first one is used (index clamped: branch-free, so the loads are not
    // serialised behind the stores -- one load -> wait -> store per register was 16 * NT dependent round trips)
    float bb[NT][16];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int n = n0 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg;
            bb[a][reg] = bias ? bias[min(n, Cout - 1)] : 0.0f;
        }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ho = ho0 + 2 * wave + p;
            if (ho >= Ho || wo >= Wo) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n = n0 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                if (n < Cout) {
                    float v = acc[a][p][reg] + bb[a][reg];
                    v = v >= 0.0f ? v : v * slope;
                    yb[((size_t)n * Ho + ho) * Wo + wo] = v;
                }
            }
        }
}

}  // namespace

extern "C" {

size_t islam_conv3x3_packed_elems(int Cin, int Cout) {
    const int CinP = (Cin + 15) / 16 * 16, CoutP = (Cout + 63) / 64 * 64;
    return (size_t)9 * CoutP * CinP;
}

int islam_conv3x3_mfma(const float* x, const uint16_t* wpacked, const float* bias, float* y, int B, int Cin, int H, int W,
                       int Cout, int stride, int dilation, int xoff, int xtot, int coff, int ytot, float slope, void* stream) {
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(ISLAM_EARG, "islam_conv3x3_mfma: bad shape");
    if (stride != 1 && stride != 2) return fail(ISLAM_EARG, "islam_conv3x3_mfma: stride %d (1 or 2)", stride);
    if (dilation < 1 || dilation > 16) return fail(ISLAM_EARG, "islam_conv3x3_mfma: dilation %d (1..16)", dilation);
    const int S = stride, D = dilation;
    const int Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
    if (xoff < 0 || xoff + Cin > xtot) return fail(ISLAM_EARG, "islam_conv3x3_mfma: input slice %d+%d > %d", xoff, Cin, xtot);
    if (coff < 0 || coff + Cout > ytot) return fail(ISLAM_EARG, "islam_conv3x3_mfma: channel slice %d+%d > %d", coff, Cout, ytot);
    const int CinP = (Cin + 15) / 16 * 16, CoutP = (Cout + 63) / 64 * 64;
    const int IH = (TH - 1) * S + 2 * D + 1, IW = (TW - 1) * S + 2 * D + 1;
    if ((IH * IW * (KC / 2) + THREADS - 1) / THREADS > MAX_IN_PER_THREAD)
        return fail(ISLAM_EARG, "islam_conv3x3_mfma: halo tile %dx%d too large", IH, IW);
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    hipStream_t s = (hipStream_t)stream;
    const int TN = Cout > 32 ? 64 : 32;
    const int nper = (IH * IW * (KC / 2) + THREADS - 1) / THREADS;
    const size_t lds = ((size_t)IH * IW * PS + (size_t)9 * TN * PS + 8) * sizeof(unsigned short);   // + 16-byte dummy slot
    dim3 grid(tiles_x * tiles_y, (Cout + TN - 1) / TN, B);
#define ISLAM_CONV_LAUNCH(TN_, NPRE_)                                                                                          \
    do {                                                                                                                       \
        static size_t maxlds = 0;                                                                                              \
        if (lds > maxlds) {                                                                                                    \
            ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<TN_, NPRE_>,                                  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
            maxlds = lds;                                                                                                      \
        }                                                                                                                      \
        hipLaunchKernelGGL((conv3x3_mfma_kernel<TN_, NPRE_>), grid, dim3(THREADS), lds, s, x, wpacked, bias, y, Cin, CinP, H, W, \
                           Cout, CoutP, Ho, Wo, S, D, coff, ytot, slope, tiles_x, xoff, xtot);                                             \
    } while (0)
    if (TN == 64) {
        if (nper <= 12) ISLAM_CONV_LAUNCH(64, 12); else ISLAM_CONV_LAUNCH(64, MAX_IN_PER_THREAD);
    } else {
        if (nper <= 12) ISLAM_CONV_LAUNCH(32, 12); else ISLAM_CONV_LAUNCH(32, MAX_IN_PER_THREAD);
    }
#undef ISLAM_CONV_LAUNCH
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Bilinear resize of a channels-last bf16 tensor (the SPP branches and decoder up/down-samplings of the frozen stereo
// network, Network/PSM/submodule.py:124-155 / StereoNet7.py:100-146 F.upsample / F.interpolate).  One thread per output
// pixel and group of 8 channels (16 bytes): the four taps are 16-byte reads of a source that lives in L2, the store is
// lane-contiguous.  Arithmetic as ATen's upsample_bilinear2d (fp32 interpolation weights and accumulation).
namespace {

__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// element index -> (channel octet, x, y, image) of a (B, Ho, Wo, C8) tensor.  32-bit arithmetic whenever the count allows: a 64-bit
// integer division is a ~100-instruction software routine on the GPU, and six of them per 16-byte output made the element-wise kernels
// below issue-bound instead of memory-bound.
struct Idx4 { int cg, ox, oy, b; long long pix; };
__device__ __forceinline__ Idx4 split4(long long i, long long total, int C8, int Wo, int Ho) {
    Idx4 r;
    if (total <= 0x7fffffffLL) {
        unsigned p = (unsigned)i;
        r.cg = (int)(p % (unsigned)C8); p /= (unsigned)C8;
        r.pix = p;
        r.ox = (int)(p % (unsigned)Wo); p /= (unsigned)Wo;
        r.oy = (int)(p % (unsigned)Ho);
        r.b = (int)(p / (unsigned)Ho);
    } else {
        long long p = i;
        r.cg = (int)(p % C8); p /= C8;
        r.pix = p;
        r.ox = (int)(p % Wo); p /= Wo;
        r.oy = (int)(p % Ho);
        r.b = (int)(p / Ho);
    }
    return r;
}
__device__ __forceinline__ int octet_of(long long i, long long total, int C8) {
    return total <= 0x7fffffffLL ? (int)((unsigned)i % (unsigned)C8) : (int)(i % C8);
}


__global__ __launch_bounds__(256) void resize_bilinear_nhwc_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int C8,
                                                                        int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                                        int align, long long total, int yC8, int yoff8,
                                                                        const uint4* __restrict__ add) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx4 ix = split4(i, total, C8, Wo, Ho);
    const int cg = ix.cg, ox = ix.ox, oy = ix.oy, b = ix.b;
    const long long opix = ix.pix;
    float fy, fx;
    if (align) {
        fy = sh * oy;
        fx = sw * ox;
    } else {
        fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
        fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    }
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - y0, lx = fx - x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const uint4* base = x + (size_t)b * Hi * Wi * C8 + cg;
    const uint4 a = base[((size_t)y0 * Wi + x0) * C8], bq = base[((size_t)y0 * Wi + x1) * C8];
    const uint4 c = base[((size_t)y1 * Wi + x0) * C8], d = base[((size_t)y1 * Wi + x1) * C8];
    auto mix = [&](unsigned va, unsigned vb, unsigned vc, unsigned vd) {
        const float lo = hy * (hx * bf16_lo(va) + lx * bf16_lo(vb)) + ly * (hx * bf16_lo(vc) + lx * bf16_lo(vd));
        const float hi = hy * (hx * bf16_hi(va) + lx * bf16_hi(vb)) + ly * (hx * bf16_hi(vc) + lx * bf16_hi(vd));
        return pack_bf16(lo, hi);
    };
    uint4 o;
    o.x = mix(a.x, bq.x, c.x, d.x);
    o.y = mix(a.y, bq.y, c.y, d.y);
    o.z = mix(a.z, bq.z, c.z, d.z);
    o.w = mix(a.w, bq.w, c.w, d.w);
    if (add) {                                          // u + up(low): the up-sampled value is rounded to bf16 first, like the two ops
        const uint4 u = add[opix * C8 + cg];
        auto sum = [&](unsigned va, unsigned vb) { return pack_bf16(bf16_lo(va) + bf16_lo(vb), bf16_hi(va) + bf16_hi(vb)); };
        o.x = sum(o.x, u.x); o.y = sum(o.y, u.y); o.z = sum(o.z, u.z); o.w = sum(o.w, u.w);
    }
    y[opix * yC8 + yoff8 + cg] = o;                     // channels [8*yoff8, 8*yoff8 + C) of a (B,Ho,Wo,8*yC8) tensor
}

// torch.cat((F.interpolate(torch.cat(pieces, 1), size), tail), 1) in ONE launch (Network/PSM/submodule.py:139-152 as StereoNet7's
// `bigger` feature extractor uses it: the six pieces of the 320-channel pyramid feature up-sampled to the resolution of layer1's
// output, which joins them as the last 32 channels).  Bilinear resizing is per channel, so every piece is sampled where it lies; a
// thread produces one 16-byte channel group of one output pixel and consecutive threads write consecutive groups of the same pixel
// -- whole 704-byte pixels leave the CU contiguously, where six launches wrote 64 .. 256-byte pieces 704 bytes apart (1.8 - 2.3 TB/s)
// and a seventh copied the tail.  The arithmetic of a sample is resize_bilinear_nhwc_bf16_kernel's, bit for bit.
struct UpCatArgs {
    const uint4* src[8];          // pieces, (B, Hi, Wi, 8 c8[k]) channels-last bf16
    int c8[8], first8[9];         // channel groups of piece k, first group of piece k in the output
    int n;
    const uint4* tail;            // (B, Ho, Wo, 8 tail8) or nullptr
    int tail8;
};
__global__ __launch_bounds__(256) void upsample_cat_nhwc_bf16_kernel(UpCatArgs a, uint4* __restrict__ y, int Hi, int Wi, int Ho, int Wo,
                                                                     float sh, float sw, int align, long long total, int yC8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx4 ix = split4(i, total, yC8, Wo, Ho);
    const int cgo = ix.cg, ox = ix.ox, oy = ix.oy, b = ix.b;
    if (cgo >= a.first8[a.n]) {                          // the tail: a copy
        y[i] = a.tail[ix.pix * a.tail8 + (cgo - a.first8[a.n])];
        return;
    }
    const uint4* sp = a.src[0];                          // (selects, not an index: the argument block stays in scalar registers)
    int C8 = a.c8[0], f8 = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q)
        if (q < a.n && cgo >= a.first8[q]) { sp = a.src[q]; C8 = a.c8[q]; f8 = a.first8[q]; }
    const int cg = cgo - f8;
    float fy, fx;
    if (align) {
        fy = sh * oy;
        fx = sw * ox;
    } else {
        fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
        fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    }
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - y0, lx = fx - x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const uint4* base = sp + (size_t)b * Hi * Wi * C8 + cg;
    const uint4 p00 = base[((size_t)y0 * Wi + x0) * C8], p01 = base[((size_t)y0 * Wi + x1) * C8];
    const uint4 p10 = base[((size_t)y1 * Wi + x0) * C8], p11 = base[((size_t)y1 * Wi + x1) * C8];
    auto mix = [&](unsigned va, unsigned vb, unsigned vc, unsigned vd) {
        const float lo = hy * (hx * bf16_lo(va) + lx * bf16_lo(vb)) + ly * (hx * bf16_lo(vc) + lx * bf16_lo(vd));
        const float hi = hy * (hx * bf16_hi(va) + lx * bf16_hi(vb)) + ly * (hx * bf16_hi(vc) + lx * bf16_hi(vd));
        return pack_bf16(lo, hi);
    };
    uint4 o;
    o.x = mix(p00.x, p01.x, p10.x, p11.x);
    o.y = mix(p00.y, p01.y, p10.y, p11.y);
    o.z = mix(p00.z, p01.z, p10.z, p11.z);
    o.w = mix(p00.w, p01.w, p10.w, p11.w);
    y[i] = o;
}

// The same, laid out for the one call the nets make (352 channel groups... 44 per pixel, 50 M output groups): at one thread per group the
// kernel above is bound by its index arithmetic, not by memory (three runtime divisions, a seven-step select of the piece and 64-bit
// addresses per 16 bytes: 416 us for 1.06 GB = 2.5 TB/s).  Here the grid is (groups of a row, output row, image): the row's vertical taps
// and weights are uniform, the piece of a channel group comes out of a 16-byte table entry in LDS built once per workgroup, one
// division per thread, 32-bit offsets.  Same arithmetic per sample (the `mix` expression is the one above, character for character).
constexpr int UPCAT_ROWS = 4;
struct UpCatEntry { const uint4* src; int c8, cg; };              // the piece (or the tail: c8 < 0, -c8 channel groups) and the group within it
__global__ __launch_bounds__(256) void upsample_cat_rows_kernel(UpCatArgs a, uint4* __restrict__ y, int Hi, int Wi, int Ho, int Wo,
                                                                float sh, float sw, int align, int yC8, int B) {
    __shared__ UpCatEntry tab[256];
    // XCD-aware order (workgroup i runs on XCD i % 8): XCD x takes the contiguous range [x Q, (x + 1) Q) of the (image, row block, column
    // block) items, so the row blocks that read the same source rows sit behind the same L2 (in plain order every XCD fetched every
    // source row: 2.2 x the algorithmic reads from HBM)
    const int gx = (Wo * yC8 + 255) / 256, gy = (Ho + UPCAT_ROWS - 1) / UPCAT_ROWS, nwork = gx * gy * B;
    const int Q = (nwork + 7) / 8, L = (blockIdx.x & 7) * Q + (blockIdx.x >> 3);
    if (L >= nwork || (int)(blockIdx.x >> 3) >= Q) return;
    const int bx = L % gx, by = (L / gx) % gy, b = L / (gx * gy);
    if ((int)threadIdx.x < yC8) {
        const int cgo = threadIdx.x;
        UpCatEntry e;
        if (cgo >= a.first8[a.n]) { e.src = a.tail; e.c8 = -a.tail8; e.cg = cgo - a.first8[a.n]; }
        else {
            const uint4* sp = a.src[0];
            int C8 = a.c8[0], f8 = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q)
                if (q < a.n && cgo >= a.first8[q]) { sp = a.src[q]; C8 = a.c8[q]; f8 = a.first8[q]; }
            e.src = sp; e.c8 = C8; e.cg = cgo - f8;
        }
        tab[cgo] = e;
    }
    __syncthreads();
    const unsigned item = bx * 256u + threadIdx.x;
    if (item >= (unsigned)(Wo * yC8)) return;
    const unsigned ox = item / (unsigned)yC8, cgo = item - ox * (unsigned)yC8;
    const UpCatEntry e = tab[cgo];
    // UPCAT_ROWS consecutive output rows per workgroup: a 2x up-sampling reads every source row from about four output rows, and rows
    // handled by different workgroups (different CUs) fetch it from the L2 each time -- 64 bytes of L2 traffic per 16 bytes of output
    // made the kernel L2-bound; back to back in one workgroup the repeats hit the CU's L1
    for (int oy = by * UPCAT_ROWS; oy < min(Ho, (by + 1) * UPCAT_ROWS); ++oy) {
        const unsigned opix = (unsigned)((b * Ho + oy) * Wo) + ox;
        if (e.c8 < 0) {                                  // the tail: a copy
            y[(size_t)opix * yC8 + cgo] = e.src[(size_t)opix * (unsigned)(-e.c8) + e.cg];
            continue;
        }
        float fy, fx;
        if (align) {
            fy = sh * oy;
            fx = sw * (int)ox;
        } else {
            fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
            fx = fmaxf(sw * ((int)ox + 0.5f) - 0.5f, 0.0f);
        }
        const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
        const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
        const float ly = fy - y0, lx = fx - x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const unsigned C8 = e.c8, r0 = (unsigned)((b * Hi + y0) * Wi), r1 = (unsigned)((b * Hi + y1) * Wi);
        const uint4* base = e.src + e.cg;
        const uint4 p00 = base[(r0 + x0) * C8], p01 = base[(r0 + x1) * C8];
        const uint4 p10 = base[(r1 + x0) * C8], p11 = base[(r1 + x1) * C8];
        auto mix = [&](unsigned va, unsigned vb, unsigned vc, unsigned vd) {
            const float lo = hy * (hx * bf16_lo(va) + lx * bf16_lo(vb)) + ly * (hx * bf16_lo(vc) + lx * bf16_lo(vd));
            const float hi = hy * (hx * bf16_hi(va) + lx * bf16_hi(vb)) + ly * (hx * bf16_hi(vc) + lx * bf16_hi(vd));
            return pack_bf16(lo, hi);
        };
        uint4 o;
        o.x = mix(p00.x, p01.x, p10.x, p11.x);
        o.y = mix(p00.y, p01.y, p10.y, p11.y);
        o.z = mix(p00.z, p01.z, p10.z, p11.z);
        o.w = mix(p00.w, p01.w, p10.w, p11.w);
        y[(size_t)opix * yC8 + cgo] = o;
    }
}

// The stereo pair as the feature extractor's batch: x (B, H, W, 2c) channels-last bf16 holds the left image in channels [0, c) and the
// right one in [c, 2c) (Network/StereoNet7.py:95-97 runs both through one feature extractor); y (2B, H, W, 8): image b = left image b,
// image B + b = right image b, channels [c, 8) zero -- the 8-channel input islam_conv_nhwc_bf16_s2 stages for the 3 -> 32 first layer
// (Network/PSM/submodule.py:63), written by one launch instead of torch.cat + MIOpen's own layout pass.  One 16-byte store per pixel.
__global__ __launch_bounds__(256) void stack_pair_pad8_kernel(const unsigned short* __restrict__ x, uint4* __restrict__ y, int B, int c,
                                                              long long HW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long n = i / HW, p = i - n * HW;
    const int side = n >= B ? 1 : 0;
    const unsigned short* s = x + (((long long)(n - (side ? B : 0)) * HW + p) * 2 + side) * c;
    unsigned short v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < c ? s[k] : (unsigned short)0;
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    y[i] = o;
}

extern "C" int islam_stack_pair_pad8_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C2, int H, int W, void* stream) {
    if (!x || !y || B < 1 || C2 < 2 || (C2 & 1) || C2 > 16 || H < 1 || W < 1)
        return fail(ISLAM_EARG, "islam_stack_pair_pad8_nhwc_bf16: bad argument (B=%d, C2=%d, %dx%d)", B, C2, H, W);
    const long long HW = (long long)H * W, total = 2LL * B * HW;
    hipLaunchKernelGGL(stack_pair_pad8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<uint4*>(y), B, C2 / 2, HW, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// The stereo pair as it arrives -- two fp32 NCHW tensors (B, c, H, W), c <= 4 (Network/VONet.py:31-34: torch.cat((img0_norm, img0_r_norm), 1)
// feeds the stereo net) -- straight to the two bf16 channels-last tensors the execution copy reads: x6 (B, H, W, 2c) = the concatenated
// pair (what half_image_into_kernel samples for conv_c0's input) and xs (2B, H, W, 8) = islam_stack_pair_pad8_nhwc_bf16's stacked,
// zero-padded batch.  One pass (reads 2 x 4 c bytes, writes 4 c + 32 bytes per pixel) instead of torch.cat + a bf16 cast + a layout copy +
// the stacking kernel (four passes, 84 us per batch at B = 8, 448 x 640).  Same values: round-to-nearest-even like Tensor.to(bfloat16).
__global__ __launch_bounds__(256) void stereo_pair_prepare_kernel(const float* __restrict__ left, const float* __restrict__ right, unsigned* __restrict__ x6,
                                                                  uint4* __restrict__ xs, int B, int c, long long HW, long long total) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const long long b = p / HW, q = p - b * HW;
    float l[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < c) {
            l[k] = left[(b * c + k) * HW + q];
            r[k] = right[(b * c + k) * HW + q];
        }
    xs[p] = make_uint4(pack_bf16(l[0], l[1]), pack_bf16(l[2], l[3]), 0u, 0u);
    xs[(long long)B * HW + p] = make_uint4(pack_bf16(r[0], r[1]), pack_bf16(r[2], r[3]), 0u, 0u);
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {                    // (v[k] = l[k], v[c + k] = r[k] with compile-time register indices)
                if (j == k) v[j] = l[k];
                if (j == c + k) v[j] = r[k];
            }
        }
    unsigned* o = x6 + p * c;                                // 2c bf16 = c dwords per pixel
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < c) o[k] = pack_bf16(v[2 * k], v[2 * k + 1]);
}

extern "C" int islam_stereo_pair_prepare_f32(const float* left, const float* right, uint16_t* x6, uint16_t* xs, int B, int c, int H, int W, void* stream) {
    if (!left || !right || !x6 || !xs || B < 1 || c < 1 || c > 4 || H < 1 || W < 1)
        return fail(ISLAM_EARG, "islam_stereo_pair_prepare_f32: bad argument (B=%d, c=%d, %dx%d)", B, c, H, W);
    const long long HW = (long long)H * W, total = (long long)B * HW;
    hipLaunchKernelGGL(stereo_pair_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, left, right,
                       reinterpret_cast<unsigned*>(x6), reinterpret_cast<uint4*>(xs), B, c, HW, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// 2x2 / stride-2 max pooling (floor mode), optionally of relu(x): relu and max commute
__global__ __launch_bounds__(256) void maxpool2_nhwc_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int C8, int Hi, int Wi,
                                                                 int Ho, int Wo, int relu, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx4 ix = split4(i, total, C8, Wo, Ho);
    const int cg = ix.cg, ox = ix.ox, oy = ix.oy, b = ix.b;
    const uint4* base = x + (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox) * C8 + cg;
    const uint4 a = base[0], bq = base[C8], c = base[(size_t)Wi * C8], d = base[(size_t)Wi * C8 + C8];
    auto mx = [&](unsigned va, unsigned vb, unsigned vc, unsigned vd) {
        float lo = fmaxf(fmaxf(bf16_lo(va), bf16_lo(vb)), fmaxf(bf16_lo(vc), bf16_lo(vd)));
        float hi = fmaxf(fmaxf(bf16_hi(va), bf16_hi(vb)), fmaxf(bf16_hi(vc), bf16_hi(vd)));
        if (relu) { lo = fmaxf(lo, 0.0f); hi = fmaxf(hi, 0.0f); }
        return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);      // exact: the maximum IS one of the bf16 inputs
    };
    uint4 o;
    o.x = mx(a.x, bq.x, c.x, d.x);
    o.y = mx(a.y, bq.y, c.y, d.y);
    o.z = mx(a.z, bq.z, c.z, d.z);
    o.w = mx(a.w, bq.w, c.w, d.w);
    y[i] = o;
}

// k x k / stride-k average pooling (floor mode): fp32 accumulation in row-major order, one rounding at the end (as ATen's mean)
__global__ __launch_bounds__(256) void block_mean_nhwc_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int C8, int Hi, int Wi,
                                                                   int Ho, int Wo, int k, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx4 ix = split4(i, total, C8, Wo, Ho);
    const int cg = ix.cg, ox = ix.ox, oy = ix.oy, b = ix.b;
    const uint4* base = x + (((size_t)b * Hi + (size_t)k * oy) * Wi + (size_t)k * ox) * C8 + cg;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q) {
            const uint4 v = base[((size_t)r * Wi + q) * C8];
            acc[0] += bf16_lo(v.x); acc[1] += bf16_hi(v.x); acc[2] += bf16_lo(v.y); acc[3] += bf16_hi(v.y);
            acc[4] += bf16_lo(v.z); acc[5] += bf16_hi(v.z); acc[6] += bf16_lo(v.w); acc[7] += bf16_hi(v.w);
        }
    const float inv = 1.0f / (float)(k * k);
    uint4 o;
    o.x = pack_bf16(acc[0] * inv, acc[1] * inv);
    o.y = pack_bf16(acc[2] * inv, acc[3] * inv);
    o.z = pack_bf16(acc[4] * inv, acc[5] * inv);
    o.w = pack_bf16(acc[6] * inv, acc[7] * inv);
    y[i] = o;
}

}  // namespace

static int resize_bilinear_launch(const uint16_t* x, const uint16_t* add, uint16_t* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                  int align_corners, int ytot, int yoff, void* stream) {
    if (B < 1 || C < 8 || (C & 7) || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1)
        return fail(ISLAM_EARG, "islam_resize_bilinear_nhwc_bf16: bad shape (C=%d must be a multiple of 8)", C);
    if ((ytot & 7) || (yoff & 7) || yoff < 0 || yoff + C > ytot)
        return fail(ISLAM_EARG, "islam_resize_bilinear_nhwc_bf16: output slice %d+%d of %d channels (multiples of 8)", yoff, C, ytot);
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f;
        sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
    } else {
        sh = (float)Hi / (float)Ho;
        sw = (float)Wi / (float)Wo;
    }
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(resize_bilinear_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(y), C / 8, Hi, Wi, Ho, Wo, sh, sw, align_corners, total,
                       ytot / 8, yoff / 8, reinterpret_cast<const uint4*>(add));
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

extern "C" int islam_resize_bilinear_nhwc_bf16_into(const uint16_t* x, uint16_t* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                                    int align_corners, int ytot, int yoff, void* stream) {
    return resize_bilinear_launch(x, nullptr, y, B, C, Hi, Wi, Ho, Wo, align_corners, ytot, yoff, stream);
}

// srcs / chans: host arrays of n <= 8 device pointers / channel counts (multiples of 8); tail may be NULL (tailC = 0).
// y: (B, Ho, Wo, sum(chans) + tailC).
extern "C" int islam_upsample_cat_nhwc_bf16(const uint16_t* const* srcs, const int* chans, int n, const uint16_t* tail, int tailC, uint16_t* y,
                                            int B, int Hi, int Wi, int Ho, int Wo, int align_corners, void* stream) {
    if (!srcs || !chans || !y || n < 1 || n > 8 || B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || tailC < 0 || (tailC & 7) || (tailC > 0 && !tail))
        return fail(ISLAM_EARG, "islam_upsample_cat_nhwc_bf16: bad argument (n=%d, tailC=%d, %dx%d -> %dx%d)", n, tailC, Hi, Wi, Ho, Wo);
    UpCatArgs a{};
    a.n = n;
    int off = 0;
    for (int k = 0; k < n; ++k) {
        if (!srcs[k] || chans[k] < 8 || (chans[k] & 7)) return fail(ISLAM_EARG, "islam_upsample_cat_nhwc_bf16: piece %d has %d channels", k, chans[k]);
        a.src[k] = reinterpret_cast<const uint4*>(srcs[k]);
        a.c8[k] = chans[k] / 8;
        a.first8[k] = off;
        off += chans[k] / 8;
    }
    for (int k = n; k <= 8; ++k) a.first8[k] = off;
    a.first8[n] = off;
    a.tail = reinterpret_cast<const uint4*>(tail);
    a.tail8 = tailC / 8;
    const int yC8 = off + tailC / 8;
    float sh, sw;
    if (align_corners) {
        sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f;
        sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
    } else {
        sh = (float)Hi / (float)Ho;
        sw = (float)Wi / (float)Wo;
    }
    const long long total = (long long)B * Ho * Wo * yC8;
    int maxc8 = std::max(a.tail8, 1);
    for (int k = 0; k < n; ++k) maxc8 = std::max(maxc8, a.c8[k]);
    static const bool rows_off = [] { const char* e = std::getenv("ISLAM_UPCAT_ROWS"); return e && e[0] == '0'; }();      // (A/B runs)
    if (!rows_off && yC8 <= 256 && Ho <= 65535 && B <= 65535 && total < (1LL << 31) && (long long)B * std::max(Hi, Ho) * std::max(Wi, Wo) * maxc8 < (1LL << 31)) {
        const long long nwork = (((long long)Wo * yC8 + 255) / 256) * ((Ho + UPCAT_ROWS - 1) / UPCAT_ROWS) * B;
        dim3 grid((unsigned)(8 * ((nwork + 7) / 8)));
        hipLaunchKernelGGL(upsample_cat_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, reinterpret_cast<uint4*>(y), Hi, Wi, Ho, Wo, sh, sw,
                           align_corners, yC8, B);
    } else
        hipLaunchKernelGGL(upsample_cat_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                           reinterpret_cast<uint4*>(y), Hi, Wi, Ho, Wo, sh, sw, align_corners, total, yC8);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// y = add + resize(x): the hourglass's `up1 + up2(low3)` (Network/PSM/hourglass.py:60-69) in one pass; add, y: (B,Ho,Wo,C)
extern "C" int islam_resize_bilinear_add_nhwc_bf16(const uint16_t* x, const uint16_t* add, uint16_t* y, int B, int C, int Hi, int Wi,
                                                   int Ho, int Wo, int align_corners, void* stream) {
    if (!add) return fail(ISLAM_EARG, "islam_resize_bilinear_add_nhwc_bf16: null addend");
    return resize_bilinear_launch(x, add, y, B, C, Hi, Wi, Ho, Wo, align_corners, C, 0, stream);
}

// the same into channels [yoff, yoff + C) of a (B,Ho,Wo,ytot) tensor: the decoder's torch.cat((conv_c8(...), skip), 1) etc.
// (StereoNet7.py:129-138) without copying the hourglass's half of the concatenation
extern "C" int islam_resize_bilinear_add_nhwc_bf16_into(const uint16_t* x, const uint16_t* add, uint16_t* y, int B, int C, int Hi, int Wi,
                                                        int Ho, int Wo, int align_corners, int ytot, int yoff, void* stream) {
    if (!add) return fail(ISLAM_EARG, "islam_resize_bilinear_add_nhwc_bf16_into: null addend");
    return resize_bilinear_launch(x, add, y, B, C, Hi, Wi, Ho, Wo, align_corners, ytot, yoff, stream);
}

// MaxPool2d(2, 2) / F.max_pool2d(kernel_size=2) of a channels-last bf16 tensor, optionally of relu(x) (StereoNet7.py:117-125
// `pool(act(conv(x)))`, hourglass.py:52 pool1); (B,H,W,C) -> (B,H/2,W/2,C)
extern "C" int islam_maxpool2_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C, int H, int W, int relu, void* stream) {
    if (B < 1 || C < 8 || (C & 7) || H < 2 || W < 2) return fail(ISLAM_EARG, "islam_maxpool2_nhwc_bf16: bad shape (C=%d, %dx%d)", C, H, W);
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(maxpool2_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(y), C / 8, H, W, Ho, Wo, relu, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// F.interpolate(x, scale_factor=0.5, mode='bilinear') of a channels-last bf16 image with C <= 8 channels (Network/StereoNet7.py:101: the
// half-resolution stereo pair that joins the features in front of conv_c0), written into channels [yoff, yoff + 8) of a (B,H/2,W/2,ytot)
// bf16 tensor: the C values and 8 - C zeros as ONE 16-byte store per pixel (the padded tail of the concatenation buffer).  With
// align_corners = False and scale 1/2 the source coordinate of output pixel o is 2 o + 1/2: the mean of a 2 x 2 block, evaluated as
// ATen does (0.5 (0.5 a + 0.5 b) + 0.5 (0.5 c + 0.5 d) in fp32, one rounding to bf16).  H, W even; C even.
__global__ __launch_bounds__(256) void half_image_into_kernel(const unsigned* __restrict__ x, uint4* __restrict__ y, int C2, int Hi, int Wi,
                                                              int ytot8, int yoff8, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int Wo = Wi / 2, Ho = Hi / 2;
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), b = (int)(i / ((long long)Wo * Ho));
    const unsigned* p00 = x + (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox) * C2;
    const unsigned* p10 = p00 + (size_t)Wi * C2;
    unsigned o[4] = {0u, 0u, 0u, 0u};
    for (int c = 0; c < C2; ++c) {
        const unsigned a = p00[c], bq = p00[C2 + c], cq = p10[c], d = p10[C2 + c];
        const float lo = 0.5f * (0.5f * bf16_lo(a) + 0.5f * bf16_lo(bq)) + 0.5f * (0.5f * bf16_lo(cq) + 0.5f * bf16_lo(d));
        const float hi = 0.5f * (0.5f * bf16_hi(a) + 0.5f * bf16_hi(bq)) + 0.5f * (0.5f * bf16_hi(cq) + 0.5f * bf16_hi(d));
        o[c] = pack_bf16(lo, hi);
    }
    y[(size_t)i * ytot8 + yoff8] = make_uint4(o[0], o[1], o[2], o[3]);
}

extern "C" int islam_half_image_into_nhwc_bf16(const uint16_t* x, uint16_t* y, int ytot, int yoff, int B, int C, int H, int W, void* stream) {
    if (B < 1 || C < 2 || C > 8 || (C & 1) || H < 2 || W < 2 || (H & 1) || (W & 1))
        return fail(ISLAM_EARG, "islam_half_image_into_nhwc_bf16: bad shape (C=%d even and <= 8, %dx%d even)", C, H, W);
    if ((ytot & 7) || (yoff & 7) || yoff < 0 || yoff + 8 > ytot) return fail(ISLAM_EARG, "islam_half_image_into_nhwc_bf16: output slot %d+8 of %d", yoff, ytot);
    const long long total = (long long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(half_image_into_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned*>(x), reinterpret_cast<uint4*>(y), C / 2, H, W, ytot / 8, yoff / 8, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// AvgPool2d((k,k), stride=(k,k)) of a channels-last bf16 tensor (the SPP branches, submodule.py:103-122); (B,H,W,C) -> (B,H/k,W/k,C)
extern "C" int islam_avgpool_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C, int H, int W, int k, void* stream) {
    if (B < 1 || C < 8 || (C & 7) || k < 1 || H < k || W < k) return fail(ISLAM_EARG, "islam_avgpool_nhwc_bf16: bad shape (C=%d, %dx%d, k=%d)", C, H, W, k);
    const int Ho = H / k, Wo = W / k;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(block_mean_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(y), C / 8, H, W, Ho, Wo, k, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

extern "C" int islam_resize_bilinear_nhwc_bf16(const uint16_t* x, uint16_t* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                               int align_corners, void* stream) {
    return islam_resize_bilinear_nhwc_bf16_into(x, y, B, C, Hi, Wi, Ho, Wo, align_corners, C, 0, stream);
}

// ------------------------------------------------------------------------------------------
// Epilogue of a bias-free MIOpen convolution of the frozen stereo net's execution copy, in place on channels-last bf16:
//   y <- act( bf16(y + bias[c]) [+ res] )      act = ReLU or identity
// One pass instead of ATen's broadcast add + clamp (+ residual add) kernels (Network/PSM/hourglass.py:6-40 Residual,
// StereoNet7.py:100-146).  Same roundings as the separate ops: bf16 after the bias, bf16 after the residual.
namespace {

__global__ __launch_bounds__(256) void bias_act_add_nhwc_bf16_kernel(uint4* __restrict__ y, const float* __restrict__ bias,
                                                                     const uint4* __restrict__ res, int C8, int relu, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c0 = octet_of(i, total, C8) * 8;
    const uint4 v = y[i];
    uint4 r = make_uint4(0, 0, 0, 0);
    if (res) r = res[i];
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
    auto one = [&](unsigned yv, unsigned rv, float blo, float bhi) {
        unsigned t = pack_bf16(bf16_lo(yv) + blo, bf16_hi(yv) + bhi);
        if (res) t = pack_bf16(bf16_lo(t) + bf16_lo(rv), bf16_hi(t) + bf16_hi(rv));
        if (relu) {
            if (t & 0x8000u) t &= 0xffff0000u;          // negative low half -> +0
            if (t & 0x80000000u) t &= 0x0000ffffu;      // negative high half -> +0
        }
        return t;
    };
    uint4 o;
    o.x = one(v.x, r.x, b0.x, b0.y);
    o.y = one(v.y, r.y, b0.z, b0.w);
    o.z = one(v.z, r.z, b1.x, b1.y);
    o.w = one(v.w, r.w, b1.z, b1.w);
    y[i] = o;
}

}  // namespace

extern "C" int islam_bias_act_add_nhwc_bf16(uint16_t* y, const float* bias, const uint16_t* res, long long pixels, int C, int relu,
                                            void* stream) {
    if (pixels < 1 || C < 8 || (C & 7)) return fail(ISLAM_EARG, "islam_bias_act_add_nhwc_bf16: C=%d must be a multiple of 8", C);
    const long long total = pixels * (C / 8);
    hipLaunchKernelGGL(bias_act_add_nhwc_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<uint4*>(y), bias, reinterpret_cast<const uint4*>(res), C / 8, relu, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// ------------------------------------------------------------------------------------------
// Train-mode BatchNorm2d (+ ReLU, + residual add) on a channels-last bf16 tensor, for the "frozen" stereo feature extractor
// that the reference still runs with batch statistics (TartanVO.py:90-91; Network/PSM/submodule.py:10-43):
//   y = act( bf16(x * scale[c] + shift[c]) [+ res] ),  scale = w * rsqrt(var_b + eps),  shift = b - mean * scale,
// mean / var_b (biased) over all pixels; running_mean / running_var (unbiased) / num_batches_tracked updated like
// nn.BatchNorm2d.  Three launches (MIOpen: three for the normalisation alone, ATen two more for ReLU and the residual add):
// per-workgroup partial sums -> one workgroup finalises in a fixed order (deterministic, double) -> one apply pass.
namespace {

constexpr int BN_BLOCKS = 256;

__global__ __launch_bounds__(256) void bn_partial_kernel(const uint4* __restrict__ x, int C8, long long pixels,
                                                         float* __restrict__ partial) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x;
    const int cg = tid % C8, pl = tid / C8, ppb = 256 / C8;         // C8 divides 256 (C = 8, 16, 32, 64, 128 ...)
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.0f; q[i] = 0.0f; }
    if (pl < ppb)
        for (long long p = (long long)blockIdx.x * ppb + pl; p < pixels; p += (long long)gridDim.x * ppb) {
            const uint4 v = x[p * C8 + cg];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = bf16_lo(w[i]), b = bf16_hi(w[i]);
                s[2 * i] += a; q[2 * i] = fmaf(a, a, q[2 * i]);
                s[2 * i + 1] += b; q[2 * i + 1] = fmaf(b, b, q[2 * i + 1]);
            }
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[tid][i] = s[i]; red[tid][8 + i] = q[i]; }
    __syncthreads();
    // thread (cg, j<16) adds the ppb pixel lanes of its channel group in lane order
    const int C = C8 * 8;
    for (int o = tid; o < C8 * 16; o += 256) {
        const int g = o / 16, j = o - g * 16;
        float acc = 0.0f;
        for (int l = 0; l < ppb; ++l) acc += red[l * C8 + g][j];
        const int c = g * 8 + (j & 7);
        partial[((size_t)blockIdx.x * 2 + (j >> 3)) * C + c] = acc;
    }
}

constexpr int FIN_THREADS = 1024;     // bn_finalize_kernel: one workgroup; 1024 / C slices of the partial blocks, each a short chain of loads

__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                          const float* __restrict__ weight, const float* __restrict__ bias,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ num_batches, double momentum, double eps,
                                                          float* __restrict__ scale_shift) {
    // FIN_THREADS threads = (FIN_THREADS / C) slices of the partial blocks x C channels (C <= 256); slices are combined in slice
    // order (deterministic).  With 256 threads a 32-channel layer walked 32 blocks per slice: 9 us for a 16 KB reduction.
    __shared__ double ls[FIN_THREADS], lq[FIN_THREADS];
    const int nsl = FIN_THREADS / C, c = threadIdx.x % C, sl = threadIdx.x / C;
    double s = 0.0, q = 0.0;
    if (sl < nsl) {
        const int per = (nblk + nsl - 1) / nsl, b0 = sl * per, b1 = min(nblk, b0 + per);
        int b = b0;
        for (; b + 4 <= b1; b += 4) {                     // four independent loads in flight per accumulator pair
            const float s0 = partial[((size_t)b * 2) * C + c], q0 = partial[((size_t)b * 2 + 1) * C + c];
            const float s1 = partial[((size_t)b * 2 + 2) * C + c], q1 = partial[((size_t)b * 2 + 3) * C + c];
            const float s2 = partial[((size_t)b * 2 + 4) * C + c], q2 = partial[((size_t)b * 2 + 5) * C + c];
            const float s3 = partial[((size_t)b * 2 + 6) * C + c], q3 = partial[((size_t)b * 2 + 7) * C + c];
            s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
        }
        for (; b < b1; ++b) {
            s += (double)partial[((size_t)b * 2) * C + c];
            q += (double)partial[((size_t)b * 2 + 1) * C + c];
        }
    }
    ls[threadIdx.x] = s;
    lq[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < C) {
        s = 0.0;
        q = 0.0;
        for (int j = 0; j < nsl; ++j) { s += ls[j * C + c]; q += lq[j * C + c]; }
        const double mean = s / count;
        const double var = fmax(q / count - mean * mean, 0.0);
        const double sc = (double)weight[c] / sqrt(var + eps);
        scale_shift[c] = (float)sc;
        scale_shift[C + c] = (float)((double)bias[c] - mean * sc);
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
        }
    }
    if (threadIdx.x == 0 && num_batches) *num_batches += 1;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, const uint4* __restrict__ res,
                                                       const float* __restrict__ scale_shift, int C8, int relu, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int C = C8 * 8, c0 = octet_of(i, total, C8) * 8;
    const uint4 v = x[i];
    uint4 r = make_uint4(0, 0, 0, 0);
    if (res) r = res[i];
    const float4 s0 = *reinterpret_cast<const float4*>(scale_shift + c0), s1 = *reinterpret_cast<const float4*>(scale_shift + c0 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(scale_shift + C + c0), h1 = *reinterpret_cast<const float4*>(scale_shift + C + c0 + 4);
    auto one = [&](unsigned xv, unsigned rv, float slo, float shi, float hlo, float hhi) {
        unsigned t = pack_bf16(fmaf(bf16_lo(xv), slo, hlo), fmaf(bf16_hi(xv), shi, hhi));
        if (res) t = pack_bf16(bf16_lo(t) + bf16_lo(rv), bf16_hi(t) + bf16_hi(rv));
        if (relu) {
            if (t & 0x8000u) t &= 0xffff0000u;
            if (t & 0x80000000u) t &= 0x0000ffffu;
        }
        return t;
    };
    uint4 o;
    o.x = one(v.x, r.x, s0.x, s0.y, h0.x, h0.y);
    o.y = one(v.y, r.y, s0.z, s0.w, h0.z, h0.w);
    o.z = one(v.z, r.z, s1.x, s1.y, h1.x, h1.y);
    o.w = one(v.w, r.w, s1.z, s1.w, h1.z, h1.w);
    y[i] = o;
}

}  // namespace

extern "C" size_t islam_bn_scratch_floats(int C) { return (size_t)BN_BLOCKS * 2 * C + 2 * (size_t)C; }

extern "C" int islam_bn_train_nhwc_bf16(const uint16_t* x, uint16_t* y, const uint16_t* res, const float* weight, const float* bias,
                                        float* running_mean, float* running_var, long long* num_batches_tracked, double momentum,
                                        double eps, int relu, long long pixels, int C, float* scratch, void* stream) {
    if (pixels < 1 || C < 8 || C > 256 || (C & 7) || 256 % (C / 8) != 0)      // bn_finalize_kernel: one 256-thread block over C
        return fail(ISLAM_EARG, "islam_bn_train_nhwc_bf16: C=%d must be 8 * a divisor of 256 and at most 256", C);
    hipStream_t s = (hipStream_t)stream;
    const int C8 = C / 8, ppb = 256 / C8;
    const int nblk = (int)std::min<long long>(BN_BLOCKS, (pixels + ppb - 1) / ppb);
    float* partial = scratch;
    float* scale_shift = scratch + (size_t)BN_BLOCKS * 2 * C;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const uint4*>(x), C8, pixels, partial);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, s, partial, nblk, C, (double)pixels, weight, bias, running_mean,
                       running_var, num_batches_tracked, momentum, eps, scale_shift);
    const long long total = pixels * C8;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint4*>(x),
                       reinterpret_cast<uint4*>(y), reinterpret_cast<const uint4*>(res), scale_shift, C8, relu, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// The two halves of islam_bn_train_nhwc_bf16 on their own, for convolutions that deliver the batch statistics themselves
// (islam_conv_nhwc_bf16 with `stats`): finalize = fixed-order sum of the folded partials -> scale / shift (+ running statistics);
// apply = one pass  y = act( bf16(x * scale[c] + shift[c]) [+ res] ).
extern "C" int islam_bn_finalize(const float* folded, double count, const float* weight, const float* bias, float* running_mean,
                                 float* running_var, long long* num_batches_tracked, double momentum, double eps, int C,
                                 float* scale_shift, void* stream) {
    if (count < 1 || C < 1 || C > 256) return fail(ISLAM_EARG, "islam_bn_finalize: C=%d (1..256)", C);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, (hipStream_t)stream, folded, BN_BLOCKS, C, count, weight, bias,
                       running_mean, running_var, num_batches_tracked, momentum, eps, scale_shift);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

extern "C" int islam_bn_apply_nhwc_bf16(const uint16_t* x, uint16_t* y, const uint16_t* res, const float* scale_shift, int relu,
                                        long long pixels, int C, void* stream) {
    if (pixels < 1 || C < 8 || (C & 7)) return fail(ISLAM_EARG, "islam_bn_apply_nhwc_bf16: C=%d must be a multiple of 8", C);
    const long long total = pixels * (C / 8);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(y), reinterpret_cast<const uint4*>(res), scale_shift,
                       C / 8, relu, total);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}
