// conv_nhwc.hip -- 3x3 / 1x1 stride-1 convolution of the frozen stereo network on the CDNA4 matrix cores, channels-last bf16.
//
// Replaces, for the bf16 execution copy of the stereo feature extractor, what the reference runs through cuDNN
// (Network/PSM/submodule.py:10-13 `convbn` = Conv2d(bias=False) + BatchNorm2d, :66-155 BasicBlock / feature_extraction) and
// what round 1 ran through MIOpen:
//   y[b,ho,wo,n] = epi( sum_{r,s,c} w[n,c,r,s] * pre( x[b, ho + r - P, wo + s - P, c] ) ),   P = K/2, zero padding
//   pre(v)  = v                      or   relu(bf16(v * in_scale[c] + in_shift[c]))   -- the PREVIOUS layer's train-mode
//             BatchNorm + ReLU applied while the tile is staged: that layer's normalised tensor is never written
//   epi(a)  = bf16(a)  [+ per-workgroup partial sums of bf16(a), bf16(a)^2 per output channel: the batch statistics of THIS
//             layer's BatchNorm come out of the convolution's epilogue instead of another pass over the tensor]
//          or  act( bf16( bf16(a + bias[n]) [+ res] ) )
// Why not MIOpen: (1) several of its bf16 kernels for these shapes convert the fp32 accumulator to bf16 by TRUNCATION
// (scripts/calib/bf16_rounding_probe.py: half of the outputs differ from round-to-nearest-even, all towards zero), a
// systematic -0.28 % per layer that the reference-generated golden vectors expose as a 5-6 % bias of the disparity; this
// kernel rounds to nearest even; (2) the BatchNorm that follows every convolution of the feature extractor costs two more
// passes over the activation (statistics, apply), the convolution itself being memory-bound at 32 channels.
//
// GEMM view per image: M = Cout (A operand: weights), N = pixels (B operand: im2col of x), K = K*K*Cin, on
// v_mfma_f32_32x32x16_bf16.  Workgroup = 256 threads = 4 waves, tile 32 x 16 output pixels x TN output channels, wave w
// owns pixel rows 4w .. 4w+3.  K is walked in chunks of 32 input channels: the chunk's halo tile ([y][x][32 ch], 80-byte
// pixel stride) and the K*K weight taps are staged in LDS once and feed K*K*2*4*TN/32 MFMAs per wave; activations are
// read from HBM as 16-byte vectors along C (a pixel's 32 channels = one 64-byte line) while the previous chunk is being
// multiplied.  The output tile is transposed through LDS so that the stores are 16 bytes per lane along C.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "../../include/islam_hip.h"
#include "common.h"
#include "conv_ws.h"

// scripts/conv_probe.sh builds two experiment variants of this file (never the product library): ISLAM_CONV_PROBE=1 skips the
// multiply phase (what the staging pipeline costs on its own), =2 skips fetch + staging (what the multiply phase costs on its own)
#ifndef ISLAM_CONV_PROBE
#define ISLAM_CONV_PROBE 0
#endif

// ISLAM_CONV_PROBE=3 (scripts/conv_phase_probe.sh): phase timestamps of one workgroup's wave 0
#if ISLAM_CONV_PROBE == 3
__device__ long long islam_conv_probe_buf[128];
#define CPROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if (cprb) islam_conv_probe_buf[(slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPROBE(slot) do { } while (0)
#endif

namespace {

using namespace islam;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // a REAL vector value: copies of HIP's uint4 struct become memcpy
                                                                 // intrinsics that keep a register array in scratch

constexpr int TW = 32;                               // (KC: input channels per staged chunk, template parameter; PS = KC + 8: bf16
                                                     // elements per LDS pixel / weight row -- 80- or 48-byte stride, conflict-free 16-byte reads)
constexpr int THREADS = 256;                         // 4 waves x ROWS pixel rows: tile 32 x 4*ROWS pixels

// Compile-time loop: the index is an integral_constant, so the register arrays it indexes are split into scalars as soon as the
// lambda is inlined.  With `#pragma unroll` the indices only become constant after the (late) unroll pass; an array that is
// still indexed dynamically when SROA runs is left to AMDGPUPromoteAlloca, which keeps anything above 128 bytes in SCRATCH -- the
// prefetched weights (nine uint4) then go through scratch_store right after their global loads, which forces a vmcnt wait
// in front of the multiply phase and serialises the two (found in the ISA; 194 -> see DESIGN.md).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ unsigned pack2(float a, float b) {      // round-to-nearest-even, one v_cvt_pk_bf16_f32
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo16(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi16(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
// ReLU of packed bf16: a negative bf16 is a negative int16 (sign-magnitude), so max(., 0) on the int16 lanes is the ReLU: ONE v_pk_max_i16 per pair
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2(unsigned t) {
    const s16x2_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, t), z));
}

// Register blocking: a wave owns ROWS pixel rows x TN channels; for a horizontal tap offset s it reads the ROWS + K - 1 halo rows
// once and reuses each for up to K vertical taps, and a weight operand for all its rows.  ROWS = 4 with TN = 32 (0.75 LDS
// operand reads per MFMA, two workgroups per CU), ROWS = 2 with TN = 64 (0.83; the 4-row variant needs 95 KB of LDS and
// leaves one wave per SIMD with nothing to hide its latencies behind: measured slower).
// Channel slices and the flow net's second output.  x: channels [xoff, xoff + Cin) of a (B,H,W,xs) tensor; y: channels
// [yoff, yoff + Cout) of a (B,H,W,ys) tensor (or nullptr: no channels-last output).  FLOW (template): the epilogue applies
// LeakyReLU(slope) to the fp32 accumulator + bias and ALSO stores it as fp32 NCHW into channels [coff, coff + Cout) of y32
// (B, ytot, H, W) -- PWC-Net's DenseNet buffer, whose non-convolution consumers (correlation, warp, transposed convolutions,
// flow heads) read fp32 NCHW, while the next convolution reads the bf16 channels-last mirror this kernel writes beside it.
// Dilation d > 1 (the flow net's context layers): a 3x3 convolution with dilation d is d*d independent dense 3x3 convolutions on the
// sub-grids (a::d, c::d) of the image, so the kernel runs on B*d*d "images" of (H, W) = (Hf/d, Wf/d) pixels whose neighbours are
// d pixels apart in memory (Hf, Wf must be multiples of d); Wf = full row length in pixels.
// Transposed convolution (tc = 1, KS = 2: ConvTranspose2d(k = 4, s = 2, p = 1), the stereo net's decoder): output pixel (2y + a, 2x + c)
// depends on the 2x2 input patch rows {y - 1 + a, y + a} x columns {x - 1 + c, x + c} only, so the layer is FOUR dense 2x2
// convolutions, one per output parity class (a, c), each with its own taps K[r][s] = W[:, :, 3 - 2r - a, 3 - 2s - c].  The kernel runs
// on B * 4 "images" (d = 2 decodes the class like a dilation sub-grid): the INPUT is the dense image with its origin shifted by
// (a, c), the OUTPUT is the sub-grid (a::2, c::2) of the (2H, 2W) result, the weights are the class's slice of the packed array.
struct Slices { int xs, xoff, ys, yoff; float* y32; int ytot, coff; float slope; int d, Wf; int tc; int Hi, Wi; };      // (Hi, Wi: the input's size when it differs from the output's -- stride 2)
// TWO (template): the input is the concatenation of TWO dense tensors along the channels -- channels [0, split) from x (xs = split), channels
// [split, Cin) from x2 (pixel stride xs2) -- read where they lie: the decoder's torch.cat((hourglass output, skip tensor), 1) in front of
// every transposed convolution (Network/StereoNet7.py:121-138) is never materialised.  split is a multiple of the chunk size KC, so a chunk
// comes from one tensor.
struct Src2 { const unsigned short* x2; int xs2, split; };

template <int TN, int KS, int ROWS, int KC, bool FLOW, int S = 1, bool TWO = false>
__global__ __launch_bounds__(THREADS, (ROWS == 4 && TN == 64) ? 2 : 1) void conv_nhwc_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wp,
                                                             const float* __restrict__ in_affine, const float* __restrict__ bias,
                                                             const unsigned short* __restrict__ res, unsigned short* __restrict__ y,
                                                             float* __restrict__ partial, int Cin, int CinP, int H, int W, int Cout,
                                                             int CoutP, int relu, int tiles_x, int tiles, int in_relu, Slices sl, int nimg,
                                                             int nblk_n, Src2 s2) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    // (S = 2: Conv2d(stride = 2, padding = KS / 2) -- Network/PSM/submodule.py:76-85 layer2's first block and its 1x1 downsample, the
    //  quarter-resolution tail of StereoNet7: output pixel (y, x) reads input rows 2 y - P ... 2 y - P + KS - 1; the halo tile covers
    //  (TH - 1) S + 1 + 2 P rows, the B operand of output column li is input column li S + tap)
    constexpr int TH = 4 * ROWS, P = KS / 2, IH = (TH - 1) * S + 1 + 2 * P, IW = (TW - 1) * S + 1 + 2 * P, NPIX = IH * IW, TAPS = KS * KS;
    constexpr int PS = KC + 8, OPP = KC / 8;                 // LDS row stride (elements), 16-byte octets per pixel / weight row
    constexpr int NIN = (NPIX * (KC / 8) + THREADS - 1) / THREADS;       // 16-byte items of the halo tile per thread
    constexpr int NWT = (TAPS * TN * (KC / 8) + THREADS - 1) / THREADS;  // 16-byte items of the weight taps per thread
    constexpr int NT = TN / 32, NR = (ROWS - 1) * S + KS;
    unsigned short* lin = lds;                               // [IH][IW][PS]
    unsigned short* lw = lds + (size_t)NPIX * PS;            // [TAPS][TN][PS]
    constexpr int DUMMY = (NPIX + TAPS * TN) * PS;           // 16 bytes of scratch behind both tiles
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware work mapping (workgroup i runs on XCD i % 8, MI355X_MICROARCH.md): the launch is one-dimensional; logical work
    // item s = (pixel tile, channel block) with the channel block fastest, and XCD x takes the contiguous range [x Q, (x+1) Q) of
    // s -- so the channel blocks of a pixel tile (which read the SAME input tile) and vertically adjacent tiles (which share
    // halo rows) run at the same time on CUs behind the same L2 instead of re-reading the input from HBM.  Speed only.
    // (Measured, B = 16: 352->128 @224x320 1004 -> 948 us, 128->128 @112x160 111 -> 107 us, 64->128 1x1 32.5 -> 26.8 us; layers with ONE
    // channel block lose 3-7 % under the contiguous ranges, so they keep the plain order.)
    const int nwork = tiles * nimg * nblk_n, Q = (nwork + 7) / 8;
    const int sidx = nblk_n > 1 ? (blockIdx.x & 7) * Q + (blockIdx.x >> 3) : blockIdx.x;
    if (sidx >= nwork) return;
    const int wtile = sidx / nblk_n;                         // (image, pixel tile): the index of the per-workgroup partial sums
    const int tile = wtile % tiles, b = wtile / tiles;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wo0 = tx * TW, ho0 = ty * TH, n0 = (sidx - wtile * nblk_n) * TN;
    [[maybe_unused]] const bool cprb = tid == 0 && sidx == nwork / 2 + 3;
    [[maybe_unused]] int cpi = 2;
    CPROBE(0);
#if ISLAM_CONV_PROBE == 3
    if (cprb) islam_conv_probe_buf[124] = clock64();
#endif
    // image b = (batch index, sub-grid row a, sub-grid column c); pixel (gy, gx) of it is full-resolution pixel (gy*d + a, gx*d + c)
    const int dd = sl.d * sl.d, bb = b / dd, sga = (b - bb * dd) / sl.d, sgc = b - bb * dd - sga * sl.d;
    const size_t img0 = ((size_t)(bb * H * sl.d + sga)) * sl.Wf + sgc;          // first pixel of the image, in full-resolution pixels
    const bool tcm = sl.tc != 0;                             // transposed convolution: dense input image, output on a sub-grid
    const int Hin = S > 1 ? sl.Hi : H, Win = S > 1 ? sl.Wi : W;                 // (S > 1: dense input of its own size, d = 1, tc = 0)
    const int idil = tcm ? 1 : sl.d, iWf = S > 1 ? Win : (tcm ? W : sl.Wf), oWf = tcm ? 2 * W : sl.Wf;
    const int oy = tcm ? sga : 0, ox = tcm ? sgc : 0;        // origin shift of the 2x2 patch of parity class (sga, sgc)
    const size_t in_img0 = S > 1 ? (size_t)bb * Hin * Win : (tcm ? (size_t)bb * H * W : img0);
    const size_t out_img0 = tcm ? ((size_t)(bb * 2 * H + sga)) * (2 * W) + sgc : img0;
    if (tcm) wp += (size_t)(sga * 2 + sgc) * TAPS * CoutP * CinP;
    const unsigned short* xb = x + in_img0 * sl.xs + sl.xoff;
    [[maybe_unused]] const unsigned short* xb2 = TWO ? s2.x2 + in_img0 * s2.xs2 : nullptr;

    f32x16 acc[NT][ROWS];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int p = 0; p < ROWS; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][p][i] = 0.0f;

    // staging map, the same for every chunk: item -> (halo pixel, channel octet); outside the image or past Cin -> zero.
    // THREADS is a multiple of OPP, so a thread's items all carry the same channel octet.
    const int coct = 8 * (tid & (OPP - 1));
    int goff[NIN], loff[NIN];
    static_for<0, NIN>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int it = tid + k * THREADS;
        goff[k] = -1; loff[k] = DUMMY;
        if (it < NPIX * (KC / 8)) {
            const int pix = it / OPP;
            const int yy = pix / IW, xx = pix - yy * IW;
            const int gy = ho0 * S - P + oy + yy, gx = wo0 * S - P + ox + xx;
            loff[k] = pix * PS + coct;
            if (gy >= 0 && gy < Hin && gx >= 0 && gx < Win) goff[k] = TWO ? (gy * iWf + gx) * idil : (gy * iWf + gx) * idil * sl.xs + coct;      // (TWO: the pixel index)
        }
    });
    int woff[NWT], wlds[NWT];
    static_for<0, NWT>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int it = tid + k * THREADS;                    // ((tap*TN + n)*OPP + octet)
        woff[k] = 0; wlds[k] = DUMMY;
        if (it < TAPS * TN * (KC / 8)) {
            const int row = it / OPP, tap = row / TN, n = row - tap * TN;
            woff[k] = (tap * CoutP + n0 + n) * CinP + coct;
            wlds[k] = row * PS + coct;
        }
    });
    u32x4 pre[NIN];                                          // (native vectors: see u32x4)
    u32x4 prew[NWT];
    float4 as0 = {1, 1, 1, 1}, as1 = as0, ah0 = {0, 0, 0, 0}, ah1 = ah0;     // the chunk's BatchNorm scale / shift, fetched with it
    auto fetch = [&](int c0) {
        const bool cok = c0 + coct < Cin;
        if (in_affine && cok) {                              // previous layer's BatchNorm (batch statistics) + ReLU on load
            const float* sc = in_affine + c0 + coct;
            as0 = *reinterpret_cast<const float4*>(sc); as1 = *reinterpret_cast<const float4*>(sc + 4);
            ah0 = *reinterpret_cast<const float4*>(sc + Cin); ah1 = *reinterpret_cast<const float4*>(sc + Cin + 4);
        }
        static_for<0, NIN>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            if constexpr (TWO) {
                const bool second = c0 >= s2.split;                     // (uniform: the chunk lies in one tensor)
                const unsigned short* base = second ? xb2 : xb;
                const int xsc = second ? s2.xs2 : sl.xs, cc = (second ? c0 - s2.split : c0) + coct;
                pre[k] = *reinterpret_cast<const u32x4*>(base + ((cok && goff[k] >= 0) ? (size_t)goff[k] * xsc + cc : (size_t)0));
            } else
            pre[k] = *reinterpret_cast<const u32x4*>(xb + ((cok && goff[k] >= 0) ? (size_t)goff[k] + c0 : (size_t)0));   // masked: any valid address
        });
        static_for<0, NWT>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            prew[k] = *reinterpret_cast<const u32x4*>(wp + (size_t)woff[k] + c0);
        });
    };
    auto stage = [&](int c0) {
        const bool cok = c0 + coct < Cin;
        const float4 s0 = as0, s1 = as1, h0 = ah0, h1 = ah1;
        static_for<0, NIN>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            u32x4 v = pre[k];
            if (in_affine) {
                v.x = relu2(pack2(fmaf(lo16(v.x), s0.x, h0.x), fmaf(hi16(v.x), s0.y, h0.y)));
                v.y = relu2(pack2(fmaf(lo16(v.y), s0.z, h0.z), fmaf(hi16(v.y), s0.w, h0.w)));
                v.z = relu2(pack2(fmaf(lo16(v.z), s1.x, h1.x), fmaf(hi16(v.z), s1.y, h1.y)));
                v.w = relu2(pack2(fmaf(lo16(v.w), s1.z, h1.z), fmaf(hi16(v.w), s1.w, h1.w)));
            }
            else if (in_relu) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }     // ReLU of the input on load
            if (!(cok && goff[k] >= 0)) v = u32x4{0, 0, 0, 0};           // zero padding of the NORMALISED activation
            *reinterpret_cast<u32x4*>(lin + loff[k]) = v;
        });
        static_for<0, NWT>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            *reinterpret_cast<u32x4*>(lw + wlds[k]) = prew[k];
        });
    };

    const int kg = lane >> 5, li = lane & 31;
    const unsigned short* bbase = lin + ((size_t)(ROWS * wave * S) * IW + li * S) * PS + 8 * kg;
    const unsigned short* abase = lw + (size_t)li * PS + 8 * kg;
    fetch(0);
    CPROBE(1);
    for (int c0 = 0; c0 < CinP; c0 += KC) {
        CPROBE(cpi); ++cpi;
        __syncthreads();                                     // the previous chunk's operand reads are done
        CPROBE(cpi); ++cpi;
#if ISLAM_CONV_PROBE != 2
        stage(c0);
#endif
        CPROBE(cpi); ++cpi;
        __syncthreads();
        CPROBE(cpi); ++cpi;
#if ISLAM_CONV_PROBE != 2
        if (c0 + KC < CinP) fetch(c0 + KC);                  // in flight while this chunk is multiplied
#endif
#if ISLAM_CONV_PROBE == 1
        continue;
#endif
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 bf[NR];
#pragma unroll
                for (int q = 0; q < NR; ++q) bf[q] = *reinterpret_cast<const bf16x8*>(bbase + ((size_t)q * IW + s) * PS + 16 * ks);
#pragma unroll
                for (int r = 0; r < KS; ++r) {
                    bf16x8 af[NT];
#pragma unroll
                    for (int a = 0; a < NT; ++a)
                        af[a] = *reinterpret_cast<const bf16x8*>(abase + ((size_t)(r * KS + s) * TN + a * 32) * PS + 16 * ks);
#pragma unroll
                    for (int a = 0; a < NT; ++a)
#pragma unroll
                        for (int p = 0; p < ROWS; ++p)
                            acc[a][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[p * S + r], acc[a][p], 0, 0, 0);
                }
            }
        CPROBE(cpi); ++cpi;
    }
    CPROBE(120);

    // ---- epilogue.  D row (channel) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), D col (pixel x) = lane&31.  The tile goes through LDS
    // ([pixel][TN channels] bf16, rounded once) so that the stores are 16 bytes per lane along C -- consecutive lanes complete
    // whole 64 / 128-byte pixel rows -- and the same read feeds the residual add, the ReLU and the BatchNorm partial sums.
    constexpr int TS = TN + 8;                               // bf16 elements per pixel row of the staging tile (16-byte pad)
    constexpr int OCT = TN / 8, PPT = TW * TH * OCT / THREADS;     // channel octets per pixel, (pixel, octet) items per thread
    unsigned short* tl = lds;                                // [TW*TH][TS]
    float* red = reinterpret_cast<float*>(lds);              // [THREADS][17], over the staging tile once every thread has read its items
    f32x4 bv[NT * 4];                                        // the lane's bias values, all requested before the first one is used
    static_for<0, NT * 4>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const int nl = (i / 4) * 32 + 8 * (i % 4) + 4 * kg;
        bv[i] = (bias && n0 + nl < Cout) ? *reinterpret_cast<const f32x4*>(bias + n0 + nl) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    });
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int p = 0; p < ROWS; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * kg;
                float v0 = acc[a][p][4 * g], v1 = acc[a][p][4 * g + 1], v2 = acc[a][p][4 * g + 2], v3 = acc[a][p][4 * g + 3];
                if (bias) { const f32x4 bq = bv[a * 4 + g]; v0 += bq.x; v1 += bq.y; v2 += bq.z; v3 += bq.w; }
                if constexpr (FLOW) {
                    v0 = v0 >= 0.0f ? v0 : v0 * sl.slope; v1 = v1 >= 0.0f ? v1 : v1 * sl.slope;
                    v2 = v2 >= 0.0f ? v2 : v2 * sl.slope; v3 = v3 >= 0.0f ? v3 : v3 * sl.slope;
                    const int ho = ho0 + ROWS * wave + p, wo = wo0 + li, nn = n0 + nl;      // lanes = 32 consecutive pixels of a row:
                    if (sl.y32 && ho < H && wo < W && nn < Cout) {                          // 128-byte segments of a channel plane (d = 1)
                        const size_t plane = (size_t)H * sl.d * sl.Wf;
                        float* yp = sl.y32 + ((size_t)bb * sl.ytot + sl.coff + nn) * plane + ((size_t)(ho * sl.d + sga)) * sl.Wf + wo * sl.d + sgc;
                        yp[0] = v0; yp[plane] = v1; yp[2 * plane] = v2; yp[3 * plane] = v3;
                    }
                }
                *reinterpret_cast<uint2*>(tl + (size_t)((ROWS * wave + p) * TW + li) * TS + nl) = make_uint2(pack2(v0, v1), pack2(v2, v3));
            }
    if (FLOW && !y) return;                                  // (uniform) no channels-last mirror asked for
    __syncthreads();
    unsigned short* yb = y + out_img0 * sl.ys + sl.yoff;
    const unsigned short* rb = res ? res + (size_t)b * H * W * Cout : nullptr;
    const int oct = tid % OCT, n = n0 + 8 * oct;
    float sm[8], sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[i] = 0.0f; sq[i] = 0.0f; }
    u32x4 rv[PPT];                                           // the residual values of all the thread's items, requested at once
    if (rb)                                                  // (one load -> wait -> store per item is PPT dependent round trips)
        static_for<0, PPT>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int px = (tid + k * THREADS) / OCT;
            const int ho = ho0 + px / TW, wo = wo0 + (px % TW);
            const bool ok = ho < H && wo < W && n < Cout;
            rv[k] = *reinterpret_cast<const u32x4*>(rb + (ok ? ((size_t)ho * W + wo) * Cout + n : (size_t)0));
        });
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int px = (tid + k * THREADS) / OCT;            // THREADS is a multiple of OCT: the octet is the same for every k
        const int ho = ho0 + px / TW, wo = wo0 + (px % TW);
        if (ho >= H || wo >= W || n >= Cout) continue;
        uint4 v = *reinterpret_cast<const uint4*>(tl + (size_t)px * TS + 8 * oct);
        if (partial) {
            const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lo = lo16(wv[i]), hi = hi16(wv[i]);
                sm[2 * i] += lo; sq[2 * i] = fmaf(lo, lo, sq[2 * i]);
                sm[2 * i + 1] += hi; sq[2 * i + 1] = fmaf(hi, hi, sq[2 * i + 1]);
            }
        }
        const size_t o = ((size_t)ho * oWf + wo) * sl.d * sl.ys + n;
        if (rb) {
            const u32x4 r = rv[k];
            v.x = pack2(lo16(v.x) + lo16(r.x), hi16(v.x) + hi16(r.x));
            v.y = pack2(lo16(v.y) + lo16(r.y), hi16(v.y) + hi16(r.y));
            v.z = pack2(lo16(v.z) + lo16(r.z), hi16(v.z) + hi16(r.z));
            v.w = pack2(lo16(v.w) + lo16(r.w), hi16(v.w) + hi16(r.w));
        }
        if (relu) { v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w); }
        *reinterpret_cast<uint4*>(yb + o) = v;
    }
    if (partial) {
        // per-workgroup sums of the (bf16-rounded, as nn.BatchNorm2d sees them) outputs over the tile's valid pixels, per
        // channel, in a fixed order: every thread holds the sums of its octet over its pixels, one lane per (channel, moment)
        // adds the THREADS / OCT threads of that octet in thread order
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 17 + i] = sm[i]; red[tid * 17 + 8 + i] = sq[i]; }
        __syncthreads();
        if (tid < 2 * TN) {
            const int which = tid / TN, c = tid - which * TN, o2 = c >> 3, i = c & 7;
            float t = 0.0f;
            for (int m = 0; m < THREADS / OCT; ++m) t += red[(o2 + OCT * m) * 17 + 8 * which + i];
            if (n0 + c < Cout) partial[((size_t)wtile * 2 + which) * Cout + n0 + c] = t;
        }
    }
    CPROBE(121);
#if ISLAM_CONV_PROBE == 3
    if (cprb) islam_conv_probe_buf[125] = clock64();
#endif
}

// tile of 64 pixels x 32 channels: reads are 256-byte rows of a channel plane, writes 64 bytes per pixel (16 bytes per lane)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int stot, int soff, unsigned short* __restrict__ dst,
                                                           int dtot, int doff, int C, int Cpad, long long pixels) {
    __shared__ unsigned short t[64][40];                     // [pixel][32 channels + pad]: 80-byte rows
    const int b = blockIdx.z, c0 = blockIdx.y * 32;
    const long long p0 = (long long)blockIdx.x * 64;
    const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;  // 4 channel groups of 8
    const float* sp = src + ((size_t)b * stot + soff + c0) * pixels + p0 + px;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cq * 8 + j;
        const float v = (c0 + c < C && p0 + px < pixels) ? sp[(size_t)c * pixels] : 0.0f;
        t[px][c] = (unsigned short)(pack2(v, 0.0f) & 0xffffu);
    }
    __syncthreads();
    const int q = threadIdx.x >> 2, oct = threadIdx.x & 3;   // pixel, channel octet
    if (p0 + q < pixels && c0 + 8 * oct < Cpad)
        *reinterpret_cast<uint4*>(dst + ((size_t)b * pixels + p0 + q) * dtot + doff + c0 + 8 * oct) = *reinterpret_cast<const uint4*>(&t[q][8 * oct]);
}

constexpr int RED_BLOCKS = 256;         // = BN_BLOCKS of conv_mfma.hip: the layout bn_finalize_kernel reads

// partial[nblk][2][C] -> out[RED_BLOCKS][2][C], block j adds the slices j, j + RED_BLOCKS, ... in that order (deterministic)
__global__ __launch_bounds__(256) void partial_fold_kernel(const float* __restrict__ partial, int nblk, int C2, float* __restrict__ out) {
    const int j = blockIdx.x;
    for (int c = threadIdx.x; c < C2; c += 256) {
        float s = 0.0f;
        for (int bI = j; bI < nblk; bI += RED_BLOCKS) s += partial[(size_t)bI * C2 + c];
        out[(size_t)j * C2 + c] = s;
    }
}

// partial_fold_kernel and conv_mfma.hip's bn_finalize_kernel in ONE launch (islam_conv_nhwc_bf16_bn): every workgroup folds its
// share of the per-tile partial sums as partial_fold_kernel does (write-through stores), takes a ticket, and the workgroup that
// draws the last one turns the folded [RED_BLOCKS][2][C] sums into the BatchNorm's [scale | shift] and running statistics with
// bn_finalize_kernel's arithmetic in its order (1024 threads = 1024 / C slices of the folded blocks, slices combined in slice
// order): the same bits as the two launches, one ~4.5 us launch less behind each of the stereo net's ~45 convbn layers -- they
// sit on the critical path of the frozen nets' graph replay.  counter: one zero-initialised int, left at zero.
constexpr int FF_THREADS = 1024;
__device__ __forceinline__ float ldc_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(FF_THREADS) void fold_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ folded,
                                                                   double count, const float* __restrict__ weight, const float* __restrict__ bias,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   long long* __restrict__ num_batches, double momentum, double eps,
                                                                   float* __restrict__ scale_shift, int* __restrict__ counter) {
    __shared__ double ls[FF_THREADS], lq[FF_THREADS];
    __shared__ int s_last;
    const int j = blockIdx.x, C2 = 2 * C;
    for (int c = threadIdx.x; c < C2; c += FF_THREADS) {
        float s = 0.0f;
        for (int bI = j; bI < nblk; bI += RED_BLOCKS) s += partial[(size_t)bI * C2 + c];
        __hip_atomic_store(&folded[(size_t)j * C2 + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores have completed
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup: bn_finalize_kernel over folded[RED_BLOCKS][2][C], read past this XCD's L2
    const int nsl = FF_THREADS / C, c = threadIdx.x % C, sl = threadIdx.x / C;
    double s = 0.0, q = 0.0;
    if (sl < nsl) {
        const int per = (RED_BLOCKS + nsl - 1) / nsl, b0 = sl * per, b1 = min(RED_BLOCKS, b0 + per);
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            const float s0 = ldc_f32(&folded[((size_t)b * 2) * C + c]), q0 = ldc_f32(&folded[((size_t)b * 2 + 1) * C + c]);
            const float s1 = ldc_f32(&folded[((size_t)b * 2 + 2) * C + c]), q1 = ldc_f32(&folded[((size_t)b * 2 + 3) * C + c]);
            const float s2 = ldc_f32(&folded[((size_t)b * 2 + 4) * C + c]), q2 = ldc_f32(&folded[((size_t)b * 2 + 5) * C + c]);
            const float s3 = ldc_f32(&folded[((size_t)b * 2 + 6) * C + c]), q3 = ldc_f32(&folded[((size_t)b * 2 + 7) * C + c]);
            s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
        }
        for (; b < b1; ++b) {
            s += (double)ldc_f32(&folded[((size_t)b * 2) * C + c]);
            q += (double)ldc_f32(&folded[((size_t)b * 2 + 1) * C + c]);
        }
    }
    ls[threadIdx.x] = s;
    lq[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < C) {
        s = 0.0;
        q = 0.0;
        for (int k = 0; k < nsl; ++k) { s += ls[k * C + c]; q += lq[k * C + c]; }
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
    if (threadIdx.x == 0) {
        if (num_batches) *num_batches += 1;
        *counter = 0;
    }
}

// The persistent kernels write ONE row of partial sums per workgroup -- 249 rows (conv3x3_ws_kernel) or 448 (conv3x3_ws32_kernel) at the
// benched size, where the tile kernel writes 2240-4480: folding them to RED_BLOCKS rows costs a launch (~5 us on the critical path of
// the replay, 20 such layers per forward) to move 100-250 KB.  This is bn_finalize_kernel reading the UNFOLDED rows: the value of folded
// row b is formed on the fly exactly as partial_fold_kernel forms it (0.0f + row b + row b + 256, in float; rows past nblk do not
// exist = the zeros the fold would have written), then the same slices, the same groups of four, the same order: the same bits.
// nblk <= NF * RED_BLOCKS.
template <int NF>
__global__ __launch_bounds__(FF_THREADS) void finalize_rows_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                                   const float* __restrict__ weight, const float* __restrict__ bias,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   long long* __restrict__ num_batches, double momentum, double eps,
                                                                   float* __restrict__ scale_shift) {
    __shared__ double ls[FF_THREADS], lq[FF_THREADS];
    const int nsl = FF_THREADS / C, c = threadIdx.x % C, sl = threadIdx.x / C, C2 = 2 * C;
    auto row = [&](int b, int which) {                  // folded[b][which][c]
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int r = b + k * RED_BLOCKS;
            const float t = partial[(size_t)min(r, nblk - 1) * C2 + which * C + c];      // (always a load, then a select: a branch per
            v += r < nblk ? t : 0.0f;                                                    //  row would serialise the eight loads of a group)
        }
        return v;
    };
    double s = 0.0, q = 0.0;
    if (sl < nsl) {
        const int per = (RED_BLOCKS + nsl - 1) / nsl, b0 = sl * per, b1 = min(RED_BLOCKS, b0 + per);
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            const float s0 = row(b, 0), q0 = row(b, 1), s1 = row(b + 1, 0), q1 = row(b + 1, 1);
            const float s2 = row(b + 2, 0), q2 = row(b + 2, 1), s3 = row(b + 3, 0), q3 = row(b + 3, 1);
            s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
        }
        for (; b < b1; ++b) {
            s += (double)row(b, 0);
            q += (double)row(b, 1);
        }
    }
    ls[threadIdx.x] = s;
    lq[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < C) {
        s = 0.0;
        q = 0.0;
        for (int k = 0; k < nsl; ++k) { s += ls[k * C + c]; q += lq[k * C + c]; }
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

template <int TN, int KS, int ROWS, int KC, bool FLOW = false, int S = 1, bool TWO = false>
int launch(const unsigned short* x, const unsigned short* wp, const float* in_affine, const float* bias, const unsigned short* res,
           unsigned short* y, float* partial, int B, int Cin, int CinP, int H, int W, int Cout, int CoutP, int relu, int in_relu, hipStream_t s,
           Slices sl = Slices{0, 0, 0, 0, nullptr, 0, 0, 0.0f, 1, 0, 0, 0, 0}, Src2 s2 = Src2{nullptr, 0, 0}) {
    if (sl.xs == 0) { sl.xs = Cin; sl.ys = Cout; }           // dense tensors
    if (sl.Wf == 0) sl.Wf = W;
    constexpr int TH = 4 * ROWS, P = KS / 2, NPIX = ((TH - 1) * S + 1 + 2 * P) * ((TW - 1) * S + 1 + 2 * P), TAPS = KS * KS, PS = KC + 8;
    const size_t conv_lds = ((size_t)NPIX * PS + (size_t)TAPS * TN * PS + 8) * sizeof(unsigned short);
    const size_t epi_lds = std::max((size_t)TW * TH * (TN + 8) * sizeof(unsigned short), (size_t)THREADS * 17 * sizeof(float));
    const size_t lds = std::max(conv_lds, epi_lds);
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv_nhwc_kernel<TN, KS, ROWS, KC, FLOW, S, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set[dev] = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int nblk_n = (Cout + TN - 1) / TN, nwork = tiles_x * tiles_y * B * nblk_n;      // TH = tile_h(Cout): TN = 64 <=> Cout > 32
    dim3 grid(nblk_n > 1 ? 8 * ((nwork + 7) / 8) : nwork);
    hipLaunchKernelGGL((conv_nhwc_kernel<TN, KS, ROWS, KC, FLOW, S, TWO>), grid, dim3(THREADS), lds, s, x, wp, in_affine, bias, res, y, partial, Cin, CinP,
                       H, W, Cout, CoutP, relu, tiles_x, tiles_x * tiles_y, in_relu, sl, B, nblk_n, s2);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // namespace

extern "C" {

#if ISLAM_CONV_PROBE == 3
int islam_conv_probe_read(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(islam_conv_probe_buf), sizeof(long long) * 128) == hipSuccess ? 0 : 1;
}
#endif

int islam_conv_ws_mode(int mode) { return conv_ws_set_mode(mode); }
int islam_conv_ws_launch_counts(long long* out2) { if (!out2) return ISLAM_EARG; conv_ws_read_counts(out2); return ISLAM_OK; }

size_t islam_conv_nhwc_packed_elems(int Cin, int Cout, int ksize) {
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    return (size_t)ksize * ksize * CoutP * CinP;
}

// The 64-channel-block 3x3 kernel with FOUR pixel rows per wave and 16-channel chunks (conv_nhwc_kernel<64, 3, 4, 16>): tile 32 x 16,
// 0.5 instead of 0.83 LDS operand reads per MFMA, the same MFMAs per barrier pair, 47 KB of LDS and 244 VGPRs (two workgroups per
// CU as before).  Measured (B = 16, scripts/conv_nhwc_bench.py, alternating runs): 352->128 @224x320 1000 -> 910 us (1.02 PFLOP/s),
// 128->128 @112x160 110 -> 111 us, 64->128 the same, 64->64 39 -> 44 us: it pays where the chunk loop dominates the workgroup's life,
// so it serves the layers with at least 256 input channels (ISLAM_CONV_R4=0: never, =1: every 3x3 layer with more than 32 outputs).
// ... AND enough 32 x 16 tiles to fill the chip twice: on the flow net's DenseNet layers (up to 565 input channels, but maps of
// 112 x 160 and smaller at B = 8) the larger tile costs parallelism -- flow forward 3.21 -> 3.50 ms with the rule on channels alone.
static bool conv_r4(int Cin, int Cout, int ksize, int B, int H, int W) {
    static const int mode = [] { const char* e = std::getenv("ISLAM_CONV_R4"); return !e ? -1 : (e[0] == '1' ? 1 : 0); }();
    if (ksize != 3 || Cout <= 32 || mode == 0) return false;
    if (mode == 1) return true;
    const long long work = (long long)((W + 31) / 32) * ((H + 15) / 16) * B * ((Cout + 63) / 64);
    static const long long min_work = [] { const char* e = std::getenv("ISLAM_CONV_R4_WORK"); return e ? std::atoll(e) : 1024ll; }();      // (A/B runs)
    return Cin >= 256 && work >= min_work;
}
static int tile_h(int Cout, int Cin = 0, int ksize = 0, int B = 0, int H = 0, int W = 0) {
    return Cout > 32 ? (conv_r4(Cin, Cout, ksize, B, H, W) ? 16 : 8) : 16;
}
static int tiles_of(int B, int H, int W, int th) { return ((W + TW - 1) / TW) * ((H + th - 1) / th) * B; }
// rows of the per-workgroup partial sums a caller must provide room for (the smallest tile any kernel variant uses for this Cout)
int islam_conv_nhwc_stat_blocks(int B, int H, int W, int Cout) { return tiles_of(B, H, W, Cout > 32 ? 8 : 16); }

size_t islam_conv_nhwc_stats_floats(int B, int H, int W, int Cout) {
    return (size_t)islam_conv_nhwc_stat_blocks(B, H, W, Cout) * 2 * Cout + (size_t)RED_BLOCKS * 2 * Cout;
}

int islam_conv_nhwc_bf16(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, const float* bias, const uint16_t* res,
                         uint16_t* y, float* stats, int B, int Cin, int H, int W, int Cout, int ksize, int relu, void* stream) {
    const int in_relu = (relu >> 1) & 1;       // bit 1: ReLU of the INPUT while it is staged (hourglass.py:25-33 `self.relu(x)` before conv1)
    relu &= 1;
    if (B < 1 || H < 1 || W < 1 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7))
        return fail(ISLAM_EARG, "islam_conv_nhwc_bf16: bad shape (Cin=%d, Cout=%d must be multiples of 8)", Cin, Cout);
    if (ksize != 1 && ksize != 3) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16: kernel size %d (1 or 3)", ksize);
    if (stats && (bias || res || relu)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16: statistics go with the raw output (no bias / residual / ReLU)");
    if ((size_t)B * H * W * std::max(Cin, Cout) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = Cout > 32;
    // input channels per staged chunk: 32.  ISLAM_CONV_KC=16 selects 16-channel chunks (44 KB of LDS: three workgroups per CU
    // instead of two, but twice the barriers) -- measured 10-20 % slower on every shape of the stereo net, kept for A/B runs
    static const bool kc16 = [] { const char* e = std::getenv("ISLAM_CONV_KC"); return e && e[0] == '1'; }();
    int rc;
    const bool ws = !bias && !res && !relu && !in_relu && conv_ws_applies(Cin, Cout, ksize, B, H, W);      // weight-stationary persistent kernel (conv_ws.hip)
    const bool ws32 = !bias && !res && !relu && !in_relu && conv_ws32_applies(Cin, Cout, ksize, B, H, W);  // persistent 32 -> 32 kernel (conv_ws32.hip)
#define ISLAM_CONV_LAUNCH(TN_, KS_, ROWS_)                                                                                              \
    (kc16 ? launch<TN_, KS_, ROWS_, 16>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s) \
          : launch<TN_, KS_, ROWS_, 32>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s))
    if (ws) rc = conv_ws_launch(x, wpacked, in_affine, y, stats, B, Cin, H, W, Cout, CoutP, Cin, 0, Cout, 0, s);
    else if (ws32) rc = conv_ws32_launch(x, wpacked, in_affine, y, stats, B, H, W, CoutP, Cin, 0, Cout, 0, s);
    else if (conv_r4(Cin, Cout, ksize, B, H, W))
        rc = launch<64, 3, 4, 16>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s);
    else if (ksize == 3) rc = wide ? ISLAM_CONV_LAUNCH(64, 3, 2) : ISLAM_CONV_LAUNCH(32, 3, 4);
    else rc = wide ? ISLAM_CONV_LAUNCH(64, 1, 2) : ISLAM_CONV_LAUNCH(32, 1, 4);
#undef ISLAM_CONV_LAUNCH
    if (rc != ISLAM_OK) return rc;
    if (stats) {
        const int nblk = ws ? conv_ws_blocks(B, H, W) : ws32 ? conv_ws32_blocks(B, H, W) : tiles_of(B, H, W, tile_h(Cout, Cin, ksize, B, H, W));      // rows of partial sums the launch above wrote
        hipLaunchKernelGGL(partial_fold_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, stats, nblk, 2 * Cout,
                           stats + (size_t)islam_conv_nhwc_stat_blocks(B, H, W, Cout) * 2 * Cout);
        ISLAM_LAUNCH_CHECK();
    }
    return ISLAM_OK;
}


// islam_conv_nhwc_bf16 with the result written into channels [yoff, yoff + Cout) of a (B,H,W,ytot) bf16 tensor -- straight into a
// concatenation under construction (Network/StereoNet7.py:103-105: torch.cat((left features, right features, half-resolution image))
// in front of conv_c0) instead of a dense tensor that torch.cat then copies.  No residual, no statistics.
int islam_conv_nhwc_bf16_into(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, const float* bias, uint16_t* y, int ytot,
                              int yoff, int B, int Cin, int H, int W, int Cout, int ksize, int relu, void* stream) {
    const int in_relu = (relu >> 1) & 1;
    relu &= 1;
    if (B < 1 || H < 1 || W < 1 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7))
        return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_into: bad shape (Cin=%d, Cout=%d must be multiples of 8)", Cin, Cout);
    if (ksize != 1 && ksize != 3) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_into: kernel size %d (1 or 3)", ksize);
    if ((ytot & 7) || (yoff & 7) || yoff < 0 || yoff + Cout > ytot) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_into: output slice %d+%d of %d", yoff, Cout, ytot);
    if ((size_t)B * H * W * std::max(Cin, ytot) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_into: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const Slices sl{Cin, 0, ytot, yoff, nullptr, 0, 0, 0.0f, 1, W, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    const bool wide = Cout > 32;
    if (!bias && !relu && !in_relu && conv_ws_applies(Cin, Cout, ksize, B, H, W))
        return conv_ws_launch(x, wpacked, in_affine, y, nullptr, B, Cin, H, W, Cout, CoutP, Cin, 0, ytot, yoff, s);
    if (!bias && !relu && !in_relu && conv_ws32_applies(Cin, Cout, ksize, B, H, W))
        return conv_ws32_launch(x, wpacked, in_affine, y, nullptr, B, H, W, CoutP, Cin, 0, ytot, yoff, s);
    if (conv_r4(Cin, Cout, ksize, B, H, W)) return launch<64, 3, 4, 16>(x, wpacked, in_affine, bias, nullptr, y, nullptr, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s, sl);
    if (ksize == 3) return wide ? launch<64, 3, 2, 32>(x, wpacked, in_affine, bias, nullptr, y, nullptr, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s, sl)
                                : launch<32, 3, 4, 32>(x, wpacked, in_affine, bias, nullptr, y, nullptr, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s, sl);
    return wide ? launch<64, 1, 2, 32>(x, wpacked, in_affine, bias, nullptr, y, nullptr, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s, sl)
                : launch<32, 1, 4, 32>(x, wpacked, in_affine, bias, nullptr, y, nullptr, B, Cin, CinP, H, W, Cout, CoutP, relu, in_relu, s, sl);
}

// Conv2d(Cin, Cout, k, stride = 2, padding = k / 2) on the same kernel (template parameter S = 2): the feature extractor's layer2 opens
// with a stride-2 3x3 convbn and a stride-2 1x1 downsample convbn (Network/PSM/submodule.py:76-85 `_make_layer(BasicBlock, 64, 16, 2, ...)`,
// :24-26), and the quarter-resolution tail of StereoNet7 is a 2x2 stride-2 convolution (islam_amd/nets.py::_deconv_c11_quarter).
// x (B,Hi,Wi,Cin) -> y (B,Ho,Wo,Cout) with Ho <= (Hi + 2 (k/2) - k) / 2 + 1 (fewer rows / columns may be asked for); in_affine / bias / res
// / stats / relu as in islam_conv_nhwc_bf16.  k = 1, 2, 3.
int islam_conv_nhwc_bf16_s2(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, const float* bias, const uint16_t* res,
                            uint16_t* y, float* stats, int B, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int relu,
                            void* stream) {
    const int in_relu = (relu >> 1) & 1;
    relu &= 1;
    if (B < 1 || Hi < 1 || Wi < 1 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7))
        return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2: bad shape (Cin=%d, Cout=%d must be multiples of 8)", Cin, Cout);
    if (ksize < 1 || ksize > 3) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2: kernel size %d (1, 2 or 3)", ksize);
    const int P = ksize / 2, Hmax = (Hi + 2 * P - ksize) / 2 + 1, Wmax = (Wi + 2 * P - ksize) / 2 + 1;
    if (Ho < 1 || Wo < 1 || Ho > Hmax || Wo > Wmax) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2: output %dx%d of at most %dx%d", Ho, Wo, Hmax, Wmax);
    if (stats && (bias || res || relu)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2: statistics go with the raw output (no bias / residual / ReLU)");
    if ((size_t)B * Hi * Wi * std::max(Cin, Cout) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const Slices sl{Cin, 0, Cout, 0, nullptr, 0, 0, 0.0f, 1, Wo, 0, Hi, Wi};
    hipStream_t s = (hipStream_t)stream;
    const bool wide = Cout > 32;
    int rc, th;
    if (ksize == 3) { th = wide ? 4 : 8;
        rc = wide ? launch<64, 3, 1, 16, false, 2>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu, in_relu, s, sl)
                  : launch<32, 3, 2, 16, false, 2>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu, in_relu, s, sl);
    } else if (ksize == 2) { th = 8;
        rc = wide ? launch<64, 2, 2, 16, false, 2>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu, in_relu, s, sl)
                  : launch<32, 2, 2, 16, false, 2>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu, in_relu, s, sl);
    } else { th = 8;
        rc = wide ? launch<64, 1, 2, 16, false, 2>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu, in_relu, s, sl)
                  : launch<32, 1, 2, 16, false, 2>(x, wpacked, in_affine, bias, res, y, stats, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu, in_relu, s, sl);
    }
    if (rc != ISLAM_OK) return rc;
    if (stats) {
        // (the stride-2 variants use tiles of 4 or 8 rows: islam_conv_nhwc_stats_floats(B, Ho, Wo, Cout) sizes `stats` for 8-row tiles when
        //  Cout > 32 and 16-row tiles otherwise -- the caller sizes it with islam_conv_nhwc_s2_stats_floats)
        const int nblk = tiles_of(B, Ho, Wo, th);
        hipLaunchKernelGGL(partial_fold_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, stats, nblk, 2 * Cout, stats + (size_t)tiles_of(B, Ho, Wo, 4) * 2 * Cout);
        ISLAM_LAUNCH_CHECK();
    }
    return ISLAM_OK;
}

// floats of `stats` for islam_conv_nhwc_bf16_s2 (per-tile partial sums for the smallest tile any of its variants uses + the folded block)
size_t islam_conv_nhwc_s2_stats_floats(int B, int Ho, int Wo, int Cout) {
    return (size_t)tiles_of(B, Ho, Wo, 4) * 2 * Cout + (size_t)RED_BLOCKS * 2 * Cout;
}

// convbn in training mode (Network/PSM/submodule.py:10-13: Conv2d(bias=False) + BatchNorm2d) up to the BatchNorm's [scale | shift]:
// islam_conv_nhwc_bf16 with `stats` followed by islam_bn_finalize, as TWO launches instead of three (fold_finalize_kernel).  Same
// results bit for bit.  counter: one int in device memory, zero before the call and after it (not shared with a call that may run
// concurrently on another stream).  count = B H W; C = Cout <= 256.
int islam_conv_nhwc_bf16_bn(const uint16_t* x, const uint16_t* wpacked, const float* in_affine, uint16_t* y, float* stats, int B, int Cin,
                            int H, int W, int Cout, int ksize, int in_relu, const float* weight, const float* bias, float* running_mean,
                            float* running_var, long long* num_batches_tracked, double momentum, double eps, float* scale_shift,
                            int* counter, void* stream) {
    if (B < 1 || H < 1 || W < 1 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7) || Cout > 256)
        return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_bn: bad shape (Cin=%d, Cout=%d: multiples of 8, Cout <= 256)", Cin, Cout);
    if (ksize != 1 && ksize != 3) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_bn: kernel size %d (1 or 3)", ksize);
    if (!stats || !weight || !bias || !scale_shift || !counter) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_bn: null argument");
    if ((size_t)B * H * W * std::max(Cin, Cout) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_bn: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = Cout > 32;
    const int ir = in_relu ? 1 : 0;
    int rc;
    const bool ws = (!ir || in_affine) && conv_ws_applies(Cin, Cout, ksize, B, H, W);
    const bool ws32 = (!ir || in_affine) && conv_ws32_applies(Cin, Cout, ksize, B, H, W);
    if (ws) rc = conv_ws_launch(x, wpacked, in_affine, y, stats, B, Cin, H, W, Cout, CoutP, Cin, 0, Cout, 0, s);
    else if (ws32) rc = conv_ws32_launch(x, wpacked, in_affine, y, stats, B, H, W, CoutP, Cin, 0, Cout, 0, s);
    else if (conv_r4(Cin, Cout, ksize, B, H, W)) rc = launch<64, 3, 4, 16>(x, wpacked, in_affine, nullptr, nullptr, y, stats, B, Cin, CinP, H, W, Cout, CoutP, 0, ir, s);
    else if (ksize == 3) rc = wide ? launch<64, 3, 2, 32>(x, wpacked, in_affine, nullptr, nullptr, y, stats, B, Cin, CinP, H, W, Cout, CoutP, 0, ir, s)
                                   : launch<32, 3, 4, 32>(x, wpacked, in_affine, nullptr, nullptr, y, stats, B, Cin, CinP, H, W, Cout, CoutP, 0, ir, s);
    else rc = wide ? launch<64, 1, 2, 32>(x, wpacked, in_affine, nullptr, nullptr, y, stats, B, Cin, CinP, H, W, Cout, CoutP, 0, ir, s)
                   : launch<32, 1, 4, 32>(x, wpacked, in_affine, nullptr, nullptr, y, stats, B, Cin, CinP, H, W, Cout, CoutP, 0, ir, s);
    if (rc != ISLAM_OK) return rc;
    const int nblk = ws ? conv_ws_blocks(B, H, W) : ws32 ? conv_ws32_blocks(B, H, W) : tiles_of(B, H, W, tile_h(Cout, Cin, ksize, B, H, W));
    // how the statistics get from the per-workgroup rows to [scale | shift]: 0 = fold, then finalize (two launches); 1 = both in one launch
    // with a ticket (fold_finalize_kernel: measured no faster than two launches, kept for A/B runs); default = the persistent kernels' few
    // rows straight into the finalize (one launch, no fold), everything else as 0
    static const int how = [] { const char* e = std::getenv("ISLAM_BN_FINALIZE"); return e ? std::atoi(e) : 2; }();
    float* folded = stats + (size_t)islam_conv_nhwc_stat_blocks(B, H, W, Cout) * 2 * Cout;
    if (how != 1) {
        const double count = (double)B * H * W;
        if (how == 2 && nblk <= RED_BLOCKS)
            hipLaunchKernelGGL(finalize_rows_kernel<1>, dim3(1), dim3(FF_THREADS), 0, s, stats, nblk, Cout, count, weight, bias, running_mean, running_var,
                               num_batches_tracked, momentum, eps, scale_shift);
        else if (how == 2 && nblk <= 2 * RED_BLOCKS)
            hipLaunchKernelGGL(finalize_rows_kernel<2>, dim3(1), dim3(FF_THREADS), 0, s, stats, nblk, Cout, count, weight, bias, running_mean, running_var,
                               num_batches_tracked, momentum, eps, scale_shift);
        else {
            hipLaunchKernelGGL(partial_fold_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, stats, nblk, 2 * Cout, folded);
            hipLaunchKernelGGL(finalize_rows_kernel<1>, dim3(1), dim3(FF_THREADS), 0, s, folded, RED_BLOCKS, Cout, count, weight, bias, running_mean,
                               running_var, num_batches_tracked, momentum, eps, scale_shift);
        }
        ISLAM_LAUNCH_CHECK();
        return ISLAM_OK;
    }
    hipLaunchKernelGGL(fold_finalize_kernel, dim3(RED_BLOCKS), dim3(FF_THREADS), 0, s, stats, nblk, Cout,
                       folded, (double)B * H * W, weight, bias, running_mean,
                       running_var, num_batches_tracked, momentum, eps, scale_shift, counter);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}


// ConvTranspose2d(Cin, Cout, kernel 4, stride 2, padding 1) + bias (+ ReLU) of the frozen stereo net's decoder
// (Network/StereoNet7.py:78-90 deconv_c7_2 ... deconv_c10, their ReLU and concatenation at :121-136) on the kernel above: four 2x2 convolutions, one per
// output parity class (see Slices).  x: (B,H,W,Cin) bf16; wpacked: [4 classes][4 taps][CoutP][CinP] bf16 (islam_amd/ops.py
// pack_deconv_nhwc_weight); the result goes to channels [yoff, yoff + Cout) of y = (B,2H,2W,ytot) bf16 -- straight into the channel
// slice of the concatenation the decoder builds next, no torch.cat copy of this half.  MIOpen ran these as bf16 implicit-GEMM
// backward-data kernels (0.52 ms per forward at B = 8).
size_t islam_deconv_nhwc_packed_elems(int Cin, int Cout) {
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    return (size_t)16 * CoutP * CinP;
}

int islam_deconv4x4s2_nhwc_bf16(const uint16_t* x, const uint16_t* wpacked, const float* bias, uint16_t* y, int ytot, int yoff, int B, int Cin,
                                int H, int W, int Cout, int relu, void* stream) {
    if (B < 1 || H < 1 || W < 1 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7))
        return fail(ISLAM_EARG, "islam_deconv4x4s2_nhwc_bf16: bad shape (Cin=%d, Cout=%d must be multiples of 8)", Cin, Cout);
    if ((ytot & 7) || (yoff & 7) || yoff < 0 || yoff + Cout > ytot) return fail(ISLAM_EARG, "islam_deconv4x4s2_nhwc_bf16: output slice %d+%d of %d", yoff, Cout, ytot);
    if ((size_t)B * 4 * H * W * std::max(Cin, ytot) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_deconv4x4s2_nhwc_bf16: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const Slices sl{Cin, 0, ytot, yoff, nullptr, 0, 0, 0.0f, 2, W, 1, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    return Cout > 32 ? launch<64, 2, 2, 32>(x, wpacked, nullptr, bias, nullptr, y, nullptr, B * 4, Cin, CinP, H, W, Cout, CoutP, relu & 1, 0, s, sl)
                     : launch<32, 2, 4, 32>(x, wpacked, nullptr, bias, nullptr, y, nullptr, B * 4, Cin, CinP, H, W, Cout, CoutP, relu & 1, 0, s, sl);
}

// The same with the input given as TWO dense tensors x1 (B,H,W,C1), x2 (B,H,W,C2) whose channel concatenation torch.cat((x1, x2), 1) is
// what the layer convolves (Network/StereoNet7.py:121-138: every transposed convolution of the decoder reads a concatenation of the
// previous stage's result and a skip tensor): the concatenation is never written.  C1 a multiple of 32 (whole chunks), C2 of 8; wpacked as
// for Cin = C1 + C2.  Bit-identical to islam_deconv4x4s2_nhwc_bf16 on the concatenated tensor.
int islam_deconv4x4s2_nhwc_bf16_cat(const uint16_t* x1, int C1, const uint16_t* x2, int C2, const uint16_t* wpacked, const float* bias, uint16_t* y,
                                    int ytot, int yoff, int B, int H, int W, int Cout, int relu, void* stream) {
    const int Cin = C1 + C2;
    if (!x1 || !x2 || B < 1 || H < 1 || W < 1 || C1 < 32 || (C1 & 31) || C2 < 8 || (C2 & 7) || Cout < 8 || (Cout & 7))
        return fail(ISLAM_EARG, "islam_deconv4x4s2_nhwc_bf16_cat: bad shape (C1=%d must be a multiple of 32, C2=%d and Cout=%d of 8)", C1, C2, Cout);
    if ((ytot & 7) || (yoff & 7) || yoff < 0 || yoff + Cout > ytot) return fail(ISLAM_EARG, "islam_deconv4x4s2_nhwc_bf16_cat: output slice %d+%d of %d", yoff, Cout, ytot);
    if ((size_t)B * 4 * H * W * std::max(Cin, ytot) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_deconv4x4s2_nhwc_bf16_cat: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const Slices sl{C1, 0, ytot, yoff, nullptr, 0, 0, 0.0f, 2, W, 1, 0, 0};
    const Src2 s2{x2, C2, C1};
    hipStream_t s = (hipStream_t)stream;
    return Cout > 32 ? launch<64, 2, 2, 32, false, 1, true>(x1, wpacked, nullptr, bias, nullptr, y, nullptr, B * 4, Cin, CinP, H, W, Cout, CoutP, relu & 1, 0, s, sl, s2)
                     : launch<32, 2, 4, 32, false, 1, true>(x1, wpacked, nullptr, bias, nullptr, y, nullptr, B * 4, Cin, CinP, H, W, Cout, CoutP, relu & 1, 0, s, sl, s2);
}

// islam_conv_nhwc_bf16_s2 (kernel size 2: the quarter-resolution tail of the stereo net, Network/StereoNet7.py:88-90 through
// islam_amd/nets.py::_deconv_c11_quarter) on the concatenation of two dense tensors, as above.  C1 a multiple of 16.  No statistics.
int islam_conv_nhwc_bf16_s2_cat(const uint16_t* x1, int C1, const uint16_t* x2, int C2, const uint16_t* wpacked, const float* bias, uint16_t* y, int B,
                                int Hi, int Wi, int Cout, int Ho, int Wo, int ksize, int relu, void* stream) {
    const int Cin = C1 + C2;
    if (!x1 || !x2 || B < 1 || Hi < 1 || Wi < 1 || C1 < 16 || (C1 & 15) || C2 < 8 || (C2 & 7) || Cout < 8 || (Cout & 7))
        return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2_cat: bad shape (C1=%d must be a multiple of 16, C2=%d and Cout=%d of 8)", C1, C2, Cout);
    if (ksize != 2) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2_cat: kernel size %d (2)", ksize);
    const int Hmax = Hi / 2 + 1, Wmax = Wi / 2 + 1;                  // padding ksize / 2 = 1: (Hi + 2 - 2) / 2 + 1
    if (Ho < 1 || Wo < 1 || Ho > Hmax || Wo > Wmax) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2_cat: output %dx%d of at most %dx%d", Ho, Wo, Hmax, Wmax);
    if ((size_t)B * Hi * Wi * std::max(Cin, Cout) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_conv_nhwc_bf16_s2_cat: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const Slices sl{C1, 0, Cout, 0, nullptr, 0, 0, 0.0f, 1, Wo, 0, Hi, Wi};
    const Src2 s2{x2, C2, C1};
    hipStream_t s = (hipStream_t)stream;
    return Cout > 32 ? launch<64, 2, 2, 16, false, 2, true>(x1, wpacked, nullptr, bias, nullptr, y, nullptr, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu & 1, (relu >> 1) & 1, s, sl, s2)
                     : launch<32, 2, 2, 16, false, 2, true>(x1, wpacked, nullptr, bias, nullptr, y, nullptr, B, Cin, CinP, Ho, Wo, Cout, CoutP, relu & 1, (relu >> 1) & 1, s, sl, s2);
}

// 3x3 stride-1 convolution of the flow net's DenseNet blocks (Network/PWC/PWCNet.py:16-20 `conv()` = Conv2d + LeakyReLU(0.1),
// :237-292) on the channels-last kernel above.  x: bf16 channels [xoff, xoff + Cin) of a (B,H,W,xtot) mirror of the block's
// concatenation buffer; the output goes as fp32 NCHW into channels [coff, coff + Cout) of y32 (B,ytot,H,W) -- what the
// correlation / warp / transposed-convolution / flow-head consumers read -- and, when ymir is given, as bf16 into channels
// [moff, moff + Cout) of the (B,H,W,mtot) mirror for the next convolution.  Same arithmetic as islam_conv3x3_mfma (bf16
// operands rounded to nearest even, fp32 accumulation, bias, LeakyReLU(slope); slope 1: no activation), at 16-byte loads and
// 32-channel chunks instead of fp32 loads converted on the fly.  Cin, Cout, xtot, xoff, mtot, moff: multiples of 8.  y32 or ymir may
// be NULL (not both).  dilation d > 1: d*d dense convolutions on the sub-grids of the image (H, W multiples of d).
int islam_conv_nhwc_flow(const uint16_t* x, int xtot, int xoff, int Cin, const uint16_t* wpacked, const float* bias, float* y32, int ytot,
                         int coff, uint16_t* ymir, int mtot, int moff, int B, int H, int W, int Cout, int dilation, float slope,
                         void* stream) {
    if (B < 1 || H < 1 || W < 1 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7) || (!y32 && !ymir))
        return fail(ISLAM_EARG, "islam_conv_nhwc_flow: bad shape (Cin=%d, Cout=%d must be multiples of 8; one output at least)", Cin, Cout);
    if (dilation < 1 || H % dilation || W % dilation)
        return fail(ISLAM_EARG, "islam_conv_nhwc_flow: dilation %d must divide the image size %dx%d", dilation, H, W);
    if ((xtot & 7) || (xoff & 7) || xoff < 0 || xoff + Cin > xtot) return fail(ISLAM_EARG, "islam_conv_nhwc_flow: input slice %d+%d of %d", xoff, Cin, xtot);
    if (y32 && (coff < 0 || coff + Cout > ytot)) return fail(ISLAM_EARG, "islam_conv_nhwc_flow: output slice %d+%d of %d", coff, Cout, ytot);
    if (ymir && ((mtot & 7) || (moff & 7) || moff < 0 || moff + Cout > mtot)) return fail(ISLAM_EARG, "islam_conv_nhwc_flow: mirror slice %d+%d of %d", moff, Cout, mtot);
    if ((size_t)B * H * W * std::max(std::max(xtot, mtot), ytot) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_conv_nhwc_flow: tensor too large for 32-bit offsets");
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const int d = dilation;
    const Slices sl{xtot, xoff, ymir ? mtot : Cout, ymir ? moff : 0, y32, ytot, coff, slope, d, W, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    if (conv_r4(Cin, Cout, 3, B * d * d, H / d, W / d))
        return launch<64, 3, 4, 16, true>(x, wpacked, nullptr, bias, nullptr, ymir, nullptr, B * d * d, Cin, CinP, H / d, W / d, Cout, CoutP, 0, 0, s, sl);
    return Cout > 32 ? launch<64, 3, 2, 32, true>(x, wpacked, nullptr, bias, nullptr, ymir, nullptr, B * d * d, Cin, CinP, H / d, W / d, Cout, CoutP, 0, 0, s, sl)
                     : launch<32, 3, 4, 32, true>(x, wpacked, nullptr, bias, nullptr, ymir, nullptr, B * d * d, Cin, CinP, H / d, W / d, Cout, CoutP, 0, 0, s, sl);
}

// fp32 NCHW channels [soff, soff + C) of src (B,stot,H,W)  ->  bf16 channels [doff, doff + C) of dst (B,H,W,dtot), rounded to
// nearest even; channels [doff + C, doff + Cpad) are zeroed (Cpad = C rounded up to 8).  Fills the mirror's slice of what the
// correlation / warp / transposed convolutions produced.  doff, dtot: multiples of 8.
int islam_nchw_f32_to_nhwc_bf16(const float* src, int stot, int soff, uint16_t* dst, int dtot, int doff, int B, int C, int H, int W,
                                void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1 || soff < 0 || soff + C > stot) return fail(ISLAM_EARG, "islam_nchw_f32_to_nhwc_bf16: source slice %d+%d of %d", soff, C, stot);
    const int Cpad = (C + 7) / 8 * 8;
    if ((dtot & 7) || (doff & 7) || doff < 0 || doff + Cpad > dtot) return fail(ISLAM_EARG, "islam_nchw_f32_to_nhwc_bf16: destination slice %d+%d of %d", doff, Cpad, dtot);
    const long long pixels = (long long)H * W;
    dim3 grid((unsigned)((pixels + 63) / 64), (Cpad + 31) / 32, B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, stot, soff, dst, dtot, doff, C, Cpad, pixels);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // extern "C"
