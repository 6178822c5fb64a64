// conv_ws32.hip -- persistent 3x3 stride-1 convolution, 32 -> 32 channels, bf16 channels-last, for the half-resolution head of the frozen
// stereo network's feature extractor (Network/PSM/submodule.py:10-13 `convbn`, :66-155 feature_extraction: firstconv's second and third
// convolution and the six convolutions of layer1 -- eight launches per forward at 224 x 320 x 16 images).  Same arithmetic contract as
// conv_nhwc.hip's conv_nhwc_kernel<32, 3, 4, 32> (bf16 operands, fp32 accumulation in the same order, output rounded to nearest even, the
// PREVIOUS BatchNorm + ReLU applied while the input is staged, THIS layer's BatchNorm partial sums from the epilogue): bit-identical outputs.
//
// At 32 channels the layer is memory-bound (147 MB in + out per launch = 24 us at 6 TB/s; 288 MFMA-clocks per 32 output pixels), and the
// tile kernel spends a workgroup's whole life outside any steady state: one 32-channel chunk = set-up (index arithmetic, the first loads'
// full latency), one staging pass that also re-stages the 18 KB of weights, 72 MFMAs per wave, epilogue -- 56 us per launch at 3 TB/s.
// Here two workgroups per CU are persistent: the wave's 18 weight operands live in 72 registers for the whole launch, every workgroup walks
// a contiguous range of 32 x 16-pixel tiles, and the NEXT tile's halo (10 x 16 bytes per thread) is requested into registers before the
// current tile is multiplied and stored, so no load latency is exposed after the first tile; the per-item index arithmetic is done once
// per launch (byte offsets from the tile's first halo pixel, edge masks), not once per tile.  The two workgroups of a CU overlap each
// other's phases as the tile kernel's did.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/islam_hip.h"
#include "common.h"
#include "conv_ws.h"

namespace islam {
namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ unsigned relu2(unsigned t) {            // ReLU of packed bf16 = max on the int16 lanes (sign-magnitude)
    const s16x2_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, t), z));
}

constexpr int C = 32, ROWS = 4, TW = 32, TH = 4 * ROWS, THREADS = 256;     // tile 32 x 16 pixels, wave w owns rows 4 w .. 4 w + 3
constexpr int IH = TH + 2, IW = TW + 2, NPIX = IH * IW;            // 18 x 34 halo pixels
constexpr int PS = C + 8;                                          // LDS pixel stride (elements): 80 bytes, conflict-free 16-byte reads
constexpr int OPP = C / 8;                                         // 16-byte octets per pixel
constexpr int NIN = (NPIX * OPP + THREADS - 1) / THREADS;          // 10 staging items per thread
constexpr int NK = C / 16, NR = ROWS + 2;
constexpr int PPT = TW * TH * OPP / THREADS;                       // 8 store items per thread
constexpr int L_DUMMY = NPIX * PS;                                 // halo tile (the output tile and the final reduction reuse it) | 16 bytes
constexpr size_t LDS_BYTES = (size_t)(L_DUMMY + 8) * sizeof(unsigned short);
static_assert(TW * TH * PS <= NPIX * PS && (size_t)THREADS * 17 * sizeof(float) <= (size_t)NPIX * PS * sizeof(unsigned short), "aliases of the halo tile");

template <bool AFFINE>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_ws32_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wp,
                                                                  const float* __restrict__ in_affine, unsigned short* __restrict__ y,
                                                                  float* __restrict__ partial, int H, int W, int tiles_x, int tiles_y, int ntiles,
                                                                  int xs, int xoff, int ys, int yoff, int CoutP) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* lin = lds;                                     // [IH][IW][PS]; after the multiply phase [TW*TH][PS] output pixels
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kg = lane >> 5, li = lane & 31;
    const int G = gridDim.x;
    const int wgl = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;      // XCD x: a contiguous range of workgroups
    const int t0 = (int)((long long)wgl * ntiles / G), t1 = (int)((long long)(wgl + 1) * ntiles / G);

    // the wave's weights (all four waves hold the same 18 operands: output channel li, taps x k-slices)
    bf16x8 wr[9 * NK];
    static_for<0, 9 * NK>([&](auto ii) {
        constexpr int i = decltype(ii)::value, tap = i / NK, ks = i % NK;
        wr[i] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)tap * CoutP + li) * C + 16 * ks + 8 * kg);
    });
    const int oct = tid & (OPP - 1), prow = tid >> 2;             // the thread's channel octet (the same for all its items), first halo pixel
    f32x4 s0 = {1, 1, 1, 1}, s1 = s0, h0 = {0, 0, 0, 0}, h1 = h0;
    if constexpr (AFFINE) {
        const float* sc = in_affine + 8 * oct;                     // [scale(32) | shift(32)]
        s0 = *reinterpret_cast<const f32x4*>(sc); s1 = *reinterpret_cast<const f32x4*>(sc + 4);
        h0 = *reinterpret_cast<const f32x4*>(sc + C); h1 = *reinterpret_cast<const f32x4*>(sc + C + 4);
    }
    // item k of a thread = halo pixel prow + 64 k, channel octet oct: byte offset from the tile's first halo pixel and its bit in the masks of
    // the halo's top / bottom row and left / right column (whole tiles only: these are the only pixels that can lie outside the image)
    unsigned relb[NIN], m_pix = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
    static_for<0, NIN>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int pix = prow + 64 * k, yy = pix / IW, xx = pix - yy * IW;
        relb[k] = (unsigned)(((yy * W + xx) * xs + 8 * oct) * 2);
        m_pix |= (unsigned)(pix < NPIX) << k;
        m_top |= (unsigned)(yy == 0) << k; m_bot |= (unsigned)(yy == IH - 1) << k;
        m_left |= (unsigned)(xx == 0) << k; m_right |= (unsigned)(xx == IW - 1) << k;
    });
    const unsigned safeb = (unsigned)(((W + 1) * xs + 8 * oct) * 2);      // the tile's first interior pixel: what masked items read
    const unsigned lds_st = (unsigned)((prow * PS + 8 * oct) * 2);
    const unsigned g_st = (unsigned)((((tid >> 7) * W + ((tid >> 2) & 31)) * ys + 8 * oct) * 2);      // store item j: row 2 j + (tid >> 7), column (tid >> 2) & 31

    u32x4 pre[NIN];
    unsigned okmask = 0, oknext = 0;
    auto fetch_tile = [&](int t) {                                 // requests the halo of tile t (past the range: a valid address, everything masked)
        const int tc = t < ntiles ? t : ntiles - 1;
        const int ty = tc % tiles_y, q = tc / tiles_y, tx = q % tiles_x, b = q / tiles_x;
        const char* fbase = reinterpret_cast<const char*>(x + xoff) + ((long long)((b * H + ty * TH - 1) * W + tx * TW - 1) * xs) * 2;
        const unsigned out = (ty == 0 ? m_top : 0u) | (ty == tiles_y - 1 ? m_bot : 0u) | (tx == 0 ? m_left : 0u) | (tx == tiles_x - 1 ? m_right : 0u);
        oknext = t < t1 ? (m_pix & ~out) : 0u;
        static_for<0, NIN>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            pre[k] = *reinterpret_cast<const u32x4*>(fbase + (((oknext >> k) & 1u) ? relb[k] : safeb));
        });
    };
    float sm[8], sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[i] = 0.0f; sq[i] = 0.0f; }

    const unsigned short* bbase = lin + ((size_t)(ROWS * wave) * IW + li) * PS + 8 * kg;
    fetch_tile(t0);
    for (int t = t0; t < t1; ++t) {
        // ---- stage: normalise, zero-pad, registers -> LDS
        okmask = oknext;
        static_for<0, NIN>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            u32x4 v = pre[k];
            if constexpr (AFFINE) {
                v.x = relu2(pack2(fmaf(lo16(v.x), s0.x, h0.x), fmaf(hi16(v.x), s0.y, h0.y)));
                v.y = relu2(pack2(fmaf(lo16(v.y), s0.z, h0.z), fmaf(hi16(v.y), s0.w, h0.w)));
                v.z = relu2(pack2(fmaf(lo16(v.z), s1.x, h1.x), fmaf(hi16(v.z), s1.y, h1.y)));
                v.w = relu2(pack2(fmaf(lo16(v.w), s1.z, h1.z), fmaf(hi16(v.w), s1.w, h1.w)));
            }
            if (!((okmask >> k) & 1u)) v = u32x4{0, 0, 0, 0};     // zero padding of the NORMALISED activation
            char* d = reinterpret_cast<char*>(lin) + lds_st + k * 64 * PS * 2;
            if constexpr (k == NIN - 1) d = ((m_pix >> k) & 1u) ? d : reinterpret_cast<char*>(lds + L_DUMMY);
            *reinterpret_cast<u32x4*>(d) = v;
        });
        __syncthreads();
        fetch_tile(t + 1);                                         // in flight while this tile is multiplied and stored
        // ---- multiply: 72 MFMAs per wave (its four pixel rows x 9 taps x 2 k-slices), operand rows reused across the vertical taps
        f32x16 acc[ROWS];
#pragma unroll
        for (int p = 0; p < ROWS; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[p][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                bf16x8 bf[NR];
#pragma unroll
                for (int q = 0; q < NR; ++q) bf[q] = *reinterpret_cast<const bf16x8*>(bbase + ((size_t)q * IW + s) * PS + 16 * ks);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int p = 0; p < ROWS; ++p)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[(r * 3 + s) * NK + ks], bf[p + r], acc[p], 0, 0, 0);
            }
        __syncthreads();                                           // every wave is done with the halo tile
        // ---- accumulators -> LDS.  D row (channel) = (reg&3) + 8*(reg>>2) + 4*kg, D col (pixel x) = li: [pixel][32 channels] bf16, rounded once
#pragma unroll
        for (int p = 0; p < ROWS; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(lin + (size_t)((ROWS * wave + p) * TW + li) * PS + 8 * g + 4 * kg) =
                    make_uint2(pack2(acc[p][4 * g], acc[p][4 * g + 1]), pack2(acc[p][4 * g + 2], acc[p][4 * g + 3]));
        __syncthreads();
        // ---- store phase: 16 bytes per lane along C (4 lanes = one pixel's 64 bytes), BatchNorm sums of the stored (rounded) values
        {
            const int ty = t % tiles_y, q = t / tiles_y, tx = q % tiles_x, b = q / tiles_x;
            char* ebase = reinterpret_cast<char*>(y + yoff) + ((long long)((b * H + ty * TH) * W + tx * TW) * ys) * 2;
            static_for<0, PPT>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lin) + lds_st + j * 64 * PS * 2);
                if (partial) {
                    const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float lo = lo16(wv[i]), hi = hi16(wv[i]);
                        sm[2 * i] += lo; sq[2 * i] = fmaf(lo, lo, sq[2 * i]);
                        sm[2 * i + 1] += hi; sq[2 * i + 1] = fmaf(hi, hi, sq[2 * i + 1]);
                    }
                }
                *reinterpret_cast<uint4*>(ebase + ((long long)(2 * j) * W * ys) * 2 + g_st) = v;
            });
        }
        __syncthreads();                                           // the next tile's staging overwrites the output tile
    }
    if (partial) {
        // per-workgroup sums over all its tiles, per channel, in a fixed order: thread tid holds the sums of channel octet `oct` over its
        // pixels; one lane per (channel, moment) adds the THREADS / OPP threads of that octet in thread order
        float* red = reinterpret_cast<float*>(lds);                // [THREADS][17]
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 17 + i] = sm[i]; red[tid * 17 + 8 + i] = sq[i]; }
        __syncthreads();
        if (tid < 2 * C) {
            const int which = tid / C, c = tid - which * C, o2 = c >> 3, i = c & 7;
            float tsum = 0.0f;
            for (int m = 0; m < THREADS / OPP; ++m) tsum += red[(o2 + OPP * m) * 17 + 8 * which + i];
            partial[((size_t)wgl * 2 + which) * C + c] = tsum;
        }
    }
}

}  // namespace

// whole tiles only: the image is a multiple of 32 x 16 pixels; mode 1: at least 1024 tiles (two per workgroup)
bool conv_ws32_applies(int Cin, int Cout, int ksize, int B, int H, int W) {
    const int mode = conv_ws_set_mode(-1);
    if (!mode || ksize != 3 || Cin != C || Cout != C || (H % TH) || (W % TW)) return false;
    return mode == 2 || (long long)B * (H / TH) * (W / TW) >= 1024;
}

// two workgroups per CU, and never more than the callers' room for per-workgroup partial sums (one row per 32 x 16-pixel tile)
int conv_ws32_blocks(int B, int H, int W) {
    const long long ntiles = (long long)B * (H / TH) * (W / TW);
    const int slots = 2 * (256 - conv_ws_spare_cus());
    return conv_ws_balanced(ntiles, slots);                        // (2240 tiles: 448 workgroups of 5, a quarter of the slots stay free)
}

int conv_ws32_launch(const unsigned short* x, const unsigned short* wp, const float* in_affine, unsigned short* y, float* partial, int B, int H, int W,
                     int CoutP, int xs, int xoff, int ys, int yoff, hipStream_t s) {
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_ws32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_ws32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set[dev] = true;
    }
    const int tiles_x = W / TW, tiles_y = H / TH, ntiles = tiles_x * tiles_y * B;
    const int G = conv_ws32_blocks(B, H, W);
    if (in_affine)
        hipLaunchKernelGGL(conv3x3_ws32_kernel<true>, dim3(G), dim3(THREADS), LDS_BYTES, s, x, wp, in_affine, y, partial, H, W, tiles_x, tiles_y, ntiles,
                           xs, xoff, ys, yoff, CoutP);
    else
        hipLaunchKernelGGL(conv3x3_ws32_kernel<false>, dim3(G), dim3(THREADS), LDS_BYTES, s, x, wp, in_affine, y, partial, H, W, tiles_x, tiles_y, ntiles,
                           xs, xoff, ys, yoff, CoutP);
    ISLAM_LAUNCH_CHECK();
    conv_ws_count_launch(1);
    return ISLAM_OK;
}

}  // namespace islam
