// hourglass.hip -- one launch per Residual module of the frozen stereo net's hourglass stacks, channels-last bf16.
//
// Replaces, for the bf16 execution copy, what the reference runs as three cuDNN convolutions + elementwise ops
// (Network/PSM/hourglass.py:28-52 `Residual.forward`; 35 of them per stereo forward, Network/StereoNet7.py:56-90):
//   y = conv3( relu( conv2( relu( conv1( relu(x) ) ) ) ) ) + res
//   conv1: 1x1 Cin -> h, conv2: 3x3 h -> h (zero padding 1), conv3: 1x1 h -> Cout, each with a bias; h = Cout / 2;
//   res = x (Cin == Cout) or skip_layer(x) (a 1x1 convolution of the RAW x, evaluated by islam_conv_nhwc_bf16 beforehand).
// Launched layer by layer (islam_conv_nhwc_bf16 x 3) a Residual at 1/8 ... 1/32 resolution is three launch-latency-bound
// launches of 11-15 us each, and at 1/2 resolution it moves its h-channel intermediates through HBM four times.  Here a
// workgroup produces an 8 x 16 pixel tile of y from the 10 x 18 patch of x it depends on; the two intermediates stay in LDS:
//   phase 1  t1[10x18 px][h]  = relu(bf16(conv1(relu(x)) + b1)), zero outside the image (conv2's zero padding); the patch of x is
//            streamed through LDS in 32-channel chunks (ReLU applied while it is staged)
//   phase 2  t2[8x16 px][h]   = relu(bf16(conv2(t1) + b2))
//   phase 3  y = bf16(bf16(conv3(t2) + b3) + res), 128 output channels per pass, stored through an LDS transpose (16 bytes per
//            lane along C), the residual read with 16-byte loads
// -- the rounding points of the layer-by-layer path.  GEMM view as in conv_nhwc.hip: M = output channels (A = weights), N = pixels
// (B = activations, [pixel][channel] rows in LDS), v_mfma_f32_32x32x16_bf16; a wave owns one 32-pixel N tile (two tile rows) in
// phases 2 / 3 and one or two of the six N tiles of the 180-pixel patch in phase 1, with all M tiles of the layer.
// WEIGHTS never pass through registers: the host packs every K step's operand image (rows of 32 channels + 16 bytes of
// padding = the conflict-free 80-byte LDS row stride) contiguously, and the waves copy stage s + 1 into the other half of a
// double buffer with LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs, no ds_write) while stage s is
// multiplied; one workgroup barrier per stage.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "../../include/islam_hip.h"
#include "common.h"

namespace {

using namespace islam;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int TH = 8, TW = 16, DW = TW + 2, DH = TH + 2, ND = DH * DW, NT1 = (ND + 31) / 32, NDP = NT1 * 32;   // 180 -> 192 slots
constexpr int NOUT = TH * TW;                       // 128 output pixels = 4 N tiles
constexpr int KC = 32, PS = KC + 8;                 // channels per K stage; elements per staged row (80-byte stride)
constexpr int THREADS = 256;
constexpr int PIECE = 512;                          // elements per LDS-DMA wave instruction (1 KiB)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo16(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi16(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned relu2(unsigned t) {
    if (t & 0x8000u) t &= 0xffff0000u;
    if (t & 0x80000000u) t &= 0x0000ffffu;
    return t;
}

__host__ __device__ constexpr int round_piece(int elems) { return (elems + PIECE - 1) / PIECE * PIECE; }
__host__ __device__ constexpr int taps_per_stage(int MT) { return MT == 1 ? 9 : 3; }

// Packed weights (islam_hg_residual_packed_elems): [phase-1 stages: Cin/32][phase-2 stages: (h/32) * (9/TPS)][phase-3 stages:
// passes * (h/32)], every stage the LDS image [rows][PS] of its A operand, rounded up to whole 1-KiB pieces:
//   phase 1, chunk c:            rows n < h:               W1[n][32c ...]
//   phase 2, chunk c, group g:   rows tl*h + n, tl < TPS:  W2[n][32c ...][tap g*TPS + tl]
//   phase 3, pass p, chunk c:    rows n < PMR:             W3[128p + n][32c ...]   (PMR = min(Cout, 128); zero rows past Cout)
struct Plan {
    int S1, S2, S3, sb1, sb2, sb3, pmr, npass;       // stage counts, stage sizes in elements, rows per phase-3 pass
};
__host__ __device__ inline Plan make_plan(int Cin, int h, int Cout) {
    const int MT = h / 32, tps = taps_per_stage(MT);
    Plan p;
    p.pmr = Cout < 128 ? Cout : 128;
    p.npass = (Cout + 127) / 128;
    p.S1 = Cin / KC; p.S2 = MT * (9 / tps); p.S3 = p.npass * MT;
    p.sb1 = round_piece(h * PS); p.sb2 = round_piece(tps * h * PS); p.sb3 = round_piece(p.pmr * PS);
    return p;
}

template <int MT>
__global__ __launch_bounds__(THREADS) void hg_residual_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                                               unsigned short* __restrict__ y, const unsigned short* __restrict__ wpk,
                                                               const float* __restrict__ bias, int Cin, int Cout, int H, int W,
                                                               int tiles_x, int tiles, int t1_elems, int ra_elems, int wbuf_elems) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    constexpr int HC = 32 * MT, T1S = HC + 8, TPS = taps_per_stage(MT);
    unsigned short* t1 = lds;                                // [NDP][T1S]; phase 3: the output staging tile [NOUT][pmr + 8]
    unsigned short* ra = lds + t1_elems;                     // phase 1: two x-chunk buffers [NDP][PS]; then t2 [NOUT][T1S]
    unsigned short* wb = ra + ra_elems;                      // two weight stage buffers
    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int ho0 = ty * TH, wo0 = tx * TW;
    const Plan pl = make_plan(Cin, HC, Cout);
    const int S = pl.S1 + pl.S2 + pl.S3;
    const unsigned short* xb = x + (size_t)b * H * W * Cin;

    // ---- LDS-DMA of weight stage g into buffer g & 1: wave w copies pieces w, w + 4, ... (lane-linear 1-KiB pieces)
    auto issue = [&](int g) {
        int off, n;
        if (g < pl.S1) { off = g * pl.sb1; n = pl.sb1; }
        else if (g < pl.S1 + pl.S2) { off = pl.S1 * pl.sb1 + (g - pl.S1) * pl.sb2; n = pl.sb2; }
        else { off = pl.S1 * pl.sb1 + pl.S2 * pl.sb2 + (g - pl.S1 - pl.S2) * pl.sb3; n = pl.sb3; }
        const unsigned short* src = wpk + off + lane * 8;
        unsigned short* dst = wb + (g & 1) * wbuf_elems;
        for (int p = wave * PIECE; p < n; p += 4 * PIECE)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + p), (lds_ptr_t)(dst + p), 16, 0, 0);
    };

    // ---- phase 1 staging map: slot q of the 10 x 18 patch <-> (q / 18, q % 18); 192 slots x 4 octets = 768 items, 3 per thread
    constexpr int NXI = NDP * (KC / 8) / THREADS;            // 3
    const int coct = 8 * (tid & 3);
    int goff[NXI], loff[NXI];
    static_for<0, NXI>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int q = (tid + k * THREADS) >> 2;
        const int qy = q / DW, qx = q - qy * DW;
        const int gy = ho0 - 1 + qy, gx = wo0 - 1 + qx;
        loff[k] = q * PS + coct;
        goff[k] = (q < ND && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * Cin + coct : -1;
    });
    u32x4 pre[NXI];
    auto fetch_x = [&](int c0) {
        static_for<0, NXI>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            pre[k] = *reinterpret_cast<const u32x4*>(xb + (goff[k] >= 0 ? (size_t)goff[k] + c0 : (size_t)0));
        });
    };
    auto stage_x = [&](unsigned short* dst) {
        static_for<0, NXI>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            u32x4 v = pre[k];
            v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w);       // conv1 reads relu(x) (hourglass.py:44)
            if (goff[k] < 0) v = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(dst + loff[k]) = v;
        });
    };

    f32x16 acc[2][4];                                        // phase 1: [N tile of the wave][M tile]; phases 2, 3: [0][M tile]
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][a][i] = 0.0f;
    };
    zero_acc();
    issue(0);
    fetch_x(0);

    // ================= phase 1: t1 = relu(bf16(W1 relu(x) + b1)) on the 180-pixel patch =================
    const int nmy1 = wave + 4 < NT1 ? 2 : 1;                 // N tiles wave, wave + 4
    for (int g = 0; g < pl.S1; ++g) {
        unsigned short* xs = ra + (g & 1) * (NDP * PS);
        stage_x(xs);                                         // (its previous readers passed the barrier of stage g - 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of weight stage g have landed
        __syncthreads();
        if (g + 1 < S) issue(g + 1);
        if (g + 1 < pl.S1) fetch_x((g + 1) * KC);
        const unsigned short* wsb = wb + (g & 1) * wbuf_elems;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MT];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const bf16x8*>(wsb + (a * 32 + li) * PS + 16 * ks + 8 * kg);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t < nmy1) {
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(xs + ((wave + 4 * t) * 32 + li) * PS + 16 * ks + 8 * kg);
#pragma unroll
                    for (int a = 0; a < MT; ++a) acc[t][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf, acc[t][a], 0, 0, 0);
                }
            }
        }
    }
    // epilogue 1.  D row (channel) = (reg & 3) + 8 (reg >> 2) + 4 kg, D col (pixel slot) = li
    {
        f32x4 bv[MT * 4];
        static_for<0, MT * 4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            bv[i] = *reinterpret_cast<const f32x4*>(bias + (i / 4) * 32 + 8 * (i % 4) + 4 * kg);
        });
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nmy1) {
                const int q = (wave + 4 * t) * 32 + li;
                const int qy = q / DW, qx = q - qy * DW;
                const int gy = ho0 - 1 + qy, gx = wo0 - 1 + qx;
                const bool in = q < ND && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 bq = bv[a * 4 + gq];
                        unsigned p0 = relu2(pack2(acc[t][a][4 * gq] + bq.x, acc[t][a][4 * gq + 1] + bq.y));
                        unsigned p1 = relu2(pack2(acc[t][a][4 * gq + 2] + bq.z, acc[t][a][4 * gq + 3] + bq.w));
                        if (!in) { p0 = 0; p1 = 0; }         // conv2's zero padding is a zero of ITS input, not relu(b1)
                        *reinterpret_cast<uint2*>(t1 + (size_t)q * T1S + a * 32 + 8 * gq + 4 * kg) = make_uint2(p0, p1);
                    }
            }
        }
    }
    zero_acc();

    // ================= phase 2: t2 = relu(bf16(W2 * t1 + b2)) on the 128 output pixels =================
    // N tile of the wave = tile rows 2 wave, 2 wave + 1.  Second-row lanes take pixel x = (li - 18) mod 16, so that the patch rows a
    // 16-lane group of ds_read_b128 touches stay distinct mod 16 (t1's row pitch is 18 pixels): conflict-free operand reads.
    const int py2 = 2 * wave + (li >> 4), px2 = li < 16 ? li : ((li - 18) & 15);
    const unsigned short* b2base = t1 + (size_t)(py2 * DW + px2) * T1S + 8 * kg;
    for (int g2 = 0; g2 < pl.S2; ++g2) {
        const int g = pl.S1 + g2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // (first pass: also publishes t1)
        if (g + 1 < S) issue(g + 1);
        const unsigned short* wsb = wb + (g & 1) * wbuf_elems;
        const int c = g2 / (9 / TPS), tg = g2 - c * (9 / TPS);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tl = 0; tl < TPS; ++tl) {
                const int tap = tg * TPS + tl, r = tap / 3, s = tap - 3 * r;
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(b2base + (size_t)(r * DW + s) * T1S + c * 32 + 16 * ks);
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(wsb + ((tl * MT + a) * 32 + li) * PS + 16 * ks + 8 * kg);
                    acc[0][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[0][a], 0, 0, 0);
                }
            }
    }
    unsigned short* t2 = ra;                                 // (the x-chunk buffers are dead since the end of phase 1)
    {
        f32x4 bv[MT * 4];
        static_for<0, MT * 4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            bv[i] = *reinterpret_cast<const f32x4*>(bias + HC + (i / 4) * 32 + 8 * (i % 4) + 4 * kg);
        });
        const int p = py2 * TW + px2;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bq = bv[a * 4 + gq];
                const unsigned p0 = relu2(pack2(acc[0][a][4 * gq] + bq.x, acc[0][a][4 * gq + 1] + bq.y));
                const unsigned p1 = relu2(pack2(acc[0][a][4 * gq + 2] + bq.z, acc[0][a][4 * gq + 3] + bq.w));
                *reinterpret_cast<uint2*>(t2 + (size_t)p * T1S + a * 32 + 8 * gq + 4 * kg) = make_uint2(p0, p1);
            }
    }

    // ================= phase 3: y = bf16(bf16(W3 * t2 + b3) + res), pmr channels per pass =================
    const int p3 = wave * 32 + li;                           // natural pixel order: t2's row pitch is 16 pixels
    const unsigned short* b3base = t2 + (size_t)p3 * T1S + 8 * kg;
    const int OS = pl.pmr + 8;                               // staging tile row stride (elements)
    unsigned short* ot = t1;
    const int OCT = pl.pmr / 8;                              // channel octets per pixel and pass
    for (int ps = 0; ps < pl.npass; ++ps) {
        const int n0 = ps * 128;
        const int mt3 = (Cout - n0 < 128 ? Cout - n0 : 128) / 32;     // live M tiles of this pass (uniform)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][a][i] = 0.0f;
        for (int c = 0; c < MT; ++c) {
            const int g = pl.S1 + pl.S2 + ps * MT + c;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                 // (first stage: also publishes t2; later passes: the stores below are done)
            if (g + 1 < S) issue(g + 1);
            const unsigned short* wsb = wb + (g & 1) * wbuf_elems;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(b3base + c * 32 + 16 * ks);
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (a < mt3) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(wsb + (a * 32 + li) * PS + 16 * ks + 8 * kg);
                        acc[0][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[0][a], 0, 0, 0);
                    }
            }
        }
        // epilogue 3: through LDS (t1 is dead) so that the stores are 16 bytes per lane along C
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < mt3)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + 2 * HC + n0 + a * 32 + 8 * gq + 4 * kg);
                    *reinterpret_cast<uint2*>(ot + (size_t)p3 * OS + a * 32 + 8 * gq + 4 * kg) =
                        make_uint2(pack2(acc[0][a][4 * gq] + bq.x, acc[0][a][4 * gq + 1] + bq.y),
                                   pack2(acc[0][a][4 * gq + 2] + bq.z, acc[0][a][4 * gq + 3] + bq.w));
                }
        __syncthreads();
        const int nitem = NOUT * OCT;
        for (int it = tid; it < nitem; it += THREADS) {
            const int px = it / OCT, oc = it - px * OCT;
            const int ho = ho0 + (px >> 4), wo = wo0 + (px & 15), n = n0 + 8 * oc;
            if (ho >= H || wo >= W || n >= Cout) continue;
            const size_t o = (((size_t)b * H + ho) * W + wo) * Cout + n;
            const u32x4 r = *reinterpret_cast<const u32x4*>(res + o);
            u32x4 v = *reinterpret_cast<const u32x4*>(ot + (size_t)px * OS + 8 * oc);
            v.x = pack2(lo16(v.x) + lo16(r.x), hi16(v.x) + hi16(r.x));
            v.y = pack2(lo16(v.y) + lo16(r.y), hi16(v.y) + hi16(r.y));
            v.z = pack2(lo16(v.z) + lo16(r.z), hi16(v.z) + hi16(r.z));
            v.w = pack2(lo16(v.w) + lo16(r.w), hi16(v.w) + hi16(r.w));
            *reinterpret_cast<u32x4*>(y + o) = v;
        }
        // (the next pass's first barrier orders these reads of the staging tile before its epilogue writes)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (h = 32; conv_c1, conv_c2, conv_c10 of StereoNet7: 15 of the 35 modules, at 1/2 ... 1/8 resolution, where the
// module is bound by HBM traffic, not by weights): ALL weights of the module live in registers as MFMA A fragments (26 fragments =
// 104 VGPRs per lane, loaded once), the workgroups are persistent and walk the tiles, and the next tile's patch is requested
// while this tile is multiplied.  LDS holds activations only (56 KB: two workgroups per CU cover each other's barriers).
// Packed weights: [26 fragments][64 lanes][8] bf16, fragment f of lane (li, kg):
//   f = ks          (ks < 4):            W1[li][16 ks + 8 kg ...]
//   f = 4 + 2 tap + ks (tap < 9, ks < 2): W2[li][16 ks + 8 kg ...][tap]
//   f = 22 + 2 a + ks  (a < 2, ks < 2):   W3[32 a + li][16 ks + 8 kg ...]
constexpr int L_XS = 64 + 8, L_TS = 32 + 8, L_OS = 64 + 8;  // row strides (elements) of the patch, the intermediates, the output staging tile
constexpr int L_NFRAG = 26;
constexpr int L_LDS_ELEMS = NDP * L_XS + NOUT * L_OS + NOUT * L_TS + 256;  // xs | t1 (also the output staging tile) | t2 | biases (128 floats)

__global__ __launch_bounds__(THREADS, 2) void hg_residual64_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                                    const unsigned short* __restrict__ wpk, const float* __restrict__ bias,
                                                                    int H, int W, int tiles_x, int tiles, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    constexpr int C = 64;
    unsigned short* xs = lds;                                // [NDP][L_XS]  relu(x) of the 10 x 18 patch
    unsigned short* t1 = xs + NDP * L_XS;                    // [NDP][L_TS]; later the output staging tile [NOUT][L_OS]
    unsigned short* t2 = t1 + NOUT * L_OS;                   // [NOUT][L_TS]
    static_assert(NOUT * L_OS >= NDP * L_TS, "staging tile covers t1");
    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    bf16x8 wf[L_NFRAG];
    static_for<0, L_NFRAG>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        wf[f] = *reinterpret_cast<const bf16x8*>(wpk + f * 512 + lane * 8);
    });
    float* bl = reinterpret_cast<float*>(t2 + NOUT * L_TS);  // [b1 (32) | b2 (32) | b3 (64)] behind t2: read at the epilogues (64 VGPRs if held)
    if (tid < 128) bl[tid] = bias[tid];
    // (visible to every wave after barrier 1 of the first tile)

    constexpr int NXI = NDP * (C / 8) / THREADS;             // 6 (slot, octet) items of the patch per thread
    const int coct = 8 * (tid & 7);
    u32x4 pre[NXI];
    unsigned inmask = 0;                                     // bit k: item k of the FETCHED tile lies inside the image
    auto fetch = [&](int t) {
        const int b = t / tiles, tile = t - b * tiles;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const unsigned short* xb = x + (size_t)b * H * W * C;
        inmask = 0;
        static_for<0, NXI>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int q = (tid + k * THREADS) >> 3;
            const int qy = q / DW, qx = q - qy * DW;
            const int gy = ty * TH - 1 + qy, gx = tx * TW - 1 + qx;
            const bool in = q < ND && gy >= 0 && gy < H && gx >= 0 && gx < W;
            inmask |= (in ? 1u : 0u) << k;
            pre[k] = *reinterpret_cast<const u32x4*>(xb + (in ? (size_t)(gy * W + gx) * C + coct : (size_t)0));
        });
    };
    auto stage = [&]() {
        static_for<0, NXI>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int q = (tid + k * THREADS) >> 3;
            u32x4 v = pre[k];
            v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w);
            if (!((inmask >> k) & 1u)) v = u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(xs + q * L_XS + coct) = v;
        });
    };

    const int py2 = 2 * wave + (li >> 4), px2 = li < 16 ? li : ((li - 18) & 15);      // phase-2 pixel of the lane (see hg_residual_kernel)
    const unsigned short* b2base = t1 + (py2 * DW + px2) * L_TS + 8 * kg;
    const int p2 = py2 * TW + px2, p3 = wave * 32 + li;
    const int nmy1 = wave + 4 < NT1 ? 2 : 1;

    int t = blockIdx.x;
    if (t < ntiles) fetch(t);
    for (; t < ntiles; t += gridDim.x) {
        const int b = t / tiles, tile = t - b * tiles;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int ho0 = ty * TH, wo0 = tx * TW;
        stage();                                             // (every wave has left phase 1 of the previous tile: barriers 2-4)
        __syncthreads();                                     // barrier 1: patch staged; the previous tile's store loop is done
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);      // in flight while this tile is multiplied

        // ---- phase 1
        f32x16 a1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) a1[u][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u < nmy1) {
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(xs + ((wave + 4 * u) * 32 + li) * L_XS + 16 * ks + 8 * kg);
                    a1[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], bf, a1[u], 0, 0, 0);
                }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (u < nmy1) {
                const int q = (wave + 4 * u) * 32 + li;
                const int qy = q / DW, qx = q - qy * DW;
                const int gy = ho0 - 1 + qy, gx = wo0 - 1 + qx;
                const bool in = q < ND && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + 8 * gq + 4 * kg);
                    unsigned p0 = relu2(pack2(a1[u][4 * gq] + bq.x, a1[u][4 * gq + 1] + bq.y));
                    unsigned p1 = relu2(pack2(a1[u][4 * gq + 2] + bq.z, a1[u][4 * gq + 3] + bq.w));
                    if (!in) { p0 = 0; p1 = 0; }
                    *reinterpret_cast<uint2*>(t1 + q * L_TS + 8 * gq + 4 * kg) = make_uint2(p0, p1);
                }
            }
        __syncthreads();                                     // barrier 2: t1 complete

        // ---- phase 2
        f32x16 a2;
#pragma unroll
        for (int i = 0; i < 16; ++i) a2[i] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(b2base + ((tap / 3) * DW + tap % 3) * L_TS + 16 * ks);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[4 + 2 * tap + ks], bf, a2, 0, 0, 0);
            }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * gq + 4 * kg);
            const unsigned p0 = relu2(pack2(a2[4 * gq] + bq.x, a2[4 * gq + 1] + bq.y));
            const unsigned p1 = relu2(pack2(a2[4 * gq + 2] + bq.z, a2[4 * gq + 3] + bq.w));
            *reinterpret_cast<uint2*>(t2 + p2 * L_TS + 8 * gq + 4 * kg) = make_uint2(p0, p1);
        }
        __syncthreads();                                     // barrier 3: t2 complete, t1 dead

        // ---- phase 3
        f32x16 a3[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) a3[a][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(t2 + p3 * L_TS + 16 * ks + 8 * kg);
#pragma unroll
            for (int a = 0; a < 2; ++a) a3[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[22 + 2 * a + ks], bf, a3[a], 0, 0, 0);
        }
        unsigned short* ot = t1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + 64 + 32 * a + 8 * gq + 4 * kg);
                *reinterpret_cast<uint2*>(ot + p3 * L_OS + a * 32 + 8 * gq + 4 * kg) =
                    make_uint2(pack2(a3[a][4 * gq] + bq.x, a3[a][4 * gq + 1] + bq.y), pack2(a3[a][4 * gq + 2] + bq.z, a3[a][4 * gq + 3] + bq.w));
            }
        __syncthreads();                                     // barrier 4: output tile staged
        // y = bf16(staged + x): 128 pixels x 8 octets = 1024 items, 4 per thread, the residuals requested at once
        u32x4 rv[4];
        const size_t img = (size_t)b * H * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int it = tid + k * THREADS, px = it >> 3;
            const int ho = ho0 + (px >> 4), wo = wo0 + (px & 15);
            const bool ok = ho < H && wo < W;
            rv[k] = *reinterpret_cast<const u32x4*>(x + (ok ? (img + (size_t)ho * W + wo) * C + coct : (size_t)0));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int it = tid + k * THREADS, px = it >> 3;
            const int ho = ho0 + (px >> 4), wo = wo0 + (px & 15);
            if (ho >= H || wo >= W) continue;
            u32x4 v = *reinterpret_cast<const u32x4*>(ot + px * L_OS + coct);
            const u32x4 r = rv[k];
            v.x = pack2(lo16(v.x) + lo16(r.x), hi16(v.x) + hi16(r.x));
            v.y = pack2(lo16(v.y) + lo16(r.y), hi16(v.y) + hi16(r.y));
            v.z = pack2(lo16(v.z) + lo16(r.z), hi16(v.z) + hi16(r.z));
            v.w = pack2(lo16(v.w) + lo16(r.w), hi16(v.w) + hi16(r.w));
            *reinterpret_cast<u32x4*>(y + (img + (size_t)ho * W + wo) * C + coct) = v;
        }
    }
}

int launch64(const unsigned short* x, unsigned short* y, const unsigned short* wpk, const float* bias, int B, int H, int W, hipStream_t s) {
    const size_t lds = (size_t)L_LDS_ELEMS * sizeof(unsigned short);
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    static int ncu[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)hg_residual64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISLAM_HIP_CHECK(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_set[dev] = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = tiles_x * tiles_y * B;
    const int cus = (dev >= 0 && dev < 64 && ncu[dev] > 0) ? ncu[dev] : 256;
    const int grid = std::min(ntiles, 2 * cus);              // persistent: two workgroups per CU walk the tiles
    hipLaunchKernelGGL(hg_residual64_kernel, dim3(grid), dim3(THREADS), lds, s, x, y, wpk, bias, H, W, tiles_x, tiles_x * tiles_y, ntiles);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

struct LdsPlan { int t1, ra, wbuf; size_t bytes; };
LdsPlan lds_plan(int Cin, int h, int Cout) {
    const Plan p = make_plan(Cin, h, Cout);
    LdsPlan l;
    l.t1 = std::max(NDP * (h + 8), NOUT * (p.pmr + 8));
    l.ra = std::max(2 * NDP * PS, NOUT * (h + 8));
    l.wbuf = std::max(p.sb1, std::max(p.sb2, p.sb3));
    l.t1 = (l.t1 + 7) / 8 * 8; l.ra = (l.ra + 7) / 8 * 8;
    l.bytes = ((size_t)l.t1 + l.ra + 2 * (size_t)l.wbuf) * sizeof(unsigned short);
    return l;
}

template <int MT>
int launch(const unsigned short* x, const unsigned short* res, unsigned short* y, const unsigned short* wpk, const float* bias, int B, int Cin,
           int Cout, int H, int W, hipStream_t s) {
    const LdsPlan l = lds_plan(Cin, 32 * MT, Cout);
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)hg_residual_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL((hg_residual_kernel<MT>), dim3(tiles_x * tiles_y * B), dim3(THREADS), l.bytes, s, x, res, y, wpk, bias, Cin, Cout, H, W,
                       tiles_x, tiles_x * tiles_y, l.t1, l.ra, l.wbuf);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // namespace

extern "C" {

size_t islam_hg_residual_packed_elems(int Cin, int Cout) {
    if (Cin < 32 || (Cin & 31) || Cout < 64 || (Cout & 63) || Cout > 256) return 0;
    if (Cin == 64 && Cout == 64) return (size_t)L_NFRAG * 512;       // register-resident fragments (hg_residual64_kernel)
    const Plan p = make_plan(Cin, Cout / 2, Cout);
    return (size_t)p.S1 * p.sb1 + (size_t)p.S2 * p.sb2 + (size_t)p.S3 * p.sb3;
}

int islam_hg_residual_nhwc_bf16(const uint16_t* x, const uint16_t* res, uint16_t* y, const uint16_t* wpacked, const float* bias, int B, int Cin,
                                int H, int W, int Cout, void* stream) {
    if (B < 1 || H < 1 || W < 1 || Cin < 32 || (Cin & 31) || Cout < 64 || (Cout & 63) || Cout > 256)
        return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: bad shape (Cin=%d a multiple of 32, Cout=%d a multiple of 64 up to 256)", Cin, Cout);
    if (!x || !res || !y || !wpacked || !bias) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: null argument");
    if ((size_t)B * H * W * std::max(Cin, Cout) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: tensor too large for 32-bit offsets");
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64 && Cout == 64) {
        if (res != x) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: the 64 -> 64 module has no skip convolution (res must be x)");
        return launch64(x, y, wpacked, bias, B, H, W, s);
    }
    switch (Cout / 64) {
        case 1: return launch<1>(x, res, y, wpacked, bias, B, Cin, Cout, H, W, s);
        case 2: return launch<2>(x, res, y, wpacked, bias, B, Cin, Cout, H, W, s);
        case 3: return launch<3>(x, res, y, wpacked, bias, B, Cin, Cout, H, W, s);
        default: return launch<4>(x, res, y, wpacked, bias, B, Cin, Cout, H, W, s);
    }
}

}  // extern "C"
