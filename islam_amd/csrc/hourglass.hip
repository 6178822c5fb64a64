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
// WEIGHTS never pass through LDS: the host packs them as MFMA A fragments in consumption order (1 KiB = one fragment of all 64 lanes)
// and every wave loads the fragments of the output-channel tiles it owns straight from L2 through a small register ring (details in
// front of the kernels below; a first version that streamed weight stages through LDS by LDS-DMA was bound by the DMA round trip per
// stage and is gone).  Only the raw input patch arrives by LDS-DMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "../../include/islam_hip.h"
#include "common.h"

// scripts/hg_probe.sh builds an experiment variant (never the product library) that timestamps the phases of one workgroup
#ifndef ISLAM_HG_PROBE
#define ISLAM_HG_PROBE 0
#endif
#if ISLAM_HG_PROBE
__device__ long long islam_hg_probe_buf[64];
#define HPROBE(slot) do { __builtin_amdgcn_sched_barrier(0); if (hprb) islam_hg_probe_buf[(slot) + 16 * wave] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HPROBE(slot) do { } while (0)
#endif

// scheduling fence for vector-memory and matrix instructions (LLVM sched_barrier mask: VALU, SALU and DS may cross): keeps a prefetch
// where the source issues it -- the machine scheduler otherwise sinks global loads next to their first use to save registers
#define VMEM_PIN() __builtin_amdgcn_sched_barrier(0x386)

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

constexpr int THREADS = 256;


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
// ReLU of packed bf16: a negative bf16 is a negative int16 (sign-magnitude), so max(., 0) on the int16 lanes is the ReLU: ONE v_pk_max_i16 per pair
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu2(unsigned t) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, t), z));
}
__device__ __forceinline__ bf16x8 relu8(bf16x8 v) {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_elementwise_max(v, z);
}
__device__ __attribute__((aligned(16))) unsigned islam_hg_zero[4];      // 16 bytes of zeros: what LDS-DMA lanes of out-of-image pixels copy

// Wider modules (Cout = 128 / 192 / 256, h = Cout / 2 = 32 MT): a wave OWNS output-channel tiles.  Wave (mg, ng) computes M tile
// mg of phases 1 / 2 (and M tiles mg, mg + MT of phase 3) for the N tiles of its pixel group ng, so the weight rows it multiplies
// with are its own: it loads them straight from global memory (L2) as MFMA A fragments -- the host packs them in consumption
// order, 1 KiB = one fragment of all 64 lanes per load -- through a small register ring, several fragments ahead.  No weight ever
// touches LDS and no barrier separates K steps: LDS holds activations only (the whole relu(x) patch, t1, t2, the output staging
// tile), and the workgroup meets at four barriers (patch staged, t1, t2, output tile).  [A first version streamed the weights
// through LDS by LDS-DMA, double-buffered, one barrier per K stage: bound by the DMA round trip per stage, 37 us for a 256-channel
// module on a 14 x 20 map.]  MT = 4: four waves, one M tile each, all N tiles; MT = 3: three waves; MT = 2: four waves = 2 M tiles x 2
// pixel groups.
// Packed weights, fragments of 512 elements ([64 lanes][8], lane = (li = lane & 31, kg = lane >> 5)):
//   phase 1: [mg < MT][ks < Cin/16]                 W1[32 mg + li][16 ks + 8 kg ...]
//   phase 2: [mg < MT][c < MT][tap < 9][ks2 < 2]    W2[32 mg + li][32 c + 16 ks2 + 8 kg ...][tap]
//   phase 3: [m3 < 2 MT][ks < h/16]                 W3[32 m3 + li][16 ks + 8 kg ...]
template <int MT> struct Geo {
    static constexpr int WAVES = MT == 3 ? 3 : 4, NG = MT == 2 ? 2 : 1, THREADS_ = 64 * WAVES;
    static constexpr int HC = 32 * MT, T1S = HC + 8, COUT = 2 * HC, OS = COUT + 8;
    static constexpr int N1 = NT1 / NG, N2 = 4 / NG;        // N tiles per wave in phase 1, phases 2 / 3
};

template <int MT>
__global__ __launch_bounds__(Geo<MT>::THREADS_, (MT == 2 ? 2 : 1)) void hg_residual_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                                                         unsigned short* __restrict__ y, const unsigned short* __restrict__ wpk,
                                                                         const float* __restrict__ bias, int Cin, int H, int W, int tiles_x,
                                                                         int tiles, int rx_elems) {
    using G = Geo<MT>;
    constexpr int HC = G::HC, T1S = G::T1S, COUT = G::COUT, OS = G::OS, NG = G::NG, N1 = G::N1, N2 = G::N2, NTH = G::THREADS_;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* rx = lds;                                // phase 1: raw x patch [NDP][Cin + 8]; then t2 [NOUT][T1S] at its start
    unsigned short* t1 = lds + rx_elems;                     // [NDP][T1S]
    unsigned short* t2 = rx;
    unsigned short* ot = t1 + NDP * T1S - NOUT * OS;         // the output tile ends where t1 ends: over t1 (dead by then) and the patch's tail, clear of t2
    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mg = wave % MT, ng = wave / MT;
    const int b = blockIdx.x / tiles, tile = blockIdx.x - b * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int ho0 = ty * TH, wo0 = tx * TW;
    const int XS = Cin + 8, F1 = Cin / 16;
    const unsigned short* xb = x + (size_t)b * H * W * Cin;
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(wpk) + lane;          // fragment f of this lane: wq[64 f]
    const bf16x8* w1 = wq + (size_t)64 * mg * F1;
    const bf16x8* w2 = wq + (size_t)64 * (MT * F1 + mg * MT * 18);
    const bf16x8* w3 = wq + (size_t)64 * (MT * F1 + MT * MT * 18);

    [[maybe_unused]] const bool hprb = lane == 0 && blockIdx.x == gridDim.x / 2;
    HPROBE(0);
    // ALL weight fragments of phase 1 (Cin / 16 <= 16): in flight while the patch is staged
    bf16x8 a1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a1[i] = w1[64 * (i < F1 ? i : 0)];

    // biases -> LDS (read at the three epilogues; 64 VGPRs if held)
    float* bl = reinterpret_cast<float*>(t1 + NDP * T1S);
    for (int i = tid; i < 2 * COUT; i += NTH) bl[i] = bias[i];
    // ---- the RAW x patch, all channels, by LDS-DMA (no VGPRs, every piece in flight at once): LDS row = Cin / 8 data slots of 16
    // bytes + one padding slot; a wave instruction fills 64 consecutive slots, lane by lane, each lane from its own global address --
    // its pixel's channel octet, or 16 bytes of zeros for the padding slot and for pixels outside the image.  The ReLU of
    // hourglass.py:44 is applied when phase 1 reads its B operands (one v_pk_max_i16 per register).
    {
        const int SPR = (Cin >> 3) + 1, NS = NDP * SPR, npiece = (NS + 63) >> 6;
        const int dstep = (64 * G::WAVES) / SPR, rstep = 64 * G::WAVES - dstep * SPR;
        int sidx = 64 * wave + lane;
        int row = sidx / SPR, slot = sidx - row * SPR;
        for (int pc = wave; pc < npiece; pc += G::WAVES) {
            const int qy = row / DW, qx = row - qy * DW;
            const int gy = ho0 - 1 + qy, gx = wo0 - 1 + qx;
            const bool in = row < ND && slot < SPR - 1 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned short* src = in ? xb + (size_t)(gy * W + gx) * Cin + 8 * slot : reinterpret_cast<const unsigned short*>(islam_hg_zero);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(rx + pc * 512), 16, 0, 0);
            row += dstep; slot += rstep;
            if (slot >= SPR) { slot -= SPR; ++row; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces have landed
    HPROBE(1);
    __syncthreads();                                         // barrier 1: patch staged
    HPROBE(2);

    // ================= phase 1: t1 = relu(bf16(W1 relu(x) + b1)), this wave's 32 channels x N1 pixel tiles =================
    {
        f32x16 acc[N1];
#pragma unroll
        for (int n = 0; n < N1; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
        const unsigned short* bb = rx + (size_t)(ng * 32 + li) * XS + 8 * kg;       // N tile ng + NG n: rows 32 (ng + NG n) + li
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < F1) {                                    // (uniform)
#pragma unroll
                for (int n = 0; n < N1; ++n) {
                    const bf16x8 bf = relu8(*reinterpret_cast<const bf16x8*>(bb + (size_t)(NG * n * 32) * XS + 16 * i));
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], bf, acc[n], 0, 0, 0);
                }
            }
        HPROBE(3);
        f32x4 bv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bv[gq] = *reinterpret_cast<const f32x4*>(bl + 32 * mg + 8 * gq + 4 * kg);
#pragma unroll
        for (int n = 0; n < N1; ++n) {
            const int q = (ng + NG * n) * 32 + li;
            const int qy = q / DW, qx = q - qy * DW;
            const int gy = ho0 - 1 + qy, gx = wo0 - 1 + qx;
            const bool in = q < ND && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                unsigned p0 = relu2(pack2(acc[n][4 * gq] + bv[gq].x, acc[n][4 * gq + 1] + bv[gq].y));
                unsigned p1 = relu2(pack2(acc[n][4 * gq + 2] + bv[gq].z, acc[n][4 * gq + 3] + bv[gq].w));
                if (!in) { p0 = 0; p1 = 0; }                 // conv2's zero padding is a zero of ITS input, not relu(b1)
                *reinterpret_cast<uint2*>(t1 + (size_t)q * T1S + 32 * mg + 8 * gq + 4 * kg) = make_uint2(p0, p1);
            }
        }
    }
    // first fragments of phase 2: requested before the barrier
    constexpr int F2 = MT * 18, D2 = 18;                     // fragments of phase 2 per wave, ring depth: one 32-channel chunk ahead
    bf16x8 ring[D2];
#pragma unroll
    for (int i = 0; i < D2; ++i) ring[i] = w2[64 * i];
    VMEM_PIN();
    HPROBE(4);
    __syncthreads();                                         // barrier 2: t1 complete (the patch is dead)
    HPROBE(5);

    // ================= phase 2: t2 = relu(bf16(W2 * t1 + b2)) =================
    // N tile = tile rows 2 nt, 2 nt + 1.  Second-row lanes take pixel x = (li - 18) mod 16, so that the patch rows a 16-lane group of
    // ds_read_b128 touches stay distinct mod 16 (t1's row pitch is 18 pixels): conflict-free operand reads.
    const int pyl = li >> 4, px2 = li < 16 ? li : ((li - 18) & 15);
    {
        f32x16 acc[N2];
#pragma unroll
        for (int n = 0; n < N2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
        const unsigned short* bb = t1 + (size_t)((2 * ng * N2 + pyl) * DW + px2) * T1S + 8 * kg;      // N tile nt = ng N2 + n: + 2 n rows
        static_for<0, F2>([&](auto ff) {
            constexpr int f = decltype(ff)::value;
            constexpr int c = f / 18, tap = (f % 18) / 2, ks2 = f % 2, r = tap / 3, s_ = tap % 3;
            const bf16x8 af = ring[f % D2];
            if constexpr (f + D2 < F2) ring[f % D2] = w2[64 * (f + D2)];
            VMEM_PIN();                                      // (the scheduler otherwise sinks every load next to its use: one load in flight)
#pragma unroll
            for (int n = 0; n < N2; ++n) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bb + (size_t)((2 * n + r) * DW + s_) * T1S + 32 * c + 16 * ks2);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[n], 0, 0, 0);
            }
        });
        HPROBE(6);
        f32x4 bv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bv[gq] = *reinterpret_cast<const f32x4*>(bl + HC + 32 * mg + 8 * gq + 4 * kg);
#pragma unroll
        for (int n = 0; n < N2; ++n) {
            const int p = (2 * (ng * N2 + n) + pyl) * TW + px2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const unsigned p0 = relu2(pack2(acc[n][4 * gq] + bv[gq].x, acc[n][4 * gq + 1] + bv[gq].y));
                const unsigned p1 = relu2(pack2(acc[n][4 * gq + 2] + bv[gq].z, acc[n][4 * gq + 3] + bv[gq].w));
                *reinterpret_cast<uint2*>(t2 + (size_t)p * T1S + 32 * mg + 8 * gq + 4 * kg) = make_uint2(p0, p1);
            }
        }
    }
    // all fragments of phase 3 for this wave's two M tiles (2 x h/16 <= 16 fragments)
    constexpr int F3 = HC / 16;
    bf16x8 a3[2][F3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < F3; ++ks) a3[m][ks] = w3[64 * ((mg + MT * m) * F3 + ks)];
    VMEM_PIN();
    HPROBE(7);
    __syncthreads();                                         // barrier 3: t2 complete
    HPROBE(8);
    // the residual values of all this thread's output items: requested now, used behind barrier 4
    constexpr int OCT = COUT / 8, NIT = (NOUT * OCT + NTH - 1) / NTH;
    u32x4 rv[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int it = tid + u * NTH, px = it / OCT, oc = it - px * OCT;
        const int ho = ho0 + (px >> 4), wo = wo0 + (px & 15);
        const bool ok = it < NOUT * OCT && ho < H && wo < W;
        rv[u] = *reinterpret_cast<const u32x4*>(res + (ok ? (((size_t)b * H + ho) * W + wo) * COUT + 8 * oc : (size_t)0));
    }
    VMEM_PIN();

    // ================= phase 3: y = bf16(bf16(W3 * t2 + b3) + res) =================
    {
        f32x16 acc[2][N2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < N2; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;
        const unsigned short* bb = t2 + (size_t)(ng * N2 * 32 + li) * T1S + 8 * kg;      // natural pixel order: t2's pitch is 16 pixels
#pragma unroll
        for (int ks = 0; ks < F3; ++ks)
#pragma unroll
            for (int n = 0; n < N2; ++n) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bb + (size_t)(n * 32) * T1S + 16 * ks);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[m][ks], bf, acc[m][n], 0, 0, 0);
            }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ch = 32 * (mg + MT * m);
            f32x4 bv[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bv[gq] = *reinterpret_cast<const f32x4*>(bl + 2 * HC + ch + 8 * gq + 4 * kg);
#pragma unroll
            for (int n = 0; n < N2; ++n) {
                const int p = (ng * N2 + n) * 32 + li;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<uint2*>(ot + (size_t)p * OS + ch + 8 * gq + 4 * kg) =
                        make_uint2(pack2(acc[m][n][4 * gq] + bv[gq].x, acc[m][n][4 * gq + 1] + bv[gq].y),
                                   pack2(acc[m][n][4 * gq + 2] + bv[gq].z, acc[m][n][4 * gq + 3] + bv[gq].w));
            }
        }
    }
    HPROBE(9);
    __syncthreads();                                         // barrier 4: output tile staged
    HPROBE(10);
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int it = tid + u * NTH, px = it / OCT, oc = it - px * OCT;
        const int ho = ho0 + (px >> 4), wo = wo0 + (px & 15);
        if (it >= NOUT * OCT || ho >= H || wo >= W) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(ot + (size_t)px * OS + 8 * oc);
        const u32x4 r = rv[u];
        v.x = pack2(lo16(v.x) + lo16(r.x), hi16(v.x) + hi16(r.x));
        v.y = pack2(lo16(v.y) + lo16(r.y), hi16(v.y) + hi16(r.y));
        v.z = pack2(lo16(v.z) + lo16(r.z), hi16(v.z) + hi16(r.z));
        v.w = pack2(lo16(v.w) + lo16(r.w), hi16(v.w) + hi16(r.w));
        *reinterpret_cast<u32x4*>(y + (((size_t)b * H + ho) * W + wo) * COUT + 8 * oc) = v;
    }
    HPROBE(11);
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

template <int MT>
int launch(const unsigned short* x, const unsigned short* res, unsigned short* y, const unsigned short* wpk, const float* bias, int B, int Cin,
           int H, int W, hipStream_t s) {
    using G = Geo<MT>;
    const int npiece = (NDP * (Cin / 8 + 1) + 63) / 64;      // whole 1-KiB LDS-DMA pieces of the patch
    const int rx = std::max(npiece * 512, NOUT * G::T1S + NOUT * G::OS - NDP * G::T1S);      // t2 | ... | output tile (ends with t1) must not overlap
    const size_t lds = ((size_t)rx + (size_t)NDP * G::T1S) * sizeof(unsigned short) + (size_t)2 * G::COUT * sizeof(float);
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)hg_residual_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    if (lds > 160 * 1024) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: %zu bytes of LDS for Cin=%d, Cout=%d", lds, Cin, G::COUT);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL((hg_residual_kernel<MT>), dim3(tiles_x * tiles_y * B), dim3(G::THREADS_), lds, s, x, res, y, wpk, bias, Cin, H, W, tiles_x,
                       tiles_x * tiles_y, rx);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // namespace

extern "C" {

#if ISLAM_HG_PROBE
int islam_hg_probe_read(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(islam_hg_probe_buf), sizeof(long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

size_t islam_hg_residual_packed_elems(int Cin, int Cout) {
    if (Cin < 64 || (Cin & 63) || Cout < 64 || (Cout & 63) || Cout > 256 || (Cout == 64 && Cin != 64)) return 0;
    if (Cin == 64 && Cout == 64) return (size_t)L_NFRAG * 512;       // register-resident fragments (hg_residual64_kernel)
    const int MT = Cout / 64;
    return (size_t)512 * (MT * (Cin / 16) + 18 * MT * MT + 2 * MT * 2 * MT);
}

int islam_hg_residual_nhwc_bf16(const uint16_t* x, const uint16_t* res, uint16_t* y, const uint16_t* wpacked, const float* bias, int B, int Cin,
                                int H, int W, int Cout, void* stream) {
    if (B < 1 || H < 1 || W < 1 || Cin < 64 || (Cin & 63) || Cin > 256 || Cout < 64 || (Cout & 63) || Cout > 256 || (Cout == 64 && Cin != 64))
        return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: bad shape (Cin=%d, Cout=%d: multiples of 64 up to 256; Cout = 64 with Cin = 64 only)", Cin, Cout);
    if (!x || !res || !y || !wpacked || !bias) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: null argument");
    if ((size_t)B * H * W * std::max(Cin, Cout) >= ((size_t)1 << 31)) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: tensor too large for 32-bit offsets");
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64 && Cout == 64) {
        if (res != x) return fail(ISLAM_EARG, "islam_hg_residual_nhwc_bf16: the 64 -> 64 module has no skip convolution (res must be x)");
        return launch64(x, y, wpacked, bias, B, H, W, s);
    }
    switch (Cout / 64) {
        case 2: return launch<2>(x, res, y, wpacked, bias, B, Cin, H, W, s);
        case 3: return launch<3>(x, res, y, wpacked, bias, B, Cin, H, W, s);
        default: return launch<4>(x, res, y, wpacked, bias, B, Cin, H, W, s);
    }
}

}  // extern "C"
