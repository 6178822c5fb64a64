// conv_ws.hip -- weight-STATIONARY 3x3 stride-1 convolution, 128 (or 64) -> 128 channels, bf16 channels-last, for the 128-channel
// residual blocks of the frozen stereo network's feature extractor (Network/PSM/submodule.py:10-13 `convbn`, :24-43 BasicBlock, :66-155
// feature_extraction layer3 / layer4 -- eleven 128 -> 128 convolutions per forward at 112 x 160 x 16 images, the largest share of the
// frozen nets' kernel time, and the 64 -> 128 one that opens layer3).  Same arithmetic contract as conv_nhwc.hip's conv_nhwc_kernel
// (bf16 operands, fp32 accumulation in the same order, output rounded to nearest even, the PREVIOUS BatchNorm + ReLU applied while the
// input is staged, THIS layer's BatchNorm partial sums from the epilogue): the outputs are bit-identical to that kernel's.  Different
// machine mapping (numbers for 128 input channels; Cfg<64> halves the weights, the steps per tile and the staging items):
//   * conv_nhwc_kernel re-stages the weight taps (9 x 64 x 32 bf16 = 63 % of its LDS staging traffic) for every 32 x 8-pixel tile and
//     every 32-channel chunk, reads 0.83 LDS operands per MFMA, and lives 25 us per workgroup of which 6.5 us are set-up and epilogue;
//   * here ONE workgroup per CU (4 waves, one per SIMD) owns the gfx950 register file: wave w keeps the weights of output channels
//     32 w .. 32 w + 31 for all 9 taps x 128 input channels in 288 registers (all 256 AGPRs + 32 VGPRs) for the whole launch and walks
//     a contiguous range of 32 x 4-pixel tiles (persistent, XCD-contiguous, vertical neighbours back to back).  Only activations go
//     through LDS: the 6 x 34-pixel halo tile of all 128 input channels is staged ONCE per tile into one of two buffers while the
//     other one is being multiplied; 0.5 LDS operand reads per MFMA, 288 MFMAs per wave between two barriers.
//   * one wave per SIMD has no partner to hide anything behind, so the instruction stream is laid out by hand: every tile's multiply
//     phase is 288 slots of {one MFMA; at most ~4 independent VALU / one memory instruction}, and everything else the kernel has to do
//     -- staging the next tile (13 items per thread), requesting the tile after it, storing the PREVIOUS tile's outputs and summing its
//     BatchNorm statistics -- is cut into micro-operations that ride in those slots (scripts/probes/mfma_valu_overlap.hip: up to
//     ~5 VALU instructions issue for free in the shadow of a v_mfma_f32_32x32x16_bf16; a clump of 30 between two MFMAs, or a dependent
//     chain of 6, leaves the matrix pipe idle).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
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
template <int V> using ic = std::integral_constant<int, V>;
// The values must exist HERE: sched_barrier fences the machine scheduler only, and the optimiser otherwise sinks a micro-operation's
// arithmetic to the place its result is finally needed (all eight store items' statistics ended up in one clump of 100 instructions).
#define PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

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

constexpr int TN = 128, ROWS = 4, TW = 32, THREADS = 256;
constexpr int IH = ROWS + 2, IW = TW + 2, NPIX = IH * IW;          // 6 x 34 halo pixels
constexpr int TS = TN + 8, OCT = TN / 8, PPT = TW * ROWS * OCT / THREADS;      // output staging tile: [128 pixels][TS]; 8 items per thread
constexpr int SLOTS = 12;                                          // MFMAs of a (k-slice, horizontal tap) step
// Everything that depends on the number of input channels (128: the residual blocks of layer3 / layer4; 64: layer3's first convolution)
template <int CIN_>
struct Cfg {
    static constexpr int CIN = CIN_;
    static constexpr int PS = CIN + 8;                             // LDS pixel stride (elements): 272 / 144 bytes, conflict-free 16-byte reads
    static constexpr int OPP = CIN / 8;                            // 16-byte octets per pixel
    static constexpr int PPI = THREADS / OPP;                      // halo pixels one round of staging items covers (16 / 32)
    static constexpr int NIN = (NPIX * OPP + THREADS - 1) / THREADS;       // staging items per thread (13 / 7)
    static constexpr int NPRE = (NIN + 1) / 2;                     // 16-byte registers they go through: item k and item k + NPRE share one (7 / 4)
    static constexpr int NB = NIN - NPRE;                          // items of the second batch
    static constexpr int HALO = NPIX * PS;                         // elements of one halo buffer
    static constexpr int NK = CIN / 16;                            // k-slices of one MFMA (16 input channels)
    static constexpr int NSTEP = 3 * NK;                           // (k-slice, horizontal tap) steps of 12 MFMAs
    static constexpr int NW = 9 * NK, NWA = NW < 64 ? NW : 64;     // weight operands of a wave; those that live in AGPRs
    // LDS (elements of 2 bytes): two halo buffers | output staging tile | [scale | shift] of the input's BatchNorm (2 x CIN floats) | 16 bytes
    static constexpr int L_TL = 2 * HALO, L_AFF = L_TL + TW * ROWS * TS, L_DUMMY = L_AFF + 4 * CIN;
    static constexpr size_t LDS_BYTES = (size_t)(L_DUMMY + 8) * sizeof(unsigned short);
    // the weights enter through LDS, TPR taps per round, behind the first halo buffer
    static constexpr int TPR = (HALO + TW * ROWS * TS) / (TN * PS), NROUND = (9 + TPR - 1) / TPR;
    // what rides where in a tile's multiply phase (see `ride`): first batch staged at steps [0, NPRE), store phase of the previous tile
    // in the five steps behind it, second batch at [BS0, BS0 + NB) together with the requests of the tile after the next
    static constexpr int ST0 = NPRE, BS0 = NPRE + 5;
    static_assert(BS0 + NB <= NSTEP && NPRE - NB <= NB && 7 * PPT + 1 <= 5 * SLOTS, "the riders fit the multiply phase");
    static_assert((size_t)THREADS * 17 * sizeof(float) <= (size_t)2 * HALO * sizeof(unsigned short), "the final reduction reuses the halo buffers");
    static_assert((TPR * TN * OPP) % THREADS == 0 && ((9 - (NROUND - 1) * TPR) * TN * OPP) % THREADS == 0, "whole 16-byte pieces per thread and round");
};

// One MFMA with the weight operand taken straight from where it lives.  At 128 input channels the 72 weight operands of a wave are 288
// registers: the first NWA = 64 of them fill the 256 AGPRs, the rest sit in VGPRs beside the accumulators (64 channels: 36, all in AGPRs).  Written as inline assembly because the register
// allocator otherwise treats the AGPRs as spill space for VGPR-class values and copies every operand back (v_accvgpr_read / _mov, four per
// MFMA operand, ~300 VALU-class instructions per tile).  FIRST: the accumulator starts from zero.  The compiler does not know these are
// MFMAs: the hazard it would guard (MFMA result -> VALU read) is covered by the s_nop in front of the epilogue; source registers are only
// ever rewritten by LDS reads that return long after the MFMA has read them.
template <int J, bool FIRST, int NWA, int NW>
__device__ __forceinline__ void mma(f32x16& acc, const bf16x8 (&wr)[NW], const bf16x8& b) {
    if constexpr (J < NWA) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(wr[J]), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wr[J]), "v"(b));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wr[J]), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wr[J]), "v"(b));
    }
}

// the 12 MFMAs of a step in the order that frees operand row 0 first: MFMA m reads halo row MQ[m] with vertical tap MR[m] (output row
// p = q - r; every accumulator still sees r = 0, 1, 2 in order); after the last reader of a row its register is refilled for the next step
constexpr int MQ[SLOTS] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5};
constexpr int MR[SLOTS] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 1, 2, 2};
constexpr bool MLAST[SLOTS] = {true, false, true, false, false, true, false, false, true, false, true, true};

#ifdef ISLAM_WS_STAMPS                  // scripts/debug/conv_ws_stamps.sh: phase clocks of one workgroup's wave 0 (never in the product build)
}  // namespace
__device__ long long islam_ws_stamps[16];
namespace {
#define WSTAMP(j) do { __builtin_amdgcn_sched_barrier(0); if (stamp) { const long long now_ = clock64(); acc_t[j] += now_ - last_; last_ = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WSTAMP(j) do { } while (0)
#endif

template <int CIN, bool AFFINE>
__global__ __launch_bounds__(THREADS, 1) void conv3x3_ws_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wp,
                                                                const float* __restrict__ in_affine, unsigned short* __restrict__ y,
                                                                float* __restrict__ partial, int H, int W, int tiles_x, int tiles_y, int ntiles,
                                                                int xs, int xoff, int ys, int yoff, int Cout, int CoutP) {
    using K = Cfg<CIN>;
    constexpr int PS = K::PS, OPP = K::OPP, PPI = K::PPI, NIN = K::NIN, NPRE = K::NPRE, NB = K::NB, HALO = K::HALO, NK = K::NK, NSTEP = K::NSTEP;
    constexpr int L_TL = K::L_TL, L_AFF = K::L_AFF, L_DUMMY = K::L_DUMMY, ST0 = K::ST0, BS0 = K::BS0;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* tl = lds + L_TL;                               // output staging tile
    const float* afl = reinterpret_cast<const float*>(lds + L_AFF);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kg = lane >> 5, li = lane & 31;
    const int G = gridDim.x;
    const int wgl = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;      // XCD x: a contiguous range of workgroups
    const int t0 = (int)((long long)wgl * ntiles / G), t1 = (int)((long long)(wgl + 1) * ntiles / G);

#ifdef ISLAM_WS_STAMPS
    const long long begin_ = wall_clock64();
#endif
    const int oct = tid & (OPP - 1), prow = tid / OPP;            // the thread's channel octet (the same for all its items), first halo pixel
    if constexpr (AFFINE) { if (tid < 2 * CIN) reinterpret_cast<float*>(lds + L_AFF)[tid] = in_affine[tid]; }      // [scale(CIN) | shift(CIN)]
    f32x4 s0 = {1, 1, 1, 1}, s1 = s0, h0 = {0, 0, 0, 0}, h1 = h0;
    auto load_affine = [&]() {                                     // (reloaded where a staging block starts: the 16 registers are free in between)
        if constexpr (AFFINE) {
            const float* a8 = afl + 8 * oct;
            s0 = *reinterpret_cast<const f32x4*>(a8); s1 = *reinterpret_cast<const f32x4*>(a8 + 4);
            h0 = *reinterpret_cast<const f32x4*>(a8 + CIN); h1 = *reinterpret_cast<const f32x4*>(a8 + CIN + 4);
        }
    };

    // ---- staging: item k of a thread = halo pixel prow + PPI k, channel octet oct.  Requests (global -> registers) and the staging proper
    // (normalise, zero-pad, registers -> LDS) are separate micro-operations so that they can ride in the MFMA slots; everything about
    // an item that does not depend on the tile is computed once: its byte offset from the tile's first halo pixel and its bit in the
    // masks of the halo's top / bottom row and left / right column (whole tiles only: these are the only pixels that can lie outside).
    unsigned relb[NIN], m_pix = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
    static_for<0, NIN>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int pix = prow + PPI * k, yy = pix / IW, xx = pix - yy * IW;
        relb[k] = (unsigned)(((yy * W + xx) * xs + 8 * oct) * 2);
        m_pix |= (unsigned)(pix < NPIX) << k;
        m_top |= (unsigned)(yy == 0) << k; m_bot |= (unsigned)(yy == IH - 1) << k;
        m_left |= (unsigned)(xx == 0) << k; m_right |= (unsigned)(xx == IW - 1) << k;
    });
    const unsigned safeb = (unsigned)(((W + 1) * xs + 8 * oct) * 2);      // the tile's first interior pixel: what masked items read
    const unsigned lds_st = (unsigned)((prow * PS + 8 * oct) * 2);
    // (the OUTPUT tile has 128 channels whatever the input has: store item j of a thread = pixel tid / 16 + 16 j, channel octet tid % 16)
    const unsigned tl_ld = (unsigned)(((tid / OCT) * TS + 8 * (tid & (OCT - 1))) * 2), g_st = (unsigned)(((tid / OCT) * ys + 8 * (tid & (OCT - 1))) * 2);
    u32x4 pre[NPRE];
    unsigned okmask = 0, oknext = 0;                                // validity bits of the tile being staged / of the tile requested last
    float tf[4] = {0, 0, 0, 0};
    const char* fbase = nullptr;                                   // first halo pixel of the tile being requested (uniform; may lie outside the tensor)
    auto fetch_tile = [&](int t) {
        const int tc = t < ntiles ? t : ntiles - 1;                // (past the range: a valid address, everything masked)
        const int ty = tc % tiles_y, q = tc / tiles_y, tx = q % tiles_x, b = q / tiles_x;
        fbase = reinterpret_cast<const char*>(x + xoff) + ((long long)((b * H + ty * ROWS - 1) * W + tx * TW - 1) * xs) * 2;
        const unsigned out = (ty == 0 ? m_top : 0u) | (ty == tiles_y - 1 ? m_bot : 0u) | (tx == 0 ? m_left : 0u) | (tx == tiles_x - 1 ? m_right : 0u);
        oknext = t < t1 ? (m_pix & ~out) : 0u;
    };
    auto fetch = [&](auto kk, unsigned mask) {                     // request of item k
        constexpr int k = decltype(kk)::value;
        const unsigned off = ((mask >> k) & 1u) ? relb[k] : safeb;
        pre[k % NPRE] = *reinterpret_cast<const u32x4*>(fbase + off);
    };
    // staging of item k in nine parts: 0-2 normalise dwords x, y (two independent chains side by side), 3-5 dwords z, w, 6-7 zero padding, 8 write
    auto stage_part = [&](auto kk, auto pp, unsigned short* dst) {
        constexpr int k = decltype(kk)::value, part = decltype(pp)::value;
        u32x4& v = pre[k % NPRE];
        if constexpr (part == 0 || part == 3) {
            if constexpr (AFFINE) {
                const unsigned a = part == 0 ? v.x : v.z, b = part == 0 ? v.y : v.w;
                tf[0] = lo16(a); tf[1] = hi16(a); tf[2] = lo16(b); tf[3] = hi16(b);
            }
        } else if constexpr (part == 1 || part == 4) {
            if constexpr (AFFINE) {
                const f32x4 sv = part == 1 ? s0 : s1, hv = part == 1 ? h0 : h1;
                tf[0] = fmaf(tf[0], sv.x, hv.x); tf[1] = fmaf(tf[1], sv.y, hv.y); tf[2] = fmaf(tf[2], sv.z, hv.z); tf[3] = fmaf(tf[3], sv.w, hv.w);
            }
        } else if constexpr (part == 2) {
            if constexpr (AFFINE) { v.x = relu2(pack2(tf[0], tf[1])); v.y = relu2(pack2(tf[2], tf[3])); }
        } else if constexpr (part == 5) {
            if constexpr (AFFINE) { v.z = relu2(pack2(tf[0], tf[1])); v.w = relu2(pack2(tf[2], tf[3])); }
        } else if constexpr (part == 6) {
            const bool ok = (okmask >> k) & 1u;                    // zero padding of the NORMALISED activation
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u;
        } else if constexpr (part == 7) {
            const bool ok = (okmask >> k) & 1u;
            v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
        } else {
            char* d = reinterpret_cast<char*>(dst) + lds_st + k * PPI * PS * 2;
            if constexpr (k == NIN - 1) d = ((m_pix >> k) & 1u) ? d : reinterpret_cast<char*>(lds + L_DUMMY);      // (the last item: not every thread has one)
            *reinterpret_cast<u32x4*>(d) = v;
        }
    };
    auto stage = [&](auto kk, unsigned short* dst) { static_for<0, 9>([&](auto pp) { stage_part(kk, pp, dst); }); };

    // ---- store phase of a finished tile, item j of a thread = pixel tid / 16 + 16 j of the 32 x 4 tile (row j / 2, column tid / 16 + 16 (j % 2)),
    // channel octet tid % 16: read the bf16 tile from LDS (16 bytes), add to the thread's BatchNorm sums (of the stored, rounded values), store
    // 16 bytes (16 lanes = one pixel's 256 bytes)
    float sm[8], sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[i] = 0.0f; sq[i] = 0.0f; }
    u32x4 ev[2];
    char* ebase = nullptr;                                         // first pixel of the tile being stored (uniform)
    auto store_tile = [&](int t) {
        const int ty = t % tiles_y, q = t / tiles_y, tx = q % tiles_x, b = q / tiles_x;
        ebase = reinterpret_cast<char*>(y + yoff) + ((long long)((b * H + ty * ROWS) * W + tx * TW) * ys) * 2;
    };
    auto store_part = [&](auto jj, auto pp) {
        constexpr int j = decltype(jj)::value, part = decltype(pp)::value;
        if constexpr (part == 0) {
            ev[j & 1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tl) + tl_ld + j * 16 * TS * 2);
        } else if constexpr (part == 1 || part == 4) {
            const u32x4 v = ev[j & 1];
            const unsigned a = part == 1 ? v.x : v.z, b = part == 1 ? v.y : v.w;
            tf[0] = lo16(a); tf[1] = hi16(a); tf[2] = lo16(b); tf[3] = hi16(b);
            PIN4(tf[0], tf[1], tf[2], tf[3]);
            if constexpr (part == 1 && j + 1 < PPT)
                ev[(j + 1) & 1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tl) + tl_ld + (j + 1) * 16 * TS * 2);
        } else if constexpr (part == 2 || part == 5) {
            constexpr int o = part == 2 ? 0 : 4;
            sm[o] += tf[0]; sm[o + 1] += tf[1]; sm[o + 2] += tf[2]; sm[o + 3] += tf[3];
            PIN4(sm[o], sm[o + 1], sm[o + 2], sm[o + 3]);
        } else if constexpr (part == 3 || part == 6) {
            constexpr int o = part == 3 ? 0 : 4;
            sq[o] = fmaf(tf[0], tf[0], sq[o]); sq[o + 1] = fmaf(tf[1], tf[1], sq[o + 1]);
            sq[o + 2] = fmaf(tf[2], tf[2], sq[o + 2]); sq[o + 3] = fmaf(tf[3], tf[3], sq[o + 3]);
            PIN4(sq[o], sq[o + 1], sq[o + 2], sq[o + 3]);
        } else {
            char* rowp = ebase + ((long long)((j >> 1) * W + 16 * (j & 1)) * ys) * 2;      // (uniform)
            *reinterpret_cast<u32x4*>(rowp + g_st) = ev[j & 1];
        }
    };

    // ---- what rides in slot (step i, MFMA m) of tile t's multiply phase (128 input channels: 24 steps, 13 items; 64: 12 steps, 7 items):
    //   steps [0, NPRE)          staging of the A items of tile t + 1 (requested during tile t - 1); slot 10: the request of B item i + NPRE
    //   steps [NPRE, NPRE + 5)   store phase of tile t - 1 (57 micro-operations)
    //   steps [BS0, BS0 + NB)    staging of the B items of tile t + 1; slot 9 of the first: the coordinates of tile t + 2; slot 10: the
    //                            request of its A item i - BS0 into the register the B item has just left; slot 11: its other A items
    auto ride = [&](auto ii, auto mm, auto has_prev, int t, unsigned short* bufn) {
        constexpr int i = decltype(ii)::value, m = decltype(mm)::value;
        constexpr bool HAS_PREV = decltype(has_prev)::value;
        if constexpr (i < NPRE) {
            if constexpr (i == 0 && m == 0) load_affine();
            if constexpr (m < 9) stage_part(ic<i>{}, mm, bufn);
            else if constexpr (m == 10 && i + NPRE < NIN) fetch(ic<i + NPRE>{}, okmask);
        } else if constexpr (i < BS0) {
            if constexpr (HAS_PREV) {
                constexpr int e = (i - ST0) * SLOTS + m;           // 0: first LDS read; then 7 parts per item
                if constexpr (e == 0) { store_tile(t - 1); store_part(ic<0>{}, ic<0>{}); }
                else if constexpr (e <= 7 * PPT) store_part(ic<(e - 1) / 7>{}, ic<(e - 1) % 7 + 1>{});
            }
            if constexpr (i == BS0 - 1 && m == 11) load_affine();
        } else if constexpr (i < BS0 + NB) {
            if constexpr (m < 9) stage_part(ic<i - BS0 + NPRE>{}, mm, bufn);
            else if constexpr (m == 9) { if constexpr (i == BS0) fetch_tile(t + 2); }
            else if constexpr (m == 10) fetch(ic<i - BS0>{}, oknext);
            else if constexpr (NB + (i - BS0) < NPRE) fetch(ic<NB + (i - BS0)>{}, oknext);
        }
    };

    unsigned short* bufc = lds;                                    // the halo tile being multiplied / the one being filled
    unsigned short* bufn = lds + HALO;
    // the first tile: all its items are requested BEFORE the weights (nothing else needs registers yet) and staged behind them
    u32x4 first[NIN];
    fetch_tile(t0);
    okmask = oknext;
    static_for<0, NIN>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        fetch(kk, okmask);
        first[k] = pre[k % NPRE];
    });
    // ---- the wave's weights: output channels 32 wave + li (A operand rows), taps x k-slices, for the whole launch.  A lane's operand is 16
    // bytes of a weight row, so direct loads touch 32 cache lines per instruction for a fraction of each (8-10 us for the 72 loads at 128
    // channels, with every workgroup of the chip asking the L2 for the same lines); instead the workgroup copies TPR taps at a time
    // (TPR x 128 rows, consecutive lanes = consecutive 16 bytes) into LDS rows of PS elements and every lane picks its operands from
    // there with the conflict-free pattern of the B operand reads.
    bf16x8 wr[K::NW];
    {
        constexpr int TPR = K::TPR, NROUND = K::NROUND;
        unsigned short* wl = lds + HALO;                           // second halo buffer + output tile
        u32x4 wv[2][TPR * TN * OPP / THREADS];                     // the pieces of round r + 1 are requested before round r is copied
        auto request = [&](auto rr) {
            constexpr int r = decltype(rr)::value, ntap = r + 1 < NROUND ? TPR : 9 - r * TPR, NP = ntap * TN * OPP / THREADS;      // 16-byte pieces per thread
            static_for<0, NP>([&](auto nn) {
                constexpr int n = decltype(nn)::value;
                const int pc = tid + n * THREADS, tl_ = pc / (TN * OPP), row = (pc / OPP) % TN, o = pc % OPP;
                wv[r & 1][n] = *reinterpret_cast<const u32x4*>(wp + ((size_t)((TPR * r + tl_) * CoutP + row)) * CIN + 8 * o);
            });
        };
        request(ic<0>{});
        static_for<0, NROUND>([&](auto rr) {
            constexpr int r = decltype(rr)::value, ntap = r + 1 < NROUND ? TPR : 9 - r * TPR, NP = ntap * TN * OPP / THREADS;
            if constexpr (r + 1 < NROUND) request(ic<r + 1>{});
            if constexpr (r > 0) __syncthreads();                  // the previous round's operand reads are done
            static_for<0, NP>([&](auto nn) {
                constexpr int n = decltype(nn)::value;
                const int pc = tid + n * THREADS, tl_ = pc / (TN * OPP), row = (pc / OPP) % TN, o = pc % OPP;
                *reinterpret_cast<u32x4*>(wl + (tl_ * TN + row) * PS + 8 * o) = wv[r & 1][n];
            });
            __syncthreads();
            static_for<0, ntap * NK>([&](auto ii) {
                constexpr int i = decltype(ii)::value, tl_ = i / NK, ks = i % NK;
                wr[(TPR * r + tl_) * NK + ks] = *reinterpret_cast<const bf16x8*>(wl + (tl_ * TN + 32 * wave + li) * PS + 16 * ks + 8 * kg);
            });
        });
        __syncthreads();                                           // (the region becomes the halo buffer / output tile)
    }
#ifdef ISLAM_WS_STAMPS
    const long long weights_ = wall_clock64();
#endif
    load_affine();                                                 // (the weight rounds' barriers have published the BatchNorm table)
    static_for<0, NIN>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        pre[k % NPRE] = first[k];
        stage(kk, bufc);
    });
    __syncthreads();
    fetch_tile(t0 + 1);                                            // (past the range: everything masked)
    okmask = oknext;
    static_for<0, NPRE>([&](auto kk) { fetch(kk, okmask); });

#ifdef ISLAM_WS_STAMPS
    const bool stamp = wgl == 3 && tid == 0;
    long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64(), first_ = wall_clock64();
#endif
    auto tile = [&](int t, auto has_prev) {
        WSTAMP(5);
        f32x16 acc[ROWS];
        const unsigned short* bb = bufc + (size_t)li * PS + 8 * kg;
        // B operands: bf[q] = halo row q at column li + s, 16 channels of k-slice ks; the operands of step i + 1 are read DURING step i,
        // each row into the register its step-i value has just left (10-12 MFMAs = more than 320 clocks before its first use)
        bf16x8 bf[IH];
#pragma unroll
        for (int q = 0; q < IH; ++q) bf[q] = *reinterpret_cast<const bf16x8*>(bb + (q * IW) * PS);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NSTEP>([&](auto ii) {
            constexpr int i = decltype(ii)::value, ks = i / 3, s = i % 3;
            constexpr int ks1 = (i + 1) / 3, s1n = (i + 1) % 3;
            static_for<0, SLOTS>([&](auto mm) {
                constexpr int m = decltype(mm)::value, q = MQ[m], r = MR[m];
                mma<(r * 3 + s) * NK + ks, (i == 0 && r == 0), K::NWA, K::NW>(acc[q - r], wr, bf[q]);
                if constexpr (MLAST[m] && i + 1 < NSTEP) bf[q] = *reinterpret_cast<const bf16x8*>(bb + (q * IW + s1n) * PS + 16 * ks1);
                ride(ii, mm, has_prev, t, bufn);
                __builtin_amdgcn_sched_barrier(0);                 // the order written here is the order issued
            });
        });
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");         // (the last MFMAs' results before the VALU below reads them: see mma)
        WSTAMP(0);
        __syncthreads();                                           // every wave is done with bufc; bufn is complete; the store phase has left tl
        WSTAMP(1);
        // ---- accumulators -> LDS.  D row (channel) = (reg&3) + 8*(reg>>2) + 4*kg, D col (pixel x) = li: [pixel][TN] bf16, rounded once
#pragma unroll
        for (int p = 0; p < ROWS; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 32 * wave + 8 * g + 4 * kg;
                *reinterpret_cast<uint2*>(tl + (size_t)(p * TW + li) * TS + nl) =
                    make_uint2(pack2(acc[p][4 * g], acc[p][4 * g + 1]), pack2(acc[p][4 * g + 2], acc[p][4 * g + 3]));
            }
        WSTAMP(2);
        __syncthreads();
        WSTAMP(3);
        unsigned short* tmp = bufc; bufc = bufn; bufn = tmp;
        okmask = oknext;                                           // the tile requested during this one is staged during the next
    };
    if (t0 < t1) tile(t0, std::false_type{});
    for (int t = t0 + 1; t < t1; ++t) tile(t, std::true_type{});
    if (t0 < t1) {                                                 // the last tile's store phase has no multiply phase to ride in
        store_tile(t1 - 1);
        store_part(ic<0>{}, ic<0>{});
        static_for<0, PPT>([&](auto jj) { static_for<1, 8>([&](auto pp) { store_part(jj, pp); }); });
    }
    WSTAMP(4);
#ifdef ISLAM_WS_STAMPS
    if (stamp) { for (int j = 0; j < 6; ++j) islam_ws_stamps[j] = acc_t[j]; islam_ws_stamps[6] = t1 - t0; islam_ws_stamps[7] = wall_clock64() - first_; islam_ws_stamps[8] = weights_ - begin_; islam_ws_stamps[9] = first_ - weights_; }
#endif
    if (partial) {
        // per-workgroup sums over all its tiles, per channel, in a fixed order: thread tid holds the sums of channel octet `oct` over its
        // pixels; one lane per (channel, moment) adds the THREADS / OCT threads of that octet in thread order
        float* red = reinterpret_cast<float*>(lds);                // [THREADS][17] over the halo buffers (nobody reads them any more)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[tid * 17 + i] = sm[i]; red[tid * 17 + 8 + i] = sq[i]; }
        __syncthreads();
        if (tid < 2 * TN) {
            const int which = tid / TN, c = tid - which * TN, o2 = c >> 3, i = c & 7;
            float tsum = 0.0f;
            for (int m = 0; m < THREADS / OCT; ++m) tsum += red[(o2 + OCT * m) * 17 + 8 * which + i];
            partial[((size_t)wgl * 2 + which) * Cout + c] = tsum;
        }
    }
}

}  // namespace

// 0: never, 1: layers with at least 1024 tiles (four per CU: the persistent walk needs a few tiles per workgroup to amortise the
// 72 weight loads per lane and the un-overlapped first tile), 2: every 128 -> 128 3x3 layer (tests, A/B runs).  ISLAM_CONV_WS presets it.
static std::atomic<int> g_mode{-1};
static int ws_mode() {
    int m = g_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = std::getenv("ISLAM_CONV_WS");
        m = !e ? 1 : (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1));
        g_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
static std::atomic<long long> g_launches[2];
void conv_ws_count_launch(int which) { g_launches[which & 1].fetch_add(1, std::memory_order_relaxed); }
void conv_ws_read_counts(long long out[2]) { out[0] = g_launches[0].load(std::memory_order_relaxed); out[1] = g_launches[1].load(std::memory_order_relaxed); }
int conv_ws_set_mode(int mode) {
    const int prev = ws_mode();
    if (mode >= 0 && mode <= 2) g_mode.store(mode, std::memory_order_relaxed);
    return prev;
}

#ifdef ISLAM_WS_STAMPS
}  // namespace islam
extern "C" int islam_conv_ws_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(islam::islam_ws_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}
namespace islam {
#endif

// whole tiles only (no predication in the riding store phase): the image is a multiple of 32 x 4 pixels
bool conv_ws_applies(int Cin, int Cout, int ksize, int B, int H, int W) {
    const int mode = ws_mode();
    if (!mode || ksize != 3 || (Cin != 128 && Cin != 64) || Cout != TN || (H % ROWS) || (W % TW)) return false;
    return mode == 2 || (long long)B * (H / ROWS) * (W / TW) >= 1024;
}

// one workgroup per CU, and never more than the callers' room for per-workgroup partial sums (islam_conv_nhwc_stat_blocks: one row per
// 32 x 8-pixel tile of the tile kernel)
int conv_ws_balanced(long long ntiles, int slots);
int conv_ws_spare_cus() {                                  // ISLAM_CONV_WS_SPARE: CUs a launch leaves to kernels of other streams (A/B runs)
    static const int n = [] { const char* e = std::getenv("ISLAM_CONV_WS_SPARE"); const int v = e ? std::atoi(e) : 0; return v < 0 ? 0 : (v > 128 ? 128 : v); }();
    return n;
}
int conv_ws_blocks(int B, int H, int W) {
    const long long rows = (long long)B * ((H + 2 * ROWS - 1) / (2 * ROWS)) * ((W + TW - 1) / TW);
    const int cus = 256 - conv_ws_spare_cus();
    const int cap = (int)(rows < cus ? rows : cus);
    return conv_ws_balanced((long long)B * ((H + ROWS - 1) / ROWS) * ((W + TW - 1) / TW), cap);
}
// ... and no more than the makespan needs: with `slots` workgroups the longest range has ceil(ntiles / slots) tiles; the smallest launch
// with ranges no longer than that leaves the other CUs to the kernels of concurrent streams (2240 tiles: 249 workgroups of 9 instead of
// 256 of 8-9) at no cost to this one
int conv_ws_balanced(long long ntiles, int slots) {
    if (ntiles <= slots) return (int)ntiles;
    const long long per = (ntiles + slots - 1) / slots;
    return (int)((ntiles + per - 1) / per);
}

template <int CIN>
static int ws_launch(const unsigned short* x, const unsigned short* wp, const float* in_affine, unsigned short* y, float* partial, int B, int H, int W,
                     int Cout, int CoutP, int xs, int xoff, int ys, int yoff, hipStream_t s) {
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<CIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<CIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    const int tiles_x = W / TW, tiles_y = H / ROWS, ntiles = tiles_x * tiles_y * B;
    const int G = conv_ws_blocks(B, H, W);
    if (in_affine)
        hipLaunchKernelGGL((conv3x3_ws_kernel<CIN, true>), dim3(G), dim3(THREADS), Cfg<CIN>::LDS_BYTES, s, x, wp, in_affine, y, partial, H, W, tiles_x,
                           tiles_y, ntiles, xs, xoff, ys, yoff, Cout, CoutP);
    else
        hipLaunchKernelGGL((conv3x3_ws_kernel<CIN, false>), dim3(G), dim3(THREADS), Cfg<CIN>::LDS_BYTES, s, x, wp, in_affine, y, partial, H, W, tiles_x,
                           tiles_y, ntiles, xs, xoff, ys, yoff, Cout, CoutP);
    ISLAM_LAUNCH_CHECK();
    conv_ws_count_launch(0);
    return ISLAM_OK;
}

int conv_ws_launch(const unsigned short* x, const unsigned short* wp, const float* in_affine, unsigned short* y, float* partial, int B, int Cin, int H,
                   int W, int Cout, int CoutP, int xs, int xoff, int ys, int yoff, hipStream_t s) {
    return Cin == 64 ? ws_launch<64>(x, wp, in_affine, y, partial, B, H, W, Cout, CoutP, xs, xoff, ys, yoff, s)
                     : ws_launch<128>(x, wp, in_affine, y, partial, B, H, W, Cout, CoutP, xs, xoff, ys, yoff, s);
}

}  // namespace islam
